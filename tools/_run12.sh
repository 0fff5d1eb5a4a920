python -m pytest tests -m gpu -q -x --tb=short --timeout 900 > gpurun_out/r05_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_pytest_gpu.txt; tail -4 gpurun_out/r05_pytest_gpu.txt
bash tools/ab_trees.sh 2 > gpurun_out/r05_ab_trees.txt 2>&1; cat gpurun_out/r05_ab_trees.txt
for i in 1 2; do for f in 1 0; do
MTE_OVERLAP_SHORTCUT=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap_shortcut=$f', d['value'], d['ms_per_step'])"
done; done
