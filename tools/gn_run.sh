python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15
echo "=== loss bench"; python tools/loss_bench.py
echo "=== bench graph"; python bench.py --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/r2_b2.json 2> gpurun_out/r2_b2.err; tail -c 400 gpurun_out/r2_b2.json; tail -3 gpurun_out/r2_b2.err
echo "=== bench eager"; MTE_BENCH_EAGER=1 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/r2_b2e.json 2> gpurun_out/r2_b2e.err; tail -c 400 gpurun_out/r2_b2e.json
