#!/bin/bash
# round-5 development call: [1] GPU parity suite (report kept), [2] same-box A/B of the round-4 tree (ab_old/) against this one,
# [3] MTE_HANDOFF_FENCES=1 against the default on this tree.  usage: r05_ab.sh [reps] [pytest args ...]
reps=${1:-2}; shift
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short --timeout 900 -x "$@" > gpurun_out/r05_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_pytest_gpu.txt
tail -25 gpurun_out/r05_pytest_gpu.txt
if [ -d ab_old ]; then bash tools/ab_trees.sh $reps > gpurun_out/r05_ab_trees.txt 2>&1; cat gpurun_out/r05_ab_trees.txt; fi
: > gpurun_out/r05_ab_fences.txt
for i in $(seq 1 $reps); do
  for f in 0 1; do
    MTE_HANDOFF_FENCES=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fences=$f', d['value'], d['ms_per_step'])" >> gpurun_out/r05_ab_fences.txt
  done
done
cat gpurun_out/r05_ab_fences.txt
