#!/bin/bash
# round-6 development call: full GPU parity suite (report kept) + same-box A/B of the round-5 tree (ab_old/) against this one.  usage: r06_full.sh [reps]
reps=${1:-2}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short --timeout 900 > gpurun_out/r06_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_pytest_gpu.txt
tail -12 gpurun_out/r06_pytest_gpu.txt
bash tools/ab_trees.sh $reps > gpurun_out/r06_ab_trees.txt 2>&1; cat gpurun_out/r06_ab_trees.txt
