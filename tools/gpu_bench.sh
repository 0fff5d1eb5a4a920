#!/bin/bash
# development aid: smoke + bench + rocprofv3 kernel stats in one GPU call
mkdir -p gpurun_out
python __graft_entry__.py --smoke > gpurun_out/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.txt
tail -4 gpurun_out/smoke.txt
python bench.py --steps 5 --warmup 2 "$@" > gpurun_out/bench.txt 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.txt
tail -3 gpurun_out/bench.txt; tail -5 gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof_run.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-200
