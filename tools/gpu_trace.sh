#!/bin/bash
# development aid: rocprofv3 kernel trace of 4 training steps, summarised per (kernel, grid)
mkdir -p gpurun_out; rm -rf gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > $GRAFT_REPO_ROOT/gpurun_out/trace_run.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py "$f" 4 90 > gpurun_out/trace_summary.txt
head -100 gpurun_out/trace_summary.txt
