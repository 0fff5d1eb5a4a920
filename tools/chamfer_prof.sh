#!/bin/bash
# development aid: bench + rocprofv3 kernel stats of the on-device chamfer metrics
mkdir -p gpurun_out; rm -rf /tmp/cprof
python tools/chamfer_bench.py 4 --cpu 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof -- python3 $GRAFT_REPO_ROOT/tools/chamfer_bench.py 4 > $GRAFT_REPO_ROOT/gpurun_out/chamfer_prof_run.txt 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
f=$(find /tmp/cprof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/chamfer_kernel_stats.csv; head -6 "$f" | cut -c1-220; else echo "no stats file"; fi
