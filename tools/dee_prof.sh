#!/bin/bash
# development aid: rocprofv3 kernel stats of the on-device annotation post-processing (tools/dee_bench.py)
mkdir -p gpurun_out; rm -rf /tmp/dprof
python tools/dee_bench.py 4 --cpu 2>&1 | tail -1
python tools/dee_bench.py 1 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dprof -- python3 $GRAFT_REPO_ROOT/tools/dee_bench.py 4 > $GRAFT_REPO_ROOT/gpurun_out/dee_prof_run.txt 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
f=$(find /tmp/dprof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/dee_kernel_stats.csv; head -8 "$f" | cut -c1-200; else echo "no stats file"; fi
