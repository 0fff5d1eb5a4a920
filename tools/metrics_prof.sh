#!/bin/bash
# development aid: rocprofv3 kernel stats of the on-device validation metrics (tools/metrics_bench.py)
mkdir -p gpurun_out; rm -rf /tmp/mprof
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mprof -- python3 $GRAFT_REPO_ROOT/tools/metrics_bench.py 4 > $GRAFT_REPO_ROOT/gpurun_out/metrics_prof_run.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/mprof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/metrics_kernel_stats.csv; head -14 "$f"; else echo "no stats file"; find /tmp/mprof | head; fi
tail -1 gpurun_out/metrics_prof_run.txt
