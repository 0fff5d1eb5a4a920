#!/bin/bash
# HBM-side traffic of every kernel family over 3 training steps: two separate PMC passes (FETCH_SIZE and WRITE_SIZE do not
# fit one pass), corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950; both counters are in KiB... see below).
mkdir -p gpurun_out; rm -rf /tmp/pmcF /tmp/pmcW
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/pmcF_run.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/pmcW_run.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_traffic.py /tmp/pmcF /tmp/pmcW 3 > gpurun_out/pmc_traffic.txt
cat gpurun_out/pmc_traffic.txt | head -60
