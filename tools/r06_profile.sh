#!/bin/bash
# (bench.py runs warm-up + timed + 3 unthrottled steps: 1 + 3 + 3 = 7 steps in a trace, 1 + 2 + 3 = 6 in a PMC pass)
# Round-6 evidence in ONE gpurun call (program directly after `--`; --pmc passes separate from --kernel-trace):
#   1. overlapped (default two-stream schedule) kernel trace + stats     -> gpurun_out/r06_<tag>_train_T8_kernel_stats.csv / _trace_summary.txt
#   2. SERIAL kernel trace + stats (MTE_NO_SIDE_STREAM=1 in the environment) -> ..._serial_kernel_stats.csv  (bench.py's roofline.frac is the serial figure)
#   3. per-shape conv table (bench.py --dump-conv)                       -> ..._conv_table.txt
#   4. FETCH_SIZE / WRITE_SIZE PMC passes                                -> ..._pmc_traffic_T8.txt
#   5. the bench line itself                                             -> ..._bench.json
tag=${1:-v1}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out; rm -rf /tmp/trO /tmp/trS /tmp/pmcF /tmp/pmcW
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trO -- python3 $B > $out/r06_${tag}_trace_run.txt 2>&1
export MTE_NO_SIDE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trS -- python3 $B > $out/r06_${tag}_trace_serial_run.txt 2>&1
unset MTE_NO_SIDE_STREAM
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/r06_${tag}_pmcF_run.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/r06_${tag}_pmcW_run.txt 2>&1
cd $GRAFT_REPO_ROOT
cp "$(find /tmp/trO -name '*kernel_stats.csv' | head -1)" $out/r06_${tag}_train_T8_kernel_stats.csv
cp "$(find /tmp/trS -name '*kernel_stats.csv' | head -1)" $out/r06_${tag}_train_T8_serial_kernel_stats.csv
python tools/trace_summary.py "$(find /tmp/trO -name '*kernel_trace.csv' | head -1)" 7 90 > $out/r06_${tag}_train_T8_trace_summary.txt
python tools/trace_summary.py "$(find /tmp/trS -name '*kernel_trace.csv' | head -1)" 7 60 > $out/r06_${tag}_train_T8_serial_trace_summary.txt
python3 tools/pmc_traffic.py /tmp/pmcF /tmp/pmcW 6 $out/r06_${tag}_pmc_traffic.json "profiles/r06_${tag}_pmc_traffic_T8.txt (round 6; rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB)" > $out/r06_${tag}_pmc_traffic_T8.txt
python bench.py --steps 20 --warmup 5 --dump-conv $out/r06_${tag}_conv_table.txt > $out/r06_${tag}_train_T8_bench.json 2> $out/r06_${tag}_bench.err
# stamp the traffic figures with the API launch counts of the bench run they belong to (bench.py quotes `traffic` only while its own run counts the same)
python3 - $out/r06_${tag}_pmc_traffic.json $out/r06_${tag}_train_T8_bench.json <<'PY'
import json, sys
t = json.load(open(sys.argv[1])); b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
t["conv_family_api_launches_per_step"] = b["roofline"]["launches_per_step"]
t["gn_family_api_launches_per_step"] = b["roofline_hbm"]["launches_per_step"]
json.dump(t, open(sys.argv[1], "w"), indent=1)
PY
head -3 $out/r06_${tag}_train_T8_serial_trace_summary.txt; tail -c 600 $out/r06_${tag}_train_T8_bench.json
