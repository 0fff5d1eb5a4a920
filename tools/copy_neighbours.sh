#!/bin/bash
# development aid: which kernels sit before / after the __amd_rocclr_copyBuffer launches of a training step (who issues the device copies?)
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out; rm -rf /tmp/trC
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/trC -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/copy_trace_run.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$(find /tmp/trC -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
pairs = collections.Counter()
for q, rs in byq.items():
    for i, r in enumerate(rs):
        if "copyBuffer" in r["Kernel_Name"]:
            prev = rs[i - 1]["Kernel_Name"][:60] if i else "-"
            nxt = rs[i + 1]["Kernel_Name"][:60] if i + 1 < len(rs) else "-"
            pairs[(prev, nxt)] += 1
for (p, n), c in pairs.most_common(40):
    print("%5.1f/step  after %-62s before %s" % (c / 7.0, p, n))
PY
