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
bursts = collections.Counter()
for q, rs in byq.items():
    i = 0
    while i < len(rs):
        if "copyBuffer" in rs[i]["Kernel_Name"]:
            j = i
            while j < len(rs) and "copyBuffer" in rs[j]["Kernel_Name"]:
                j += 1
            prev = rs[i - 1]["Kernel_Name"][:70] if i else "-"
            nxt = rs[j]["Kernel_Name"][:70] if j < len(rs) else "-"
            bursts[(q, prev, j - i, nxt)] += 1
            i = j
        else:
            i += 1
for (q, p, n, nx), c in sorted(bursts.items(), key=lambda kv: -kv[1] * kv[0][2])[:40]:
    print("queue %s  %4.1f x/step  burst of %2d  after %-72s before %s" % (q, c / 7.0, n, p, nx))
PY
