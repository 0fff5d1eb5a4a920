"""development aid: per (kernel, grid) summary of a rocprofv3 kernel_trace.csv — calls, total and mean duration."""
import csv
import sys
from collections import defaultdict

path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("LDS_Block_Size", ""))
        d = agg[key]
        d[0] += 1
        d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
tot = sum(v[1] for v in agg.values())
print("total kernel time / step: %.2f ms" % (tot / steps * 1e-3))
# the same sum over whole steps only (adam_dev_kernel closes a step): start-up copies, weight packing and the first-call allocations stay outside
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "adam_dev_kernel" in r["Kernel_Name"]))
ends = sorted(e for s_, e, a in rows if a)
if len(ends) >= 2:
    inside = sum(e - s_ for s_, e, a in rows if ends[0] < e <= ends[-1])
    print("kernel time / step between the first and the last adam_dev_kernel (%d whole steps): %.2f ms" % (len(ends) - 1, inside * 1e-6 / (len(ends) - 1)))
for (name, grid, lds), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 70]:
    print("%8.1f us/step %6.1f calls/step %9.1f us/call  grid=%-9s lds=%-6s %s" % (us / steps, n / steps, us / n, grid, lds, name))

# ---- GPU occupancy over time: union of kernel intervals vs wall span (idle = launch gaps / host-bound phases)
iv = []
with open(path) as f:
    for r in csv.DictReader(f):
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                   r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]))
iv.sort()
# drop the warm-up/initialisation part: keep the last 60 % of kernels
iv = iv[int(len(iv) * 0.4):]
busy, cur_s, cur_e, gaps, where, last = 0, iv[0][0], iv[0][1], [], [], iv[0][2]
for s_, e_, nm in iv[1:]:
    if s_ > cur_e:
        busy += cur_e - cur_s
        gaps.append(s_ - cur_e)
        where.append((s_ - cur_e, last, nm))
        cur_s, cur_e, last = s_, e_, nm
    else:
        if e_ > cur_e:
            cur_e, last = e_, nm
busy += cur_e - cur_s
span = cur_e - iv[0][0]
print("steady-state window: span %.2f ms, busy %.2f ms (%.1f %%), idle %.2f ms in %d gaps" % (span * 1e-6, busy * 1e-6, 100.0 * busy / span, (span - busy) * 1e-6, len(gaps)))
gaps.sort(reverse=True)
print("largest gaps (us):", [round(g * 1e-3, 1) for g in gaps[:12]])
for g_, a_, b_ in sorted(where, reverse=True)[:8]:
    print("   gap %7.1f us between %-48s and %s" % (g_ * 1e-3, a_, b_))
import bisect
for lim in (2, 5, 10, 20, 50):
    sel = [g for g in gaps if g * 1e-3 <= lim]
    print("  gaps <= %3d us: %6d, total %.2f ms" % (lim, len(sel), sum(sel) * 1e-6))

# ---- per HIP stream (HSA queue): which kernels sit on the main chain and which on the side stream
perq = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:40]
        d = perq[r.get("Queue_Id", "?")][name]
        d[0] += 1
        d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
for q, ks in sorted(perq.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
    tot = sum(v[1] for v in ks.values())
    print("queue %s: %.2f ms/step in %d launches/step" % (q, tot / steps * 1e-3, sum(v[0] for v in ks.values()) / steps))
    for name, (n, us) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:28]:
        print("    %8.1f us/step %6.1f calls/step  %s" % (us / steps, n / steps, name))
