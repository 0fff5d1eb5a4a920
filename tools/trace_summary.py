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
for (name, grid, lds), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 70]:
    print("%8.1f us/step %6.1f calls/step %9.1f us/call  grid=%-9s lds=%-6s %s" % (us / steps, n / steps, us / n, grid, lds, name))
