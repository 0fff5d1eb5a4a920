"""development aid: the kernels around the largest GPU-idle gaps of a rocprofv3 kernel_trace.csv (queue, start, duration)."""
import csv, sys
path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"),
                     r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]))
rows.sort()
rows = rows[int(len(rows) * 0.4):]
gaps, end = [], rows[0][1]
for i in range(1, len(rows)):
    if rows[i][0] > end:
        gaps.append((rows[i][0] - end, i))
    end = max(end, rows[i][1])
t0 = rows[0][0]
for g, i in sorted(gaps, reverse=True)[:n]:
    print("---- gap %.1f us before row %d" % (g * 1e-3, i))
    for s, e, q, nm in rows[max(0, i - 14):i + 14]:
        print("  q%-3s start %10.1f us  dur %8.1f us  %s" % (q, (s - t0) * 1e-3, (e - s) * 1e-3, nm))
