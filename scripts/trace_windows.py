"""Summarise a rocprofv3 kernel_trace.csv in windows of N kernels: span, busy time, and the mean duration of the
three heaviest kernel names -- shows whether a slowdown is longer kernels or wider gaps.
    python scripts/trace_windows.py <csv> [N]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tot = collections.Counter()
for r in rows:
    tot[r["Kernel_Name"][:48]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
top = [k for k, _ in tot.most_common(3)]
print("windows of", N, "kernels; top:", top)
for i in range(0, len(rows) - N + 1, N):
    w = rows[i:i + N]
    span = (int(w[-1]["End_Timestamp"]) - int(w[0]["Start_Timestamp"])) / 1e6
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in w) / 1e6
    means = []
    for k in top:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in w if r["Kernel_Name"][:48] == k]
        means.append(sum(d) / max(len(d), 1) / 1e3)
    print(f"{i:7d} span {span:8.2f} ms busy {busy:8.2f} ms util {busy/span:5.2f}  " + " ".join(f"{m:7.1f}us" for m in means))
