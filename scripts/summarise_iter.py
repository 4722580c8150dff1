"""Aggregates a rocprofv3 kernel_trace.csv over the LAST training iteration (delimited by the adam_kernel bursts):
python scripts/summarise_iter.py <csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# an iteration = 5 optimiser steps; find groups of consecutive adam launches
groups = []
prev_adam = False
for i, r in enumerate(rows):
    is_adam = "adam_" in r["Kernel_Name"]
    if is_adam and not prev_adam:
        groups.append(i)
    prev_adam = is_adam
# last iteration: from after the end of adam group[-6] to end of group[-1]
def end_of(gi):
    j = groups[gi]
    while j < len(rows) and "adam_" in rows[j]["Kernel_Name"]:
        j += 1
    return j
i0, i1 = end_of(-6), end_of(-1)
sel = rows[i0:i1]
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
busy = 0
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"].split("(")[0][-60:]
    agg[n][0] += 1; agg[n][1] += d; busy += d
print(f"iteration span {span:.3f} ms, kernels {len(sel)}, busy {busy / 1e3:.3f} ms")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{d / 1e3:8.3f} ms {c:5d}x  avg {d / c:8.1f} us  {n}")
