"""Last training iteration of a rocprofv3 kernel_trace.csv: span, sum of kernel durations, union busy time (overlap
between streams shows as sum > union), per-queue busy.  python scripts/iter_overlap.py <csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups, prev = [], False
for i, r in enumerate(rows):
    a = "adam_" in r["Kernel_Name"]
    if a and not prev:
        groups.append(i)
    prev = a
def end_of(gi):
    j = groups[gi]
    while j < len(rows) and "adam_" in rows[j]["Kernel_Name"]:
        j += 1
    return j
# an iteration has 5 adam launches; with the side stream they are no longer time-ordered like the program, so take
# the window between the G-step adam of iteration n-1 and of iteration n by counting 5 groups
i0, i1 = end_of(-6), end_of(-1)
sel = rows[i0:i1]
t0 = int(sel[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in sel)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
tot = sum(e - s for s, e in iv)
q = collections.defaultdict(float)
for r in sel:
    q[r.get("Queue_Id", "?")] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print(f"kernels {len(sel)} span {(t1 - t0) / 1e6:.3f} ms  sum {tot / 1e6:.3f} ms  union {union / 1e6:.3f} ms  idle {(t1 - t0 - union) / 1e6:.3f} ms")
print("per queue busy ms:", dict(q))
