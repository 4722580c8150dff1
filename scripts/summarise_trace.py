"""Prints the kernel timeline (start offset, duration, gap to previous kernel end) of the last N kernels of a
rocprofv3 kernel_trace.csv:  python scripts/summarise_trace.py <csv> <first-kernel-substring> [max_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2]
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i0 = starts[-1]
# the call starts a few copy kernels earlier; include up to 3 preceding kernels
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
tot_busy = 0
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for r in rows[i0:i0 + limit]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    tot_busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  {r['Kernel_Name'][:70]}")
    prev_end = e
print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {tot_busy / 1e3:.1f} us")
