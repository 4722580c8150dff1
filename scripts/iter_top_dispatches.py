"""Longest dispatches of the last iteration of a rocprofv3 kernel trace: python scripts/iter_top_dispatches.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups, prev = [], False
for i, r in enumerate(rows):
    a = "adam_" in r["Kernel_Name"]
    if a and not prev: groups.append(i)
    prev = a
def end_of(gi):
    j = groups[gi]
    while j < len(rows) and "adam_" in rows[j]["Kernel_Name"]: j += 1
    return j
sel = rows[end_of(-6):end_of(-1)]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in sorted(sel, key=lambda r: int(r["Start_Timestamp"]) - int(r["End_Timestamp"]))[:n]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{d:8.1f} us  grid {r.get('Grid_Size_X','?'):>8s}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')} wg {r.get('Workgroup_Size_X','?'):>4s}  {r['Kernel_Name'].split('(')[0][-64:]}")
