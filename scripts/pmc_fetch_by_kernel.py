"""Prints FETCH_SIZE (raw KB; x2 on gfx950 for wide coalesced reads) of the last launch of each run of consecutive
igemm_fast launches in a rocprofv3 counter-collection CSV.   python scripts/pmc_fetch_by_kernel.py <csv> [COUNTER]"""
import csv, sys
ctr = sys.argv[2] if len(sys.argv) > 2 else "FETCH_SIZE"
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == ctr]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
prev = None
runs = []
for r in rows:
    fast = "igemm_fast" in r["Kernel_Name"] or "wgrad_fast" in r["Kernel_Name"]
    if not fast:
        continue
    key = (r["Kernel_Name"], r["Grid_Size"])
    if runs and runs[-1][0] == key and int(r["Dispatch_Id"]) - runs[-1][2] <= 2:
        runs[-1] = (key, float(r["Counter_Value"]), int(r["Dispatch_Id"]))
    else:
        runs.append((key, float(r["Counter_Value"]), int(r["Dispatch_Id"])))
for key, v, _ in runs:
    print(f"{v/1024:9.1f} MB raw  grid={key[1]:>8s}  {key[0].split('(')[0][:60]}")
