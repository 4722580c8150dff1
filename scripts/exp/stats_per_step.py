"""rocprofv3 kernel_stats.csv -> per-step table:  python scripts/exp/stats_per_step.py <dir> <steps>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2])
tot = 0.0
for r in csv.DictReader(open(f)):
    n, t = int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3
    tot += t
    name = r["Name"][:70]
    print(f"{t / steps:8.1f} us/step  {n / steps:5.1f} calls/step  avg {t / n:7.1f} us  {name}")
print("sum per step", round(tot / steps, 1))
