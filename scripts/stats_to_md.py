"""rocprofv3 kernel_stats.csv -> markdown table.  python scripts/stats_to_md.py <csv> <out.md> <header text file or ->"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
head = open(sys.argv[3]).read() if len(sys.argv) > 3 and sys.argv[3] != "-" else ""
with open(sys.argv[2], "w") as f:
    f.write(head)
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |\n")
