import re, sys
tot_m = tot_b = 0
for l in open(sys.argv[1]):
    m = re.search(r"model\s+([\d.]+) us.*best\s+(\S+)\s+([\d.]+) us", l)
    if m:
        mo, be = float(m.group(1)), float(m.group(3))
        if l.startswith("dec L1 N=512     fprop"): continue
        tot_m += mo; tot_b += be
        flag = "  <<<" if mo > be * 1.04 else ""
        print(l[:24].strip().ljust(24), f"model {mo:7.1f} best {m.group(2):>5s} {be:7.1f}  {100*(mo/be-1):5.1f}%{flag}")
print("sum model", round(tot_m,1), "sum best", round(tot_b,1))
