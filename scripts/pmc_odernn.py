"""VALU issue accounting of the ODE-RNN kernels from a rocprofv3 counter CSV collected with
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -- python3 scripts/bench_odernn.py 32
(counter values come out x16 on this stack for SQ counters collected per XCD -- scripts/pmc_clock.py; SQ_WAVES of a
one-workgroup launch of 8 waves calibrates the factor here).  Prints vector instructions per wave and the share of the
kernel's cycles that two waves per SIMD need just to ISSUE them (4 cycles per wave64 instruction on a 16-lane SIMD)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    if "odernn" in r["Kernel_Name"] and "rows" not in r["Kernel_Name"]:
        k = int(r["Dispatch_Id"])
        d.setdefault(k, {"name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "grid": r["Grid_Size"],
                         "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[k][r["Counter_Name"]] = float(r["Counter_Value"])
seen = {}
for k, v in d.items():
    key = (v["name"], v["grid"])
    seen[key] = v          # last dispatch of each (kernel, grid)
for (name, grid), v in seen.items():
    waves_nominal = int(grid) // 64
    scale = v.get("SQ_WAVES", 0) / max(waves_nominal, 1)          # counter inflation factor on this stack
    if scale <= 0:
        continue
    valu = v.get("SQ_INSTS_VALU", 0) / scale / waves_nominal
    lds = v.get("SQ_INSTS_LDS", 0) / scale / waves_nominal
    salu = v.get("SQ_INSTS_SALU", 0) / scale / waves_nominal
    cycles = v["dur"] * 2.43          # ns x GHz (measured clock of these launches: scripts/exp/odernn_stamps.sh)
    issue = valu * 4 * 2              # two waves per SIMD
    print(f"{name:28s} grid={grid:>6s} dur={v['dur']/1e3:8.1f} us  VALU/wave={valu:9.0f}  LDS/wave={lds:7.0f}  SALU/wave={salu:8.0f}  "
          f"issue cycles (2 waves/SIMD) {issue:9.0f} of {cycles:9.0f} = {issue/cycles:4.2f}   (counter scale x{scale:.0f})")
