"""Effective clock and MFMA-pipe occupancy per igemm_fast dispatch from a rocprofv3 counter CSV collected with
--pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU (GRBM_GUI_ACTIVE sums the 8 XCDs; the SQ counters
come out x16 on this stack -- calibrated on the ODE kernel, whose MFMA count is known exactly)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    if "igemm_fast" in r["Kernel_Name"] or "wgrad_fast" in r["Kernel_Name"]:
        k = int(r["Dispatch_Id"])
        d.setdefault(k, {"name": r["Kernel_Name"].split("(")[0][-40:], "grid": r["Grid_Size"],
                         "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[k][r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in d.items():
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    clk = cyc / v["dur"]
    mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 16 / 1024
    va = v.get("SQ_ACTIVE_INST_VALU", 0) / 16 / 1024 * 4
    print(f"{k:4d} grid={v['grid']:>8s} dur={v['dur']/1e3:8.1f}us clk={clk:5.2f}GHz mfma_busy={mf/cyc if cyc else 0:5.2f} "
          f"valu_active={va/cyc if cyc else 0:5.2f}  {v['name']}")
