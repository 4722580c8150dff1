"""Turns two rocprofv3 counter-collection CSVs (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass of
`python scripts/bench_igemm.py --only dec --noxf --reps 2`) into profiles/<name>.json: HBM-side bytes per launch of the
three decoder GEMMs, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950 for wide coalesced reads;
counter unit KB).   python scripts/pmc_traffic.py <fetch.csv> <write.csv> <out.json>"""
import csv, json, sys, collections


def per_dispatch(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "igemm_fast" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(r["Kernel_Name"].split("(")[0].replace("void ", ""), float(r["Counter_Value"])) for r in rows]


fetch = per_dispatch(sys.argv[1], "FETCH_SIZE")
write = per_dispatch(sys.argv[2], "WRITE_SIZE")
# bench order: 3 forward layers (DGRAD direction) then 3 backward-data layers, 3 warm-up + reps launches each
names = ["convT1 512->256 k4s2", "convT2 256->128 k4s2", "convT3 128->64 k4s2"]
alg = [58.7e6, 102.7e6, 201.8e6]
per = len(fetch) // 6
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python scripts/bench_igemm.py "
                 "--only dec --noxf --reps 2; MI355X, round 1, LDS-DMA build",
       "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B for wide coalesced reads, MI355X_MICROARCH.md "
                     "section HBM); WRITE_SIZE as is; counter unit KB; Infinity-Cache hits are counted, so this is L2-miss "
                     "(fabric) traffic, an upper bound on HBM bytes",
       "per_launch": {}}
for i, n in enumerate(names):
    f = fetch[i * per + per - 1]; w = write[i * per + per - 1]
    out["per_launch"][n] = {"kernel": f[0], "fetch_raw_kb": f[1], "write_kb": w[1],
                            "traffic_bytes": int(2 * f[1] * 1024 + w[1] * 1024), "algorithmic_bytes": int(alg[i])}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["per_launch"], indent=1))
