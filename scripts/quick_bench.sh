#!/bin/bash
# iteration / step timings of the three configs with bench.py's own harness, short form:  bash scripts/quick_bench.sh <tag>
TAG=${1:-q}
for cfg in mnist ucf odernn; do
  python3 bench.py --config $cfg --steps 50 --warmup 10 --train-steps 20 --no-cpu-baseline --no-live-traffic > gpurun_out/${TAG}_$cfg.json 2> gpurun_out/${TAG}_$cfg.err
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_$cfg.json").read().strip().splitlines()[-1])
it = d["iteration"]
print("$cfg", {k: d[k] for k in ("value", "d_step_ms", "g_step_ms", "iteration_ms_eager", "iteration_ms_graph")}, "frac", it["frac"], "launches", it["libgode_launches"])
print("   ", {c: (v["launches"], v["ms"]) for c, v in list(it.values())[-1].items()})
PY
done
