#!/bin/bash
mkdir -p gpurun_out/r3y
for s in 3 4; do
 for sc in etoile bidir_room; do
  WTGPU_STREAMS=$s timeout 300 python bench.py --scene $sc --steps 8 --warmup 2 --no-traffic --no-cpu-baseline 2>/dev/null > gpurun_out/r3y/${sc}_s$s.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3y/${sc}_s$s.json").read().strip().splitlines()[-1]); print("$sc streams $s", round(d["value"],2), round(d["ms_per_step"],1))
PY
 done
done
