#!/bin/bash
mkdir -p gpurun_out/r3z
for p in 1048576 2097152 16777216; do
for s in 3 4; do
 for sc in cornell_box etoile bidir_room; do
  HSA_KERNARG_POOL_SIZE=$p WTGPU_STREAMS=$s timeout 300 python bench.py --scene $sc --steps 8 --warmup 2 --no-traffic --no-cpu-baseline 2>/dev/null > gpurun_out/r3z/${sc}_s${s}_p$p.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3z/${sc}_s${s}_p$p.json").read().strip().splitlines()[-1]); print("pool $p streams $s %-12s"%"$sc", round(d["value"],2), round(d["ms_per_step"],1))
PY
 done
done
done
