#!/bin/bash
mkdir -p gpurun_out/r3b
run() { L=$1; shift; env "$@" timeout 200 python bench.py --steps 8 --warmup 2 --no-traffic --no-cpu-baseline $EXTRA 2>/dev/null > gpurun_out/r3b/$L.json; python - <<PY
import json
d=json.loads(open("gpurun_out/r3b/$L.json").read().strip().splitlines()[-1]); print("%-24s"%"$L", round(d["value"],2), round(d["ms_per_step"],1))
PY
}
EXTRA="" run base X=1
EXTRA="--batch 1036800" run batch_half X=1
EXTRA="--batch 691200" run batch_third X=1
EXTRA="--batch 1382400" run batch_2third X=1
EXTRA="" run rb6 WTGPU_ROUND_BLOCKS=6
EXTRA="" run rb12 WTGPU_ROUND_BLOCKS=12
EXTRA="" run hw6 WTGPU_HEAVY_WAVES=6
EXTRA="" run hw12 WTGPU_HEAVY_WAVES=12
EXTRA="" run cb48 WTGPU_CONE_BUDGET=48
EXTRA="" run cb96 WTGPU_CONE_BUDGET=96
