#!/bin/bash
mkdir -p gpurun_out/r3e
run() { L=$1; shift; timeout 300 python bench.py --steps 8 --warmup 2 --no-traffic --no-cpu-baseline "$@" 2>/dev/null > gpurun_out/r3e/$L.json; python - <<PY
import json
d=json.loads(open("gpurun_out/r3e/$L.json").read().strip().splitlines()[-1]); print("%-24s"%"$L", round(d["value"],2), round(d["ms_per_step"],1), d["config"]["samples_per_step"])
PY
}
export WTGPU_STATE_GB=250
run spp2_b2_s3 --spp-per-step 2 --batch 4147200
WTGPU_STREAMS=2 run spp2_b2_s2 --spp-per-step 2 --batch 4147200
WTGPU_STREAMS=2 run spp4_b4_s2 --spp-per-step 4 --batch 8294400
WTGPU_STREAMS=2 run spp1_b1_s2
WTGPU_STREAMS=4 run spp1_b1_s4
