#!/bin/bash
mkdir -p gpurun_out/r3c
run() { L=$1; shift; env "$@" timeout 200 python bench.py --steps 12 --warmup 3 --no-traffic --no-cpu-baseline $EXTRA 2>/dev/null > gpurun_out/r3c/$L.json; python - <<PY
import json
d=json.loads(open("gpurun_out/r3c/$L.json").read().strip().splitlines()[-1]); print("%-24s"%"$L", round(d["value"],2), round(d["ms_per_step"],1))
PY
}
EXTRA="" run base X=1
EXTRA="--batch 6220800" run batch_x3 WTGPU_STATE_GB=200
EXTRA="--batch 4147200" run batch_x2 WTGPU_STATE_GB=200
EXTRA="--batch 4147200" run batch_x2_s2 WTGPU_STATE_GB=200 WTGPU_STREAMS=2
EXTRA="--batch 8294400" run batch_x4_s4 WTGPU_STATE_GB=250 WTGPU_STREAMS=4
