#!/bin/bash
mkdir -p gpurun_out/r3f
run() { L=$1; shift; env "$@" timeout 300 python bench.py --steps 9 --warmup 3 --no-traffic --no-cpu-baseline 2>/dev/null > gpurun_out/r3f/$L.json; python - <<PY
import json
d=json.loads(open("gpurun_out/r3f/$L.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernel_ms_per_step_stream_summed"]
print("%-16s %6.2f %6.1f | "%("$L", d["value"], d["ms_per_step"]) + " ".join("%s %.0f"%(a.replace("k_","")[:12],b) for a,b in k.items()))
PY
}
run base X=1
run cb48 WTGPU_CONE_BUDGET=48
run cb96 WTGPU_CONE_BUDGET=96
run rb6 WTGPU_ROUND_BLOCKS=6
run rb12 WTGPU_ROUND_BLOCKS=12
run hw12 WTGPU_HEAVY_WAVES=12
run hw16 WTGPU_HEAVY_WAVES=16
run shr_r1_8 WTGPU_SHRINK_R1=8
run shr_r1_4 WTGPU_SHRINK_R1=4
run shr_f1_2 WTGPU_SHRINK_F1=2
run gridc1 WTGPU_GRID_C=1
run gridflux4 WTGPU_GRID_FLUX=4
