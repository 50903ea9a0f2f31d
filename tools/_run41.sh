#!/bin/bash
mkdir -p gpurun_out/r3h
run() { L=$1; shift; env "$@" timeout 300 python bench.py --steps 9 --warmup 3 --no-traffic --no-cpu-baseline 2>/dev/null > gpurun_out/r3h/$L.json; python - <<PY
import json
d=json.loads(open("gpurun_out/r3h/$L.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernel_ms_per_step_stream_summed"]
print("%-16s %6.2f %6.1f | "%("$L", d["value"], d["ms_per_step"]) + " ".join("%s %.0f"%(a.replace("k_","")[:12],b) for a,b in k.items()))
PY
}
run base X=1
run rb4 WTGPU_ROUND_BLOCKS=4
run rb5 WTGPU_ROUND_BLOCKS=5
run rb4_s4 WTGPU_ROUND_BLOCKS=4 WTGPU_STREAMS=4
run hw4 WTGPU_HEAVY_WAVES=4
run rb4_hw6 WTGPU_ROUND_BLOCKS=4 WTGPU_HEAVY_WAVES=6
