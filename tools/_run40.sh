#!/bin/bash
mkdir -p gpurun_out/r3g
run() { L=$1; shift; env "$@" timeout 300 python bench.py --steps 9 --warmup 3 --no-traffic --no-cpu-baseline 2>/dev/null > gpurun_out/r3g/$L.json; python - <<PY
import json
d=json.loads(open("gpurun_out/r3g/$L.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernel_ms_per_step_stream_summed"]
print("%-16s %6.2f %6.1f | "%("$L", d["value"], d["ms_per_step"]) + " ".join("%s %.0f"%(a.replace("k_","")[:12],b) for a,b in k.items()))
PY
}
run base X=1
run reins1 WTGPU_BVH_REINSERT=1
run reins2 WTGPU_BVH_REINSERT=2
run reins4 WTGPU_BVH_REINSERT=4
run base2 X=1
