#!/bin/bash
mkdir -p gpurun_out/r3d
run() { L=$1; shift; timeout 300 python bench.py --steps 12 --warmup 3 --no-traffic --no-cpu-baseline "$@" 2>/dev/null > gpurun_out/r3d/$L.json; python - <<PY
import json
d=json.loads(open("gpurun_out/r3d/$L.json").read().strip().splitlines()[-1]); print("%-24s"%"$L", round(d["value"],2), round(d["ms_per_step"],1))
PY
}
run cornell
run etoile --scene etoile
run etoile720 --scene etoile --res 720
run bidir1920 --scene bidir_room --res 1920
run cornell_spp2 --spp-per-step 2
timeout 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_path.py -m gpu -x -q -k "not bias" 2>&1 | tail -2
