#!/bin/bash
# GPU box, round 4, run 13: four waves per SIMD for the trace kernel (after the FLAT fix), and where k_trace_heavy's time goes (WTGPU_PROFILE=2)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4m
bash tools/ab_run.sh r4m "cur|cur||" "lt4|lt4||" "lt4s16|lt4s16||" "lt4s16rb10|lt4s16|WTGPU_ROUND_BLOCKS=10|" "cur2|cur||" 2>&1 | tee gpurun_out/r4m/ab.log
WTGPU_PROFILE=2 timeout 200 python bench.py --steps 3 --warmup 1 --no-traffic --no-cpu-baseline 2>&1 >/dev/null | grep -i "profile" | tee gpurun_out/r4m/heavy_profile.log
