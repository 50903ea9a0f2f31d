#!/bin/bash
# GPU box, round 4, run 16: bounding-sphere pre-filter of the wave-cooperative cone query, spheres fetched 1 / 4 / 8 batches ahead
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4p
bash tools/ab_run.sh r4p "cur|cur||" "sp1|sp1||" "sp4|sp4||" "sp8|sp8||" "cur2|cur||" "sp8b|sp8||" 2>&1 | tee gpurun_out/r4p/ab.log
for V in cur sp1 sp8; do
  echo "== $V"; WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_$V.so WTGPU_PROFILE=2 timeout 200 python bench.py --steps 3 --warmup 1 --no-traffic --no-cpu-baseline 2>&1 >/dev/null | grep -i "profile" | tee -a gpurun_out/r4p/heavy_profile.log
done
WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_sp8.so timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_traversal.py -q -x -k "image_parity or committed_golden or cornell_dense or cone_traversal or whole_region or full_size_properties" 2>&1 | tail -3 | tee gpurun_out/r4p/tests.log
