#!/bin/bash
# GPU box, round 4, run 14: k_trace_heavy with the candidates' vertices fetched 1 / 2 / 4 batches ahead of the filter tests
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4n
bash tools/ab_run.sh r4n "pf1|pf1||" "pf2|pf2||" "pf4|pf4||" "pf1b|pf1||" "pf2b|pf2||" 2>&1 | tee gpurun_out/r4n/ab.log
for V in pf1 pf2 pf4; do
  echo "== $V"; WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_$V.so WTGPU_PROFILE=2 timeout 200 python bench.py --steps 3 --warmup 1 --no-traffic --no-cpu-baseline 2>&1 >/dev/null | grep -i "profile" | tee -a gpurun_out/r4n/heavy_profile.log
done
WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_pf2.so timeout 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_traversal.py -q -x -k "image_parity_small or committed_golden or cornell_dense or cone_traversal" 2>&1 | tail -3 | tee gpurun_out/r4n/tests.log
