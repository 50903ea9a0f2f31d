#!/bin/bash
# GPU box, round 4, run 9: round queues ordered by (kind, direction octant, origin cell) before the trace kernels (WTGPU_SORT_ROUNDS)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4i
bash tools/ab_run.sh r4i "s0|sort|WTGPU_SORT_ROUNDS=0|" "s1|sort|WTGPU_SORT_ROUNDS=1|" "s2|sort|WTGPU_SORT_ROUNDS=2|" "s4|sort|WTGPU_SORT_ROUNDS=4|" "s6|sort|WTGPU_SORT_ROUNDS=6|" \
  "s4b3|sort|WTGPU_SORT_ROUNDS=4 WTGPU_SORT_BITS=3|" "s4b7|sort|WTGPU_SORT_ROUNDS=4 WTGPU_SORT_BITS=7|" "s4om|sort|WTGPU_SORT_ROUNDS=4 WTGPU_SORT_DIR_MAJOR=0|" "s0b|sort|WTGPU_SORT_ROUNDS=0|" 2>&1 | tee gpurun_out/r4i/ab.log
WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_sort.so WTGPU_SORT_ROUNDS=4 timeout 600 python -m pytest tests/test_gpu_render.py -q -x -k "image_parity_small or committed_golden or cornell_dense" 2>&1 | tail -3 | tee gpurun_out/r4i/tests.log
