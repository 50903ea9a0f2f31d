#!/bin/bash
# GPU box, round 4: clock breakdown and batch counts of the cooperative cone query inside k_trace_heavy (WTGPU_COOP_PROF builds: with / without the sphere filter)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4o
for V in cprof0 cprof; do
  echo "== $V"; WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_$V.so WTGPU_PROFILE=2 timeout 200 python bench.py --steps 3 --warmup 1 --no-traffic --no-cpu-baseline 2>&1 >/dev/null | grep -i "coop prof" | tee -a gpurun_out/r4o/coop_profile.log
done
