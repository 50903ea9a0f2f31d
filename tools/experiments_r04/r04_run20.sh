#!/bin/bash
# GPU box, round 4, run 20: persistent-grid size of pass C (wave per walk; bidir_room has 0.15 pass-C items per sample at ~80 K clocks each)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4u
AB_STEPS=10 bash tools/ab_run.sh r4u "b_c2|-||--scene bidir_room --res 1920" "b_c1|-|WTGPU_GRID_C=1|--scene bidir_room --res 1920" "b_c1_hw12|-|WTGPU_GRID_C=1 WTGPU_HEAVY_WAVES=12|--scene bidir_room --res 1920" "b_c1_hw16|-|WTGPU_GRID_C=1 WTGPU_HEAVY_WAVES=16|--scene bidir_room --res 1920" "c_c2|-||" "c_c1|-|WTGPU_GRID_C=1|" "b_c2b|-||--scene bidir_room --res 1920" 2>&1 | tee gpurun_out/r4u/ab.log
