#!/bin/bash
# GPU box, round 4, run 22: persistent-grid size of pass B (bidir_room queues 0.7 walks per sample for it; the default grid is a quarter of the round's)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4y
AB_STEPS=10 bash tools/ab_run.sh r4y "b_b4|-||--scene bidir_room --res 1920" "b_b2|-|WTGPU_GRID_B=2|--scene bidir_room --res 1920" "b_b1|-|WTGPU_GRID_B=1|--scene bidir_room --res 1920" "c_b4|-||" "c_b2|-|WTGPU_GRID_B=2|" "c_b1|-|WTGPU_GRID_B=1|" "b_b4b|-||--scene bidir_room --res 1920" "b_b1b|-|WTGPU_GRID_B=1|--scene bidir_room --res 1920" 2>&1 | tee gpurun_out/r4y/ab.log
