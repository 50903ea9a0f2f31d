#!/bin/bash
# GPU box, round 4, run 24: wave-per-walk grids for the plt_path workload (k_path_fsd / k_path_nee / k_path_edges take n_cu x WTGPU_HEAVY_WAVES wavefronts)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4aa
AB_STEPS=8 bash tools/ab_run.sh r4aa "e_hw8|-||--scene etoile --res 720" "e_hw12|-|WTGPU_HEAVY_WAVES=12|--scene etoile --res 720" "e_hw16|-|WTGPU_HEAVY_WAVES=16|--scene etoile --res 720" "e_hw24|-|WTGPU_HEAVY_WAVES=24|--scene etoile --res 720" "e_hw8b|-||--scene etoile --res 720" 2>&1 | tee gpurun_out/r4aa/ab.log
