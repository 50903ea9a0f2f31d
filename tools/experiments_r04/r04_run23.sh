#!/bin/bash
# GPU box, round 4, run 23: larger batches for the plt_path workloads (no vertex store: ~2 KB of state per sample)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4z
AB_STEPS=8 bash tools/ab_run.sh r4z "e720_11|-||--scene etoile --res 720" "e720_16|-||--scene etoile --res 720 --spp-per-step 16" "e720_24|-||--scene etoile --res 720 --spp-per-step 24" "e720_32|-||--scene etoile --res 720 --spp-per-step 32" "e1440_3|-||--scene etoile --res 1440" "e1440_5|-||--scene etoile --res 1440 --spp-per-step 5" "e1440_8|-||--scene etoile --res 1440 --spp-per-step 8" "c_3|-||--spp-per-step 3" 2>&1 | tee gpurun_out/r4z/ab.log
