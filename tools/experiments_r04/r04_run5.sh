#!/bin/bash
# GPU box, round 4, run 5: new interrupts / preview tests, MIS batching, register budgets of interact / connect, guided fetch on the small film
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4e
timeout 600 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "pause or preview or progressive or two_gpus or parity_small" > gpurun_out/r4e/tests.log 2>&1; tail -4 gpurun_out/r4e/tests.log
AB_STEPS=8 bash tools/ab_run.sh r4e "gss0|gss0|" "new|-|" "li3|li3|" "li2|li2|" "lc1|lc1|" "new_2spp|-|WTGPU_STATE_GB=220|--spp-per-step 2 --batch 4147200" "new_2spp_s2|-|WTGPU_STATE_GB=220 WTGPU_STREAMS=2|--spp-per-step 2 --batch 4147200" "new_3spp_s2|-|WTGPU_STATE_GB=230 WTGPU_STREAMS=2|--spp-per-step 3 --batch 6220800"
AB_STEPS=12 bash tools/ab_run.sh r4e "etoile_new|-||--scene etoile --res 720" "etoile_gss1|gss1||--scene etoile --res 720" "etoile_gss1_ns|gss1|WTGPU_SHRINK_R1=96|--scene etoile --res 720" "etoile_4spp|-||--scene etoile --res 720 --spp-per-step 4 --batch 1555200" "etoile_8spp|-||--scene etoile --res 720 --spp-per-step 8 --batch 3110400" "bidir_new|-||--scene bidir_room --res 1920" "bidir_base|base||--scene bidir_room --res 1920"
