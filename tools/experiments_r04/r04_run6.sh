#!/bin/bash
# GPU box, round 4, run 6: the new bench defaults (steps that fill one batch), grid knobs, guided fetch on small films, new GPU tests
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4f
timeout 600 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "function_textures or xml_scene" > gpurun_out/r4f/tests.log 2>&1; tail -3 gpurun_out/r4f/tests.log
AB_STEPS=8 bash tools/ab_run.sh r4f "default|-|" "li1|li1|" "sh8|-|WTGPU_SHRINK_R1=8 WTGPU_SHRINK_R2=16" "sh4|-|WTGPU_SHRINK_R1=4 WTGPU_SHRINK_R2=10" "rb6|-|WTGPU_ROUND_BLOCKS=6" "rb12|-|WTGPU_ROUND_BLOCKS=12" "hw6|-|WTGPU_HEAVY_WAVES=6" "hw12|-|WTGPU_HEAVY_WAVES=12" "gss1|gss1|" "s4|-|WTGPU_STREAMS=4"
AB_STEPS=8 bash tools/ab_run.sh r4f "etoile720|-||--scene etoile --res 720" "etoile720_gss1|gss1||--scene etoile --res 720" "etoile720_16|-||--scene etoile --res 720 --spp-per-step 16" "etoile1440|-||--scene etoile --res 1440" "bidir1920|-||--scene bidir_room --res 1920"
