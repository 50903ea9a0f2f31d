#!/bin/bash
# GPU box, round 4, run 25: the resumption of the Fraunhofer interaction step after pass C by one lane per walk (k_interact_commit) instead of by thread 0 of pass C's wavefront
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4ab
WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_dc.so timeout 900 python -m pytest tests/test_gpu_render.py -q -x -k "image_parity or committed_golden or cornell_dense or double_slits or full_size_properties or full_size_bidir or two_parts" 2>&1 | tail -3 | tee gpurun_out/r4ab/tests.log
AB_STEPS=10 bash tools/ab_run.sh r4ab "b_dc1|dc||--scene bidir_room --res 1920" "b_dc0|dc|WTGPU_DEFERRED_COMMIT=0|--scene bidir_room --res 1920" "c_dc1|dc||" "c_dc0|dc|WTGPU_DEFERRED_COMMIT=0|" "b_dc1b|dc||--scene bidir_room --res 1920" "b_dc0b|dc|WTGPU_DEFERRED_COMMIT=0|--scene bidir_room --res 1920" 2>&1 | tee gpurun_out/r4ab/ab.log
