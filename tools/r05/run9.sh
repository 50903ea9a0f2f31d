#!/bin/bash
# GPU box, round 5 call 9: pass A with wave-cooperative record transfers (k_interact_coop): does it render the same, is it faster.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5i; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
WTGPU_LIB=$PWD/wave_tracer_amd/_v/libwtgpu_coop.so WTGPU_COOP_IO=1 timeout 300 python -m pytest tests/test_gpu_render.py -q -x --timeout 150 -k "parity_small or parity_scenes or polarimetric or wrappers" > $O/tests_coop.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests_coop.log
AB_STEPS=10 bash tools/ab_run.sh r5i "split_base|coop|WTGPU_COOP_IO=0|" "coop_lb3|coop|WTGPU_COOP_IO=1|" "coop_lb2|coop2|WTGPU_COOP_IO=1|" "split_base2|coop|WTGPU_COOP_IO=0|" "coop_lb3_2|coop|WTGPU_COOP_IO=1|"
