#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5l; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
WTGPU_STREAMS=1 WTGPU_LIB=$PWD/wave_tracer_amd/_v/libwtgpu_watch.so timeout 60 python tools/r05/watch_path.py sunlit_path 32 8 10 > $O/watch.log 2>&1; echo "watch rc=$?"; grep -v "amdgpu.ids" $O/watch.log | grep -v "block  [0-7]:" | head -20
