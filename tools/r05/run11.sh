#!/bin/bash
# GPU box, round 5 call 11: where k_path_fsd of the split build spins (tools/r05/watch_path.py); the GPU suite and the bench line on the shipped
# build with the new Gaussian-over-triangle quadrature.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5k; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
WTGPU_STREAMS=1 WTGPU_LIB=$PWD/wave_tracer_amd/_v/libwtgpu_watch.so timeout 60 python tools/r05/watch_path.py sunlit_path 32 8 12 > $O/watch.log 2>&1; echo "watch rc=$?"; grep -v "amdgpu.ids" $O/watch.log | head -40
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout 200 python bench.py --no-cpu-baseline --no-traffic > $O/bench.json 2>/dev/null; cut -c1-200 $O/bench.json
