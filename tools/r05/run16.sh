#!/bin/bash
# GPU box, the round's last 3 minutes: the final queue grab (lane 0 behind a convergent marker).  Shipped (unity) build: the GPU suite on six
# workers sharing the GPU, bench; split build: the plt_path tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5p; mkdir -p $O
T0=$(date +%s)
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
timeout 115 python -m pytest tests -m gpu -q --timeout 100 -n 6 > $O/tests.log 2>&1; echo "unity tests rc=$?"; tail -2 $O/tests.log
echo "elapsed $(( $(date +%s) - T0 )) s"
timeout 60 python bench.py --no-cpu-baseline --no-traffic > $O/bench.json 2>/dev/null; cut -c1-200 $O/bench.json
WTGPU_LIB=$PWD/wave_tracer_amd/_v/libwtgpu_splitfix4.so timeout 30 python -m pytest -q --timeout 25 -x tests/test_gpu_path.py tests/test_emitters.py -m gpu > $O/split_path.log 2>&1; echo "split path rc=$?"; tail -1 $O/split_path.log
echo "elapsed $(( $(date +%s) - T0 )) s"
