#!/bin/bash
# GPU box: the split build after the lane-0 broadcast fix, on the plt_path tests that did not terminate with it (runs r5a-r5f); then the shipped
# (unity) build: whole GPU suite, smoke, bench.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5m; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
WTGPU_LIB=$PWD/wave_tracer_amd/_v/libwtgpu_splitfix.so timeout 400 python -m pytest -q --timeout 120 -x tests/test_gpu_path.py "tests/test_emitters.py" tests/test_gpu_render.py -m gpu -k "path or etoile or two_parts or sunlit" > $O/split_path.log 2>&1; echo "split path rc=$?"; tail -3 $O/split_path.log
timeout 700 python -m pytest tests -m gpu -q --timeout 200 > $O/tests.log 2>&1; echo "unity tests rc=$?"; tail -3 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py --no-cpu-baseline --no-traffic > $O/bench.json 2>/dev/null; cut -c1-200 $O/bench.json
WTGPU_LIB=$PWD/wave_tracer_amd/_v/libwtgpu_splitfix.so timeout 600 python -m pytest tests -m gpu -q --timeout 200 > $O/split_all.log 2>&1; echo "split all rc=$?"; tail -3 $O/split_all.log
WTGPU_LIB=$PWD/wave_tracer_amd/_v/libwtgpu_splitfix.so timeout 300 python bench.py --no-cpu-baseline --no-traffic > $O/split_bench.json 2>/dev/null; cut -c1-200 $O/split_bench.json
