#!/bin/bash
# GPU box, round 5 call 3: the plt_path hang with variants of the path translation unit; the plt_bdpt part of the GPU suite on the new default.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
T="tests/test_emitters.py::test_directional_emitter_gpu_parity[sunlit_path-8-kw2]"
for L in pdbg pO2 plb1; do
  export WTGPU_LIB=$PWD/wave_tracer_amd/_v/libwtgpu_$L.so
  timeout 45 python -m pytest "$T" -x -q -s > $O/path_$L.log 2>&1; echo "$L rc=$? $(tail -1 $O/path_$L.log)"
done
grep -h "k_path_fsd\]" $O/path_pdbg.log | head -12
unset WTGPU_LIB
timeout 500 python -m pytest tests -m gpu -q -x -k "not path and not etoile" --timeout 200 > $O/bdpt_tests.log 2>&1; echo "bdpt tests rc=$?"; tail -6 $O/bdpt_tests.log
