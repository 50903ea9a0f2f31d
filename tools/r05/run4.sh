#!/bin/bash
# GPU box, round 5 call 4: does ONE translation unit (the layout of rounds 1-4) run the plt_path test; the plt_bdpt suite on the chunked staged
# connections; A/B of pass A's forms and of two traversal variants.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5d; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
T="tests/test_emitters.py::test_directional_emitter_gpu_parity[sunlit_path-8-kw2]"
WTGPU_LIB=$PWD/wave_tracer_amd/_v/libwtgpu_unity.so timeout 45 python -m pytest "$T" -x -q > $O/path_unity.log 2>&1; echo "unity rc=$? $(tail -1 $O/path_unity.log)"
timeout 400 python -m pytest tests -m gpu -q -x -k "not path and not etoile" --timeout 200 > $O/bdpt_tests.log 2>&1; echo "bdpt tests rc=$?"; tail -6 $O/bdpt_tests.log
AB_STEPS=6 bash tools/ab_run.sh r5d \
  "old|-|WTGPU_SORTED_INTERACT=0 WTGPU_STAGED_CONNECT=0|" \
  "fused_staged|-||" \
  "fused_only|-|WTGPU_STAGED_CONNECT=0|" \
  "lsph_old|lsph|WTGPU_SORTED_INTERACT=0 WTGPU_STAGED_CONNECT=0|" \
  "fpct_old|fpct|WTGPU_SORTED_INTERACT=0 WTGPU_STAGED_CONNECT=0|"
