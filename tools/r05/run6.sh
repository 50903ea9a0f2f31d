#!/bin/bash
# GPU box, round 5 call 6: the hybrid build (trace kernels in a translation unit of their own) on plt_path; the new test of the kernel forms; A/B.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5f; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
H=$PWD/wave_tracer_amd/_v/libwtgpu_hybA.so
WTGPU_LIB=$H WTGPU_SORTED_INTERACT=0 WTGPU_STAGED_CONNECT=0 timeout 120 python -m pytest "tests/test_emitters.py::test_directional_emitter_gpu_parity[sunlit_path-8-kw2]" tests/test_gpu_path.py -x -q --timeout 60 > $O/path_hybA.log 2>&1; echo "hybA path rc=$? $(tail -1 $O/path_hybA.log)"
timeout 300 python -m pytest tests/test_gpu_render.py -q -x -k "material_sorted" --timeout 200 > $O/forms.log 2>&1; echo "forms rc=$?"; tail -4 $O/forms.log
AB_STEPS=10 bash tools/ab_run.sh r5f \
  "u_default|-||" \
  "hybA_old|hybA|WTGPU_SORTED_INTERACT=0 WTGPU_STAGED_CONNECT=0|" \
  "u_default2|-||" \
  "hybA_old2|hybA|WTGPU_SORTED_INTERACT=0 WTGPU_STAGED_CONNECT=0|"
