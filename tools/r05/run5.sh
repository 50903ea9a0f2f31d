#!/bin/bash
# GPU box, round 5 call 5: the whole GPU suite on the single-translation-unit build; A/B of pass A's and the connections' forms on it.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5e; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
timeout 600 python -m pytest tests -m gpu -q -x --timeout 150 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
AB_STEPS=8 bash tools/ab_run.sh r5e \
  "u_old|-|WTGPU_SORTED_INTERACT=0 WTGPU_STAGED_CONNECT=0|" \
  "u_s1_staged|-|WTGPU_SORTED_INTERACT=1|" \
  "u_s2_staged|-||" \
  "u_s1_only|-|WTGPU_SORTED_INTERACT=1 WTGPU_STAGED_CONNECT=0|" \
  "u_staged_only|-|WTGPU_SORTED_INTERACT=0|" \
  "u_old_again|-|WTGPU_SORTED_INTERACT=0 WTGPU_STAGED_CONNECT=0|"
