#!/bin/bash
# GPU box, round 5 call 1: the GPU suite on the new default (material-sorted pass A + staged connections), then A/B against the one-kernel forms.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5a; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/tests.log 2>&1; RC=$?
tail -15 $O/tests.log
if [ $RC -ne 0 ]; then
  echo "=== failed; last-failed with the one-kernel pass A"; WTGPU_SORTED_INTERACT=0 timeout 300 python -m pytest tests -m gpu -q -x --lf --timeout 300 2>&1 | tail -5
  echo "=== last-failed with the one-kernel connections"; WTGPU_STAGED_CONNECT=0 timeout 300 python -m pytest tests -m gpu -q -x --lf --timeout 300 2>&1 | tail -5
fi
export WTGPU_VERBOSE=1
AB_STEPS=6 bash tools/ab_run.sh r5a \
  "old|-|WTGPU_SORTED_INTERACT=0 WTGPU_STAGED_CONNECT=0|" \
  "sorted|-|WTGPU_STAGED_CONNECT=0|" \
  "staged|-|WTGPU_SORTED_INTERACT=0|" \
  "both|-||" \
  "both_cls43|cls43||" \
  "both_mis3|mis3||" \
  "both_room|-||--scene bidir_room --res 1920" \
  "old_room|-|WTGPU_SORTED_INTERACT=0 WTGPU_STAGED_CONNECT=0|--scene bidir_room --res 1920"
grep -h "high water" $O/*.err | sort | uniq -c | head
