#!/bin/bash
# GPU box, round 5 call 8: primary triangles from the axis query always (WTGPU_PRIMARY_AXIS=1): parity suite + A/B.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5h; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
AB_STEPS=10 bash tools/ab_run.sh r5h "default|-||" "paxis|-|WTGPU_PRIMARY_AXIS=1|" "default2|-||" "paxis2|-|WTGPU_PRIMARY_AXIS=1|" "paxis_room|-|WTGPU_PRIMARY_AXIS=1|--scene bidir_room --res 1920" "default_room|-||--scene bidir_room --res 1920"
WTGPU_PRIMARY_AXIS=1 timeout 500 python -m pytest tests -m gpu -q --timeout 150 -k "not path and not etoile and not two_ranks and not two_gpus" > $O/tests_paxis.log 2>&1; echo "tests rc=$?"; tail -12 $O/tests_paxis.log
