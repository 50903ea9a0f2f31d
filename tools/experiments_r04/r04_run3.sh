#!/bin/bash
# GPU box, round 4, run 3: batched leaf fetches (1 / 2 / 4 triangles per fetch), register budgets, batch size
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4c
timeout 600 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_render.py -m gpu -x -q -k "traversal or parity_small or golden or dense" > gpurun_out/r4c/tests.log 2>&1; tail -3 gpurun_out/r4c/tests.log
AB_STEPS=8 bash tools/ab_run.sh r4c "gss0|gss0|" "new_b4|-|" "noaxis_b4|noaxis|" "b1|b1|" "b2|b2|" "b4l2|b4l2|" "b2l2|b2l2|" "new_2spp|-|WTGPU_STATE_GB=220|--spp-per-step 2 --batch 4147200" "gss0_2spp|gss0|WTGPU_STATE_GB=220|--spp-per-step 2 --batch 4147200"
AB_STEPS=8 bash tools/ab_run.sh r4c "etoile_base|base||--scene etoile --res 720" "etoile_new|-||--scene etoile --res 720" "etoile1440_base|base||--scene etoile --res 1440" "etoile1440_new|-||--scene etoile --res 1440"
cd /tmp && export TMPDIR=/tmp
WTGPU_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt1 -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $R/gpurun_out/r4c/bench_streams1.log 2>&1
DB1=$(find /tmp/prof_kt1 -name "*.db" | head -1)
[ -n "$DB1" ] && python $R/tools/rocpd_stats.py $DB1 $R/gpurun_out/r4c/kernel_stats_streams1.csv $R/gpurun_out/r4c/dispatches_streams1.csv > /dev/null
head -8 $R/gpurun_out/r4c/kernel_stats_streams1.csv
