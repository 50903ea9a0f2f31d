#!/bin/bash
# GPU box, round 4, run 1: GPU suite on the new build, A/B against round 3's library, one-stream kernel trace of the new build.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4a/tests.log 2>&1; tail -5 gpurun_out/r4a/tests.log
AB_STEPS=8 bash tools/ab_run.sh r4a "base|base|" "gss0|gss0|" "new|-|" "new_noshrink|-|WTGPU_SHRINK_R1=96"
cd /tmp && export TMPDIR=/tmp
WTGPU_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt1 -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $R/gpurun_out/r4a/bench_streams1.log 2>&1
DB1=$(find /tmp/prof_kt1 -name "*.db" | head -1)
[ -n "$DB1" ] && python $R/tools/rocpd_stats.py $DB1 $R/gpurun_out/r4a/kernel_stats_streams1.csv $R/gpurun_out/r4a/dispatches_streams1.csv > /dev/null
head -20 $R/gpurun_out/r4a/kernel_stats_streams1.csv
