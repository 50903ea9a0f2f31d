#!/bin/bash
# GPU box, round 4, run 2: axis kernel + deferred exact tests: GPU suite, A/B of the variants, one-stream kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4b/tests.log 2>&1; tail -5 gpurun_out/r4b/tests.log
AB_STEPS=8 bash tools/ab_run.sh r4b "gss0|gss0|" "new|-|" "noaxis|noaxis|" "ex1|ex1|" "ex4|ex4|" "ex16|ex16|" "gss1|gss1|" "gss1_sh12|gss1|WTGPU_SHRINK_R1=12 WTGPU_SHRINK_R2=24" "lb2|lb2|" "new_b96|-|WTGPU_CONE_BUDGET=96" "new_b128|-|WTGPU_CONE_BUDGET=128" "new_b48|-|WTGPU_CONE_BUDGET=48"
cd /tmp && export TMPDIR=/tmp
WTGPU_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt1 -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $R/gpurun_out/r4b/bench_streams1.log 2>&1
DB1=$(find /tmp/prof_kt1 -name "*.db" | head -1)
[ -n "$DB1" ] && python $R/tools/rocpd_stats.py $DB1 $R/gpurun_out/r4b/kernel_stats_streams1.csv $R/gpurun_out/r4b/dispatches_streams1.csv > /dev/null
head -8 $R/gpurun_out/r4b/kernel_stats_streams1.csv
