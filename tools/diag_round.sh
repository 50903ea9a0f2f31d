#!/bin/bash
# GPU box: diagnostics of one default bench step — in-kernel clock breakdowns (WTGPU_PROFILE) and per-dispatch durations with one stream.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=${1:-diag}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
WTGPU_PROFILE=2 $B > $OUT/${TAG}_p2.json 2> $OUT/${TAG}_p2.err
WTGPU_PROFILE=1 $B > $OUT/${TAG}_p1.json 2> $OUT/${TAG}_p1.err
WTGPU_STREAMS=1 WTGPU_PROFILE=3 $B > $OUT/${TAG}_p3.json 2> $OUT/${TAG}_p3.err
rm -rf /tmp/prof_d1
WTGPU_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d1 -o kt -- $B > $OUT/${TAG}_s1.log 2>&1
DB1=$(find /tmp/prof_d1 -name "*.db" | head -1)
[ -n "$DB1" ] && python $R/tools/rocpd_stats.py $DB1 $OUT/${TAG}_s1_kernel_stats.csv $OUT/${TAG}_s1_dispatches.csv > /dev/null
grep -h "wtgpu profile" $OUT/${TAG}_p1.err $OUT/${TAG}_p2.err $OUT/${TAG}_p3.err
