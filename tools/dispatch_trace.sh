#!/bin/bash
# kernel-trace of one bench run -> gpurun_out/<tag>_dispatches.csv + <tag>_kernel_stats.csv
TAG=${1:-kt}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB $OUT/${TAG}_kernel_stats.csv $OUT/${TAG}_dispatches.csv > /dev/null
