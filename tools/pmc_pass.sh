#!/bin/bash
# one PMC pass with arbitrary counters.  usage: tools/pmc_pass.sh <tag> "<counters>"
TAG=$1; C=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pmc_x
timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_pmc_x -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/${TAG}.log 2>&1
python $R/tools/pmc_summary.py /tmp/prof_pmc_x $OUT/${TAG}.csv > /dev/null 2>> $OUT/${TAG}.log
