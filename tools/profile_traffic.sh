#!/bin/bash
# GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE PMC passes, separate runs) of one bench step of a given workload.
# usage: tools/profile_traffic.sh <tag> <scene> <res>     -> gpurun_out/<tag>_pmc_{FETCH,WRITE}_SIZE.csv
TAG=$1; SCENE=$2; RES=$3
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --scene $SCENE --res $RES --steps 1 --warmup 0 --no-cpu-baseline"
for N in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_t_$N
  timeout 600 rocprofv3 --pmc $N --kernel-trace --output-format csv -d /tmp/prof_t_$N -o pmc -- $B > $OUT/${TAG}_pmc_${N}.log 2>&1
  python $R/tools/pmc_summary.py /tmp/prof_t_$N $OUT/${TAG}_pmc_${N}.csv > /dev/null 2>> $OUT/${TAG}_pmc_${N}.log
done
