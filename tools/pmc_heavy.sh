#!/bin/bash
# PMC breakdown per kernel (two SQ passes) on the bench workload.  usage: tools/pmc_heavy.sh <tag> [env...]
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH1="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_pmc_$i -o pmc -- $BENCH1 > $OUT/${TAG}_pass$i.log 2>&1
  python $R/tools/pmc_summary.py /tmp/prof_pmc_$i $OUT/${TAG}_pass$i.csv > /dev/null 2>> $OUT/${TAG}_pass$i.log
done
