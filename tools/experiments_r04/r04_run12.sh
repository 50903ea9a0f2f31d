#!/bin/bash
# GPU box, round 4, run 12: which instruction-fetch / cache counters exist, and what they read for one step of the headline workload
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4l; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -i -o "\b\(SQC\?_[A-Z0-9_]*\(ICACHE\|IFETCH\|INST_CACHE\|DCACHE\)[A-Z0-9_]*\)" $OUT/counters.txt | sort -u | tr '\n' ' '; echo
BENCH1="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-traffic"
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/p_$N
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$N -o pmc -- $BENCH1 > $OUT/pmc_$N.log 2>&1
  python $R/tools/pmc_summary.py /tmp/p_$N $OUT/pmc_$N.csv > /dev/null 2>> $OUT/pmc_$N.log
  echo "== $C"; head -6 $OUT/pmc_$N.csv | cut -c1-600; tail -2 $OUT/pmc_$N.log | cut -c1-300
done
