#!/bin/bash
# GPU box, round 4, run 10: exclusive kernel times (one stream) with and without sorted round queues (prototype library of run 9)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4j; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_sort.so WTGPU_STREAMS=1
for S in 0 4; do
  rm -rf /tmp/prof_s$S
  WTGPU_SORT_ROUNDS=$S timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s$S -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $OUT/bench_s$S.log 2>&1
  DB=$(find /tmp/prof_s$S -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $DB $OUT/kernel_stats_sort$S.csv $OUT/dispatches_sort$S.csv > /dev/null
  echo "== sort rounds $S"; cut -d, -f1-4 $OUT/kernel_stats_sort$S.csv | head -14
done
