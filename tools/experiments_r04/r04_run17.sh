#!/bin/bash
# GPU box, round 4, run 17: exclusive kernel times (one stream) of the build with / without the bounding-sphere filter
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4q; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WTGPU_STREAMS=1
for V in cur sp8 sp1; do
  rm -rf /tmp/prof_$V
  WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_$V.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$V -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $OUT/bench_$V.log 2>&1
  DB=$(find /tmp/prof_$V -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $DB $OUT/kernel_stats_$V.csv > /dev/null
  echo "== $V"; cut -d, -f1-4 $OUT/kernel_stats_$V.csv | head -6
done
