#!/bin/bash
# GPU box, round 4, run 19: bidir_room — how much of a batch is its thin tail (every batch has walks that restart behind empty apertures until the round cap)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4t; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WTGPU_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o kt -- python $R/bench.py --scene bidir_room --res 1920 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $OUT/bench.log 2>&1
DB=$(find /tmp/prof_b -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB $OUT/kernel_stats_bidir_streams1.csv $OUT/dispatches_bidir_streams1.csv > /dev/null
cut -d, -f1-4 $OUT/kernel_stats_bidir_streams1.csv | head -16
tail -1 $OUT/bench.log | cut -c1-200
