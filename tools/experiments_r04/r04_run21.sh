#!/bin/bash
# GPU box, round 4, run 21: exclusive kernel times of the etoile (plt_path) workload, one stream
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4v; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WTGPU_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o kt -- python $R/bench.py --scene etoile --res 720 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $OUT/bench.log 2>&1
DB=$(find /tmp/prof_e -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB $OUT/kernel_stats_etoile_streams1.csv $OUT/dispatches_etoile_streams1.csv > /dev/null
cut -d, -f1-4 $OUT/kernel_stats_etoile_streams1.csv | head -14
grep -o '"value": [0-9.]*, "unit"[^}]*"ms_per_step": [0-9.]*' $OUT/bench.log | head -1
