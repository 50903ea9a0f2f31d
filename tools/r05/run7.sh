#!/bin/bash
# GPU box, round 5 call 7: xnack- build A/B; the two-rank gloo bench test; traffic and exclusive kernel times of the two forms of pass A / connections.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out/r5g; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
AB_STEPS=10 bash tools/ab_run.sh r5g "default|-||" "xnack|xnack||" "default2|-||" "xnack2|xnack||"
timeout 200 python -m pytest tests/test_gpu_render.py -q -x -k "two_ranks_strong" --timeout 180 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for CFG in one sorted_staged; do
  if [ $CFG = one ]; then export WTGPU_SORTED_INTERACT=0 WTGPU_STAGED_CONNECT=0; else export WTGPU_SORTED_INTERACT=1 WTGPU_STAGED_CONNECT=1; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_$C; timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-traffic > $O/${CFG}_$C.log 2>&1
    python $R/tools/pmc_summary.py /tmp/p_$C $O/${CFG}_pmc_$C.csv > /dev/null 2>&1
  done
  rm -rf /tmp/p_kt; WTGPU_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $O/${CFG}_kt.log 2>&1
  DB=$(find /tmp/p_kt -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB $O/${CFG}_kernel_stats_streams1.csv > /dev/null
done
head -14 $O/one_kernel_stats_streams1.csv | cut -c1-120; head -18 $O/sorted_staged_kernel_stats_streams1.csv | cut -c1-120
