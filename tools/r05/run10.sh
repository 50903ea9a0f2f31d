#!/bin/bash
# GPU box, round 5 call 10: exclusive time of k_interact with 1 / 4 / 16 x 64 queue items per atomic on the queue head (one internal stream).
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
O=$R/gpurun_out/r5j; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
cd /tmp && export TMPDIR=/tmp
for V in g1 g4 g16; do
  rm -rf /tmp/p_kt; WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_$V.so WTGPU_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $O/${V}_kt.log 2>&1
  DB=$(find /tmp/p_kt -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB $O/${V}_stats.csv > /dev/null
  echo "$V: $(grep -E 'k_interactENS' $O/${V}_stats.csv | cut -d, -f1-6)"
done
