#!/bin/bash
# GPU box, round 4, run 4: one-stream kernel traces of variants (per-round durations) + SQ counters of the trace kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r4d
for V in gss0 b1 b2l2; do
  rm -rf /tmp/prof_$V
  WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_$V.so WTGPU_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$V -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $R/gpurun_out/r4d/bench_$V.log 2>&1
  DB1=$(find /tmp/prof_$V -name "*.db" | head -1)
  [ -n "$DB1" ] && python $R/tools/rocpd_stats.py $DB1 $R/gpurun_out/r4d/kernel_stats_$V.csv $R/gpurun_out/r4d/dispatches_$V.csv > /dev/null
done
for V in gss0 b1; do
 for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/pmc_x
  WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_$V.so WTGPU_STREAMS=1 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_x -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-traffic > $R/gpurun_out/r4d/pmc_${V}_$N.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_x $R/gpurun_out/r4d/pmc_${V}_$N.csv > /dev/null 2>> $R/gpurun_out/r4d/pmc_${V}_$N.log
 done
done
ls $R/gpurun_out/r4d
