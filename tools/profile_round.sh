#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench workload + separate PMC passes.
# usage: tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*.csv
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $BENCH > $OUT/${TAG}_bench_under_rocprof.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
if [ -n "$DB" ]; then python $R/tools/rocpd_stats.py $DB $OUT/${TAG}_kernel_stats.csv $OUT/${TAG}_dispatches.csv > /dev/null; fi
find /tmp/prof_kt -name "*stats*.csv" -exec cp {} $OUT/ \; 2>/dev/null
# the same with ONE internal stream: exclusive kernel durations (with several streams a kernel's duration includes the time it shares the GPU)
rm -rf /tmp/prof_kt1
WTGPU_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt1 -o kt -- $BENCH > $OUT/${TAG}_bench_streams1_under_rocprof.log 2>&1
DB1=$(find /tmp/prof_kt1 -name "*.db" | head -1)
if [ -n "$DB1" ]; then python $R/tools/rocpd_stats.py $DB1 $OUT/${TAG}_kernel_stats_streams1.csv > /dev/null; fi
BENCH1="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_pmc_$N -o pmc -- $BENCH1 > $OUT/${TAG}_pmc_${N}.log 2>&1
  find /tmp/prof_pmc_$N -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_pmc_${N}_raw.csv \;
  find /tmp/prof_pmc_$N -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_pmc_${N}_raw.csv \;
  python $R/tools/pmc_summary.py /tmp/prof_pmc_$N $OUT/${TAG}_pmc_${N}.csv > /dev/null 2>> $OUT/${TAG}_pmc_${N}.log
done
# calibration of FETCH_SIZE / WRITE_SIZE on a known byte count with the SoA access width (one dword per lane)
NDW=268435456; REP=4
echo "{\"n_dwords\": $NDW, \"repeats\": $REP}" > $OUT/${TAG}_calib.json
for N in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_cal_$N
  timeout 300 rocprofv3 --pmc $N --kernel-trace --output-format csv -d /tmp/prof_cal_$N -o pmc -- python -c "
import sys; sys.path.insert(0, '$R')
import ctypes, torch
from wave_tracer_amd.api import load_library
lib = load_library(); lib.wtgpu_calibrate_copy.argtypes = [ctypes.c_uint64, ctypes.c_int]
torch.cuda.init(); assert lib.wtgpu_calibrate_copy($NDW, $REP) == 0" > $OUT/${TAG}_calib_${N}.log 2>&1
  python $R/tools/pmc_summary.py /tmp/prof_cal_$N $OUT/${TAG}_calib_${N}.csv > /dev/null 2>> $OUT/${TAG}_calib_${N}.log
done
ls -la $OUT
