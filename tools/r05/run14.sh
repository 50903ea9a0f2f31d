#!/bin/bash
# GPU box, last run of round 5: the branch-free queue grab (wtgpu_kernels.h: wave_grab0).  (1) the split build on the plt_path tests that never
# ended with it; (2) the shipped (unity) build: GPU suite, smoke, bench, kernel statistics with one internal stream; (3) if (1) passed and time is
# left: the split build on the whole suite + bench.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r5n; mkdir -p $O
T0=$(date +%s); left() { echo $(( ${1} - ($(date +%s) - T0) )); }
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
SPLIT=$PWD/wave_tracer_amd/_v/libwtgpu_splitfix2.so
WTGPU_LIB=$SPLIT timeout 120 python -m pytest -q --timeout 100 -x tests/test_gpu_path.py tests/test_emitters.py -m gpu > $O/split_path.log 2>&1; S=$?; echo "split path rc=$S"; tail -2 $O/split_path.log
timeout 400 python -m pytest tests -m gpu -q --timeout 200 > $O/tests.log 2>&1; echo "unity tests rc=$?"; tail -2 $O/tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 200 python bench.py --no-cpu-baseline --no-traffic > $O/bench.json 2>/dev/null; cut -c1-200 $O/bench.json
cd /tmp && export TMPDIR=/tmp
WTGPU_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_kt1 -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $R/$O/kt1.log 2>&1
DB=$(find /tmp/p_kt1 -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB $R/$O/r05_kernel_stats_streams1_final.csv > /dev/null
cd $R
echo "elapsed $(( $(date +%s) - T0 )) s"
if [ $S -eq 0 ] && [ $(left 640) -gt 260 ]; then
  WTGPU_LIB=$SPLIT timeout $(( $(left 640) - 60 )) python -m pytest tests -m gpu -q --timeout 200 > $O/split_all.log 2>&1; echo "split all rc=$?"; tail -2 $O/split_all.log
  [ $(left 640) -gt 50 ] && { WTGPU_LIB=$SPLIT timeout 50 python bench.py --no-cpu-baseline --no-traffic --steps 8 > $O/split_bench.json 2>/dev/null; cut -c1-200 $O/split_bench.json; }
fi
echo "elapsed $(( $(date +%s) - T0 )) s"
