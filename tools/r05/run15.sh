#!/bin/bash
# GPU box, the round's last run: the uniform-addend queue grabs (wave_grab / wave_grab_item).  The shipped (unity) build: GPU suite, smoke, bench,
# kernel statistics with one internal stream; the split build on the plt_path tests + the parity tests of the headline scene.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r5o; mkdir -p $O
T0=$(date +%s)
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
timeout 230 python -m pytest tests -m gpu -q --timeout 150 > $O/tests.log 2>&1; echo "unity tests rc=$?"; tail -2 $O/tests.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 100 python bench.py --no-cpu-baseline --no-traffic > $O/bench.json 2>/dev/null; cut -c1-200 $O/bench.json
echo "elapsed $(( $(date +%s) - T0 )) s"
WTGPU_LIB=$PWD/wave_tracer_amd/_v/libwtgpu_splitfix3.so timeout 60 python -m pytest -q --timeout 50 -x tests/test_gpu_path.py tests/test_emitters.py -m gpu > $O/split_path.log 2>&1; echo "split path rc=$?"; tail -1 $O/split_path.log
cd /tmp && export TMPDIR=/tmp
WTGPU_STREAMS=1 timeout 100 rocprofv3 --kernel-trace --stats -d /tmp/p_kt1 -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $R/$O/kt1.log 2>&1
DB=$(find /tmp/p_kt1 -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB $R/$O/r05_kernel_stats_streams1_final.csv > /dev/null
echo "elapsed $(( $(date +%s) - T0 )) s"
