#!/bin/bash
# GPU box: what the driver runs at round end (GPU suite, smoke, bench lines of the three BASELINE workloads) + the round's rocprofv3 evidence, all on
# the library as built (`make`: the single-translation-unit build).   usage: tools/round_verify.sh <tag>   e.g. r06  ->  gpurun_out/<tag>_* and gpurun_out/<tag>final/
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${TAG}final; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
( time python -m pytest tests -m gpu -q --timeout 600 ) > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 500 python bench.py > $O/bench_line_1440.json 2> $O/bench_line_1440.err; cut -c1-300 $O/bench_line_1440.json
timeout 400 python bench.py --scene etoile --res 720 > $O/bench_etoile.json 2>/dev/null; cut -c1-200 $O/bench_etoile.json
timeout 400 python bench.py --scene bidir_room --res 1920 > $O/bench_bidir_room.json 2>/dev/null; cut -c1-200 $O/bench_bidir_room.json
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log
# lane utilisation, L1 / L2 / texture-address passes (one step each)
bash tools/pmc_pass.sh ${TAG}_pmc_SQ_lane_utilisation "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
bash tools/pmc_pass.sh ${TAG}_pmc_TCP "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
bash tools/pmc_pass.sh ${TAG}_pmc_TCC "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
bash tools/pmc_pass.sh ${TAG}_pmc_TA "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
# the same workload with the round's alternative forms, one internal stream (exclusive kernel times)
cd /tmp && export TMPDIR=/tmp
for CFG in "refill WTGPU_TRACE_STAGED=0" "phase_machine WTGPU_TRACE_STAGED=0 WTGPU_TRACE_SM=1" "etoile _ARGS=--scene_etoile_--res_720"; do
  set -- $CFG; N=$1; shift
  ARGS=""; ENVS=""
  for kv in "$@"; do case $kv in _ARGS=*) ARGS=$(echo ${kv#_ARGS=} | tr '_' ' ');; *) ENVS="$ENVS $kv";; esac; done
  rm -rf /tmp/p_kt; env $ENVS WTGPU_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic $ARGS > $O/${N}_kt.log 2>&1
  DB=$(find /tmp/p_kt -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_kernel_stats_streams1_$N.csv > /dev/null
done
ls $R/gpurun_out | grep ${TAG}_ | head -60
