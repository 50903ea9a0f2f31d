#!/bin/bash
# GPU box, round 5, final run: what the driver runs at round end (GPU suite, smoke, bench) + the round's rocprofv3 evidence -> gpurun_out/r5final
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r5final; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 HSA_KERNARG_POOL_SIZE=16777216
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 400 python bench.py > $O/bench_line_1440.json 2> $O/bench_line_1440.err; cut -c1-300 $O/bench_line_1440.json
timeout 300 python bench.py --scene etoile --res 720 > $O/bench_etoile.json 2>/dev/null; cut -c1-200 $O/bench_etoile.json
timeout 300 python bench.py --scene bidir_room --res 1920 > $O/bench_bidir_room.json 2>/dev/null; cut -c1-200 $O/bench_bidir_room.json
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log
# lane utilisation, L1 / L2 / texture-address passes of the final build (one step each)
bash tools/pmc_pass.sh r05_pmc_SQ_lane_utilisation "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
bash tools/pmc_pass.sh r05_pmc_TCP "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
bash tools/pmc_pass.sh r05_pmc_TCC "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
bash tools/pmc_pass.sh r05_pmc_TA "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
# exclusive kernel times (one internal stream) of the other forms of pass A / the connections
cd /tmp && export TMPDIR=/tmp
for CFG in "sorted1_staged WTGPU_SORTED_INTERACT=1 WTGPU_STAGED_CONNECT=1" "sorted2 WTGPU_SORTED_INTERACT=2" "coop_io WTGPU_COOP_IO=1"; do
  set -- $CFG; N=$1; shift
  rm -rf /tmp/p_kt; env "$@" WTGPU_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $O/${N}_kt.log 2>&1
  DB=$(find /tmp/p_kt -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB $R/gpurun_out/r05_kernel_stats_streams1_$N.csv > /dev/null
done
ls $R/gpurun_out | grep r05_ | head -40
