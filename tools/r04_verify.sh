#!/bin/bash
# GPU box, round 4, final run: the round's verification — GPU suite, smoke, the three bench lines, rocprofv3 evidence (tools/profile_round.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4final2
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r4final2/tests.log 2>&1; tail -4 gpurun_out/r4final2/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 400 python bench.py > gpurun_out/r4final2/bench_line_1440.json 2> gpurun_out/r4final2/bench_line_1440.err; cut -c1-400 gpurun_out/r4final2/bench_line_1440.json
timeout 300 python bench.py --scene etoile --res 720 > gpurun_out/r4final2/bench_etoile.json 2>/dev/null; cut -c1-200 gpurun_out/r4final2/bench_etoile.json
timeout 300 python bench.py --scene etoile --res 1440 --no-cpu-baseline --no-traffic > gpurun_out/r4final2/bench_etoile_1440.json 2>/dev/null; cut -c1-200 gpurun_out/r4final2/bench_etoile_1440.json
timeout 300 python bench.py --scene bidir_room --res 1920 > gpurun_out/r4final2/bench_bidir_room.json 2>/dev/null; cut -c1-200 gpurun_out/r4final2/bench_bidir_room.json
bash tools/profile_round.sh r04 > gpurun_out/r4final2/profile_round.log 2>&1; tail -3 gpurun_out/r4final2/profile_round.log
