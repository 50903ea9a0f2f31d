#!/bin/bash
# GPU box, round 4, run 8: non-temporal record accesses, register budgets of the interaction / connection kernels, and what ordering
# the queries by position and direction would buy the per-lane traversal (tools/bench_coherence.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4h
bash tools/ab_run.sh r4h "cur|cur||" "nt|nt||" "lc1|lc1||" "li2|li2||" "li3|li3||" "cur2|cur||" "nt2|nt||" 2>&1 | tee gpurun_out/r4h/ab.log
timeout 600 python tools/bench_coherence.py 2000000 2>&1 | grep -v Warning | tee gpurun_out/r4h/coherence.log
