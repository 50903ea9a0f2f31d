#!/bin/bash
# GPU box, round 4, run 18: what the launches of empty rounds cost — the same build launching 96 / 48 / 32 rounds per batch (the headline workload needs ~25)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4r
AB_STEPS=8 bash tools/ab_run.sh r4r "r96|r96||" "r48|r48||" "r32|r32||" "r96b|r96||" "r48b|r48||" "r32b|r32||" "e96|r96||--scene etoile --res 720" "e48|r48||--scene etoile --res 720" "e32|r32||--scene etoile --res 720" 2>&1 | tee gpurun_out/r4r/ab.log
