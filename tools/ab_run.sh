#!/bin/bash
# GPU box: A/B of device-code variants on the default bench workload.  usage: ab_run.sh <outdir> "<label>|<lib or ->|<ENV=V ...>|<extra bench.py arguments>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
cd $R
for spec in "$@"; do
  IFS='|' read -r LABEL LIB ENVS ARGS <<< "$spec"
  ( [ "$LIB" != "-" ] && export WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_$LIB.so
    for kv in $ENVS; do export $kv; done
    timeout 150 python bench.py --steps ${AB_STEPS:-6} --warmup 2 --no-cpu-baseline --no-traffic $ARGS > $OUT/$LABEL.json 2> $OUT/$LABEL.err )
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$LABEL.json").read().strip().splitlines()[-1])
    k=d["roofline"].get("kernel_ms_per_step_stream_summed",{})
    print("%-22s %6.2f Msps %7.1f ms | "%("$LABEL", d["value"], d["ms_per_step"]) + " ".join("%s %.0f"%(a.replace("k_","").replace("interact","int")[:14],b) for a,b in k.items()))
except Exception as e: print("$LABEL", "fail", e)
PY
done
