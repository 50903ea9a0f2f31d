#!/bin/bash
# ONE script for what runs on the GPU box (through gpurun: `gpurun -- 'bash tools/gpu_job.sh <out> <mode> ...'`).  Replaces the one-shot scripts of
# rounds 4-5 (tools/experiments_r04/, tools/r05/run*.sh): what they ran is recorded in profiles/rNN_ab_experiments.log.
#   tools/gpu_job.sh <outdir> tests [pytest arguments]                      the GPU suite (default: tests -m gpu -q -x), log in <outdir>/tests.log
#   tools/gpu_job.sh <outdir> ab "<label>|<lib or ->|<ENV=V ...>|<bench.py arguments>" ...
#                                                                            bench.py per spec (variant libraries: tools/build_variant.sh -> wave_tracer_amd/_v/libwtgpu_<lib>.so)
#   tools/gpu_job.sh <outdir> replay <rounds> "<label>|<lib or ->|<ENV=V ...>" ...
#                                                                            in-situ replay of the first <rounds> trace queues of one step through k_trace_refill and the
#                                                                            alternative form the knobs select (WTGPU_TRACE_AB): times per round, words that differ
#   tools/gpu_job.sh <outdir> stats "<label>|<lib or ->|<ENV=V ...>|<bench.py arguments>" ...
#                                                                            rocprofv3 --kernel-trace --stats of a 3-step run: per-kernel totals per step
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; MODE=$2; shift; shift; mkdir -p $OUT; cd $R
summ() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d["roofline"].get("kernel_ms_per_step_stream_summed", {})
    print("%-26s %6.2f Msamples/s %7.1f ms/step | " % (sys.argv[1], d["value"], d["ms_per_step"]) + " ".join("%s %.0f" % (a.replace("k_", "")[:12], b) for a, b in k.items()))
except Exception as e:
    print(sys.argv[1], "fail", e)
PY
}
case $MODE in
tests)
  ARGS="$*"; [ -z "$ARGS" ] && ARGS="tests -m gpu -q -x"
  ( time python -m pytest $ARGS ) > $OUT/tests.log 2>&1; tail -6 $OUT/tests.log ;;
ab)
  for spec in "$@"; do
    IFS='|' read -r LABEL LIB ENVS ARGS <<< "$spec"
    ( [ "$LIB" != "-" ] && export WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_$LIB.so
      for kv in $ENVS; do export $kv; done
      timeout 300 python bench.py --steps ${AB_STEPS:-8} --warmup 2 --no-cpu-baseline --no-traffic $ARGS > $OUT/$LABEL.json 2> $OUT/$LABEL.err )
    summ $LABEL $OUT/$LABEL.json
  done ;;
replay)
  N=$1; shift
  for spec in "$@"; do
    IFS='|' read -r LABEL LIB ENVS <<< "$spec"
    ( [ "$LIB" != "-" ] && export WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_$LIB.so
      for kv in $ENVS; do export $kv; done
      WTGPU_STREAMS=1 WTGPU_TRACE_AB=$N WTGPU_TRACE_AB_VERBOSE=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-traffic > $OUT/$LABEL.json 2> $OUT/$LABEL.err )
    echo "== $LABEL"; grep "trace ab" $OUT/$LABEL.err | head -$N
  done ;;
stats)
  for spec in "$@"; do
    IFS='|' read -r LABEL LIB ENVS ARGS <<< "$spec"
    ( [ "$LIB" != "-" ] && export WTGPU_LIB=$R/wave_tracer_amd/_v/libwtgpu_$LIB.so
      for kv in $ENVS; do export $kv; done
      cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/prof_$LABEL
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$LABEL -o st -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic $ARGS > $OUT/$LABEL.json 2> $OUT/$LABEL.err )
    python - $OUT/prof_$LABEL $LABEL <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("==", sys.argv[2], "(4 steps incl. the warm-up; ms per step)")
for r in rows:
    if float(r["Percentage"]) > 0.4:
        print("  %-34s calls %6s  %8.2f ms/step  avg %8.1f us  %5.1f %%" % (r["Name"].split("(")[0].replace("wtk::", "")[:34], r["Calls"], float(r["TotalDurationNs"]) / 4e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print("  sum %.1f ms/step" % (tot / 4e6))
PY
  done ;;
*) echo "unknown mode $MODE"; exit 1 ;;
esac
