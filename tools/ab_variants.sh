mkdir -p gpurun_out/ab
for V in base X Z; do
  if [ $V = base ]; then unset WTGPU_LIB; else export WTGPU_LIB=$PWD/wave_tracer_amd/_v/libwtgpu_$V.so; fi
  timeout 60 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/ab/$V.json 2> gpurun_out/ab/$V.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab/$V.json").read().strip().splitlines()[-1])
    print("$V", d["value"], d["ms_per_step"], json.dumps(d["roofline"].get("kernel_ms_per_step_stream_summed")))
except Exception as e: print("$V fail", e)
PY
done
