#!/bin/bash
mkdir -p gpurun_out/r3x
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_gpu_render.py -m gpu -x -q -k "path or etoile" > gpurun_out/r3x/tests.log 2>&1
tail -5 gpurun_out/r3x/tests.log
timeout 600 python bench.py --scene etoile --no-traffic > gpurun_out/r3x/bench_etoile.json 2> gpurun_out/r3x/bench_etoile.err
cat gpurun_out/r3x/bench_etoile.json
