#!/bin/bash
mkdir -p gpurun_out/r3final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3final/tests.log 2>&1
tail -4 gpurun_out/r3final/tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 200 python bench.py --scene etoile --res 720 --no-traffic --no-cpu-baseline 2>/dev/null | cut -c1-180
