#!/bin/bash
# GPU box: what the driver runs at round end — the full GPU suite, smoke(), and a short bench of the launch-bound configuration.
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/tests.log 2>&1
tail -4 gpurun_out/final/tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 200 python bench.py --scene etoile --res 720 --no-traffic --no-cpu-baseline 2>/dev/null | cut -c1-180
