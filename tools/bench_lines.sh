#!/bin/bash
# GPU box: the bench lines of the three BASELINE workloads (roofline with live PMC traffic, cpu_baseline) -> gpurun_out/<tag>_bench_*.json
# usage: tools/bench_lines.sh <tag>        (copy what is to be kept into profiles/)
TAG=${1:-r00}
O=gpurun_out
mkdir -p $O
timeout 600 python bench.py > $O/${TAG}_bench_line_1440.json 2> $O/${TAG}_bench_line_1440.err
timeout 600 python bench.py --scene etoile > $O/${TAG}_bench_etoile.json 2> $O/${TAG}_bench_etoile.err
timeout 600 python bench.py --scene bidir_room --res 1920 --steps 8 > $O/${TAG}_bench_bidir_room.json 2> $O/${TAG}_bench_bidir_room.err
for f in line_1440 etoile bidir_room; do python - <<PY
import json
d=json.loads(open("$O/${TAG}_bench_$f.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$f", round(d["value"],2), "Msamples/s", round(d["ms_per_step"],1), "ms/step; traffic per launch", r["traffic"], "alg bytes/step", r["whole_path"]["alg_bytes_per_step"], "cpu", d["cpu_baseline"]["value"])
PY
done
