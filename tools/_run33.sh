#!/bin/bash
O=gpurun_out
timeout 600 python bench.py > $O/r03h_bench_line_1440.json 2> $O/r03h_bench_line_1440.err
timeout 600 python bench.py --scene etoile > $O/r03h_bench_etoile.json 2> $O/r03h_bench_etoile.err
timeout 600 python bench.py --scene bidir_room --res 1920 --steps 8 > $O/r03h_bench_bidir_room.json 2> $O/r03h_bench_bidir_room.err
for f in line_1440 etoile bidir_room; do python - <<PY
import json
d=json.loads(open("$O/r03h_bench_$f.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$f", round(d["value"],2), round(d["ms_per_step"],1), "traffic", r["traffic"], r.get("traffic_source"), "alg/step", r["whole_path"]["alg_bytes_per_step"], "cpu", d["cpu_baseline"]["value"])
PY
done
ls $O | grep -i "traffic\|pmc" | tail
