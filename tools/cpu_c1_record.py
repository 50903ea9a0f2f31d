#!/usr/bin/env python3
"""BASELINE.json configs[0] (C1): cornell-box res 256, spp 64, plt_bdpt, on the CPU checker alone — the record the GPU lines' `cpu_baseline`
(a bounded sample of the 1440^2 film) does not replace.  Writes profiles/<tag>_cpu_c1_record.json: samples, seconds, threads, Msamples/s, film
sums and event counters.  usage: python tools/cpu_c1_record.py [tag] [spp]   (about a quarter of an hour on 8 cores at spp 64)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from wave_tracer_amd import Scene
from oracle_util import oracle_render

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
threads = os.cpu_count() or 1
sc = Scene("cornell_box", res=256, mesh_detail=1)
t = time.time()
v, w, l, c = oracle_render(sc, 0, spp, 1, threads=threads)
dt = time.time() - t
n = sc.width * sc.height * spp
rec = {"config": "BASELINE.json configs[0]: cornell-box stand-in (283,154 triangles) res=256 spp=%d plt_bdpt max_depth 16 RR MIS FSD, CPU checker (oracle/), %d threads" % (spp, threads),
       "samples": n, "seconds": dt, "threads": threads, "msamples_per_s": n / dt / 1e6,
       "film_sums": {"value": float(v.sum()), "weight": float(w.sum()), "light": float(l.sum())},
       "counters_per_sample": {k: x / n for k, x in c.items()},
       "note": "scalar fp32 restatement of the reference's algorithm on this container's CPU cores (not the upstream AVX2 binary, not the GPU box's host); baseline only"}
out = os.path.join(ROOT, "profiles", f"{tag}_cpu_c1_record.json")
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("samples", "seconds", "threads", "msamples_per_s")}))
