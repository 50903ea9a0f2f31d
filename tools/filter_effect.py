#!/usr/bin/env python3
"""CPU only: what the final-slab filter of the interaction record (traversal_common.hpp:131-135, implemented "as written" here and in
the device code) changes against the reference "as executed" (the traversal never records a distance, src/ads/bvh8w.cpp:175, so the
filter never fires and the record keeps every triangle met while the slab was still wider).  Paired renders of the CPU checker with
oracle_set_region_filter(1) / (0) on identical random numbers, plus a second seed with the filter on for the Monte-Carlo floor.
usage: filter_effect.py <scene: cornell_crop | double_slits> [spp] [chunk]"""
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from oracle_util import load_oracle, oracle_render, paired_bias_stats  # noqa: E402
from wave_tracer_amd import Scene  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cornell_crop"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
per = int(sys.argv[3]) if len(sys.argv) > 3 else 64
lib = load_oracle()
sc = Scene("cornell_box", res=32, mesh_detail=1, lut=(128, 128), crop_of=1440) if which == "cornell_crop" else Scene("double_slits", res=96, lut=(128, 128))
t0 = time.time()
A, B, F = [], [], []
ctr = {0: {}, 1: {}}
for b in range(0, spp, per):
    e = min(spp, b + per)
    lib.oracle_set_region_filter(1)
    v, w, l, c1 = oracle_render(sc, b, e, 31)
    A.append(v.sum(axis=2) + l.sum(axis=2))
    v, w, l, _ = oracle_render(sc, b, e, 77)          # independent seed, same configuration: the Monte-Carlo floor
    F.append(v.sum(axis=2) + l.sum(axis=2))
    lib.oracle_set_region_filter(0)
    v, w, l, c0 = oracle_render(sc, b, e, 31)
    lib.oracle_set_region_filter(1)
    B.append(v.sum(axis=2) + l.sum(axis=2))
    for k in c1:
        ctr[1][k] = ctr[1].get(k, 0) + c1[k]
        ctr[0][k] = ctr[0].get(k, 0) + c0[k]
    st = paired_bias_stats(np.array(B), np.array(A))
    a, f = np.array(A).sum(axis=0), np.array(F).sum(axis=0)
    bsum = np.array(B).sum(axis=0)
    nrm = a.mean()
    print(f"{which} {e} spp: as-executed minus as-written, all cells {st['bias_all']:+.2e} +- {st['se_all']:.1e} ({st['n_div']} divergent cells, {st['n_pos']} larger); "
          f"image nRMSE as-executed vs as-written {np.sqrt(np.mean((bsum - a) ** 2)) / nrm:.3e}; floor (two seeds, as written) {np.sqrt(np.mean((f - a) ** 2)) / nrm:.3e}; "
          f"fsd interactions {ctr[0]['fsd_interactions']} vs {ctr[1]['fsd_interactions']}; {time.time() - t0:.0f}s", flush=True)
