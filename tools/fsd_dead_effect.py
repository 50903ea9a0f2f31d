#!/usr/bin/env python3
"""CPU only: what the DEAD-aperture shortcut of the Fraunhofer rejection sampler (wt/fsd.h: kFsdDeadRatio) changes against the reference.
An aperture whose segment amplitudes cancel identically (a doubled scene edge) has a scattering function of rounding noise: the reference's
loop runs n x 1024 tries and almost always fails; the shortcut fails the sample at once.  The two differ where the full loop accepts a try
through rounding noise (the walk then continues in a noise direction instead of ending).  Paired renders of the CPU checker with
oracle_set_fsd_dead_ratio(1e-10) / (0) on identical random numbers, plus a second seed for the Monte-Carlo floor.
usage: fsd_dead_effect.py [spp] [chunk]"""
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from oracle_util import load_oracle, oracle_render, paired_bias_stats  # noqa: E402
from wave_tracer_amd import Scene  # noqa: E402
import ctypes as C  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
per = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = load_oracle()
lib.oracle_set_fsd_dead_ratio.argtypes = [C.c_float]
sc = Scene("cornell_box", res=32, mesh_detail=1, lut=(128, 128), crop_of=1440)
t0 = time.time()
A, B, F = [], [], []
ctr = {0: {}, 1: {}}
tt = {0: 0.0, 1: 0.0}
for b in range(0, spp, per):
    e = min(spp, b + per)
    lib.oracle_set_fsd_dead_ratio(1e-10)
    t = time.time()
    v, w, l, c1 = oracle_render(sc, b, e, 31)
    tt[1] += time.time() - t
    A.append(v.sum(axis=2) + l.sum(axis=2))
    v, w, l, _ = oracle_render(sc, b, e, 77)          # independent seed, same configuration: the Monte-Carlo floor
    F.append(v.sum(axis=2) + l.sum(axis=2))
    lib.oracle_set_fsd_dead_ratio(0.0)
    t = time.time()
    v, w, l, c0 = oracle_render(sc, b, e, 31)
    tt[0] += time.time() - t
    lib.oracle_set_fsd_dead_ratio(1e-10)
    B.append(v.sum(axis=2) + l.sum(axis=2))
    for k in c1:
        ctr[1][k] = ctr[1].get(k, 0) + c1[k]
        ctr[0][k] = ctr[0].get(k, 0) + c0[k]
    st = paired_bias_stats(np.array(B), np.array(A))
    a, f = np.array(A).sum(axis=0), np.array(F).sum(axis=0)
    bsum = np.array(B).sum(axis=0)
    nrm = a.mean()
    print(f"cornell crop {e} spp: full loop minus shortcut, all cells {st['bias_all']:+.2e} +- {st['se_all']:.1e} ({st['n_div']} divergent cells, {st['n_pos']} larger); "
          f"image nRMSE {np.sqrt(np.mean((bsum - a) ** 2)) / nrm:.3e}; floor (two seeds) {np.sqrt(np.mean((f - a) ** 2)) / nrm:.3e}; "
          f"fsd interactions {ctr[0]['fsd_interactions']} vs {ctr[1]['fsd_interactions']}; checker time {tt[0]:.0f}s vs {tt[1]:.0f}s", flush=True)
