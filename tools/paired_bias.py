#!/usr/bin/env python3
"""Part (1) of tests/test_gpu_render.py::test_converged_bias_dense_crop on its own (GPU box): paired GPU / CPU-checker bias on the dense
cornell crop, common random numbers.  usage: paired_bias.py [chunks] [spp per chunk]"""
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from oracle_util import oracle_render  # noqa: E402
from wave_tracer_amd import Scene, render  # noqa: E402

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 16
per = int(sys.argv[2]) if len(sys.argv) > 2 else 16
t0 = time.time()
sc = Scene("cornell_box", res=32, mesh_detail=1, lut=(128, 128), crop_of=1440)
G, C = [], []
for chunk in range(chunks):
    b, e = chunk * per, (chunk + 1) * per
    v, w, l = render(sc, e - b, seed=31, sample_begin=b)
    ov, ow, ol, _ = oracle_render(sc, b, e, 31)
    G.append(v.sum(axis=2) + l.sum(axis=2))
    C.append(ov.sum(axis=2) + ol.sum(axis=2))
    G_, C_ = np.array(G), np.array(C)
    d = G_ - C_
    div = np.abs(d) > 0.5 * np.maximum(G_, C_)
    print(f"chunks {chunk + 1}: all cells {d.sum() / C_.sum():+.2e}; {div.sum()} of {d.size} diverge ({div.mean():.2e}); non-divergent {d[~div].sum() / C_[~div].sum():+.2e}; "
          f"{time.time() - t0:.1f}s", flush=True)
