#!/usr/bin/env python3
"""Paired GPU / CPU-checker bias estimate on the dense cornell crop (common random numbers), on its own (GPU box): what
tests/test_gpu_render.py::test_converged_bias_dense_crop asserts, with the raw per-cell sums saved for offline analysis.
usage: paired_bias.py [chunks] [spp per chunk] [out.npz]"""
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from oracle_util import oracle_render  # noqa: E402
from wave_tracer_amd import Scene, render  # noqa: E402


from oracle_util import paired_bias_stats as analyse  # noqa: E402


if __name__ == "__main__":
    chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    out = sys.argv[3] if len(sys.argv) > 3 else None
    t0 = time.time()
    sc = Scene("cornell_box", res=32, mesh_detail=1, lut=(128, 128), crop_of=1440)
    G, C = [], []
    for chunk in range(chunks):
        b, e = chunk * per, (chunk + 1) * per
        v, w, l = render(sc, e - b, seed=31, sample_begin=b)
        ov, ow, ol, _ = oracle_render(sc, b, e, 31)
        G.append(v.sum(axis=2) + l.sum(axis=2))
        C.append(ov.sum(axis=2) + ol.sum(axis=2))
        if (chunk + 1) % 4 == 0 or chunk + 1 == chunks:
            st = analyse(np.array(G), np.array(C))
            print(f"chunks {chunk + 1} ({(chunk + 1) * per} spp): all cells {st['bias_all']:+.2e} +- {st['se_all']:.1e}; {st['n_div']} cells diverge ({st['frac_div']:.2e} of them, "
                  f"{st['n_pos']} GPU-larger, sign p = {st['p_sign']:.3f}, {st['div_share_of_flux']:.1%} of the flux); non-divergent {st['bias_trim']:+.2e} +- {st['se_trim']:.1e}, "
                  f"rel L1 {st['rel_l1_trim']:.2e}; {time.time() - t0:.1f}s", flush=True)
    if out:
        np.savez_compressed(out, G=np.array(G), C=np.array(C))
