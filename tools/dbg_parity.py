import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from wave_tracer_amd import Scene, render, develop
from oracle_util import oracle_render, oracle_render_tiles
cases = [("furnace", 32, 4, {}), ("furnace", 24, 4, {"fsd": 1, "lut": (128, 128)}), ("double_slits", 96, 8, {"lut": (128, 128)}),
         ("cornell_box", 32, 4, {"mesh_detail": 0, "lut": (128, 128), "crop_of": 1440}), ("cornell_box", 48, 2, {"mesh_detail": 1, "lut": (128, 128), "crop_of": 1440})]
for name, res, spp, kw in cases:
    sc = Scene(name, res=res, **kw)
    v, w, l = render(sc, spp, seed=7)
    g = develop(sc, v, w, l, spp).astype(np.float64)
    gc = sc.counters()
    ov, ow, ol, oc = oracle_render(sc, 0, spp, 7)
    c = develop(sc, ov, ow, ol, spp).astype(np.float64)
    rel = np.abs(g - c).sum() / np.abs(c).sum()
    d = np.abs(g - c).sum(axis=2); ref = np.abs(c).sum(axis=2)
    same = (d <= 1e-3 * ref + 1e-30).mean()
    print(f"{name:14s} res {res} rel_l1 {rel:.3e} frac_same {same:.4f} overflow {gc['cone_tri_overflow']} fsd {gc['fsd_interactions']}/{oc['fsd_interactions']} seg {gc['segments']}/{oc['segments']}")
