import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, time
from wave_tracer_amd import Scene, render, develop
from oracle_util import oracle_render
for name, res, spp, kw in [("double_slits", 360, 16, {"lut": (256, 256)}), ("double_slits", 720, 32, {"lut": (256, 256)}), ("cornell_box", 64, 16, {"mesh_detail": 1, "crop_of": 1440})]:
    sc = Scene(name, res=res, **kw)
    t = time.time(); v, w, l = render(sc, spp, seed=7); tg = time.time() - t
    g = develop(sc, v, w, l, spp).astype(np.float64)
    t = time.time(); ov, ow, ol, oc = oracle_render(sc, 0, spp, 7); tc = time.time() - t
    c = develop(sc, ov, ow, ol, spp).astype(np.float64)
    ov2, ow2, ol2, _ = oracle_render(sc, 0, spp, 8)
    c2 = develop(sc, ov2, ow2, ol2, spp).astype(np.float64)
    rm = lambda a, b: np.sqrt(np.mean((a - b) ** 2)) / np.mean(b)
    print(f"{name} res {res} spp {spp}: RMSE/mean GPU-vs-CPU same seed {rm(g, c):.3e}; CPU-vs-CPU other seed (noise floor) {rm(c2, c):.3e}; rel L1 {np.abs(g-c).sum()/np.abs(c).sum():.3e}; gpu {tg:.2f}s cpu {tc:.2f}s")
