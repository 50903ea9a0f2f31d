import sys, os, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from wave_tracer_amd import Scene, render, develop
name = sys.argv[1] if len(sys.argv) > 1 else "etoile_open"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
md = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sc = Scene(name, res=res, mesh_detail=md)
print("scene ok", sc.stats(), flush=True)
sc.upload(0)
print("upload ok", flush=True)
v, w, l = render(sc, spp, seed=5, device=0)
gc = sc.counters()
print("render ok", l.sum(), v.sum(), flush=True)
from oracle_util import oracle_render
ov, ow, ol, oc = oracle_render(sc, 0, spp, 5)
print("oracle", ol.sum(), ov.sum())
for k in oc:
    if gc[k] != oc[k]: print("  counter", k, gc[k], oc[k])
g = develop(sc, v, w, l, spp); c = develop(sc, ov, ow, ol, spp)
print("rel l1", np.abs(g - c).sum() / max(1e-30, np.abs(c).sum()))
