import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from wave_tracer_amd import Scene
from test_oracle import oracle_trace
sc = Scene("cornell_box", res=16, mesh_detail=0, lut=(32, 32)); sc.upload(0)
g = np.load('/root/repo/tests/golden/cornell_traversal.npz')
gd, gt, gb, gf = sc.trace_rays(g["rays"])
od, ot = g["dist"], g["tuid"]
hit = np.isfinite(od)
print("hit", hit.sum(), "same", (gt==ot)[hit].mean(), "dist close", np.allclose(gd[hit], od[hit], rtol=1e-5, atol=1e-7))
bad = np.where(hit & (gt != ot))[0][:10]
for i in bad: print(i, gd[i], od[i], gt[i], ot[i])
o2 = oracle_trace(sc, g["rays"])
print("oracle now vs golden", (o2[1]==ot)[hit].mean())
