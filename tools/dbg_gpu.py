import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from wave_tracer_amd import Scene, render, develop
from oracle_util import oracle_render
name=sys.argv[1]; res=int(sys.argv[2]); spp=int(sys.argv[3])
kw = eval(sys.argv[4]) if len(sys.argv)>4 else {}
import os
SEED=int(os.environ.get('SEED','5'))
sc = Scene(name, res=res, **kw)
v,w,l = render(sc, spp, seed=SEED)
print("gpu counters", sc.counters()); print("timings", sc.timings())
ov,ow,ol,oc = oracle_render(sc, 0, spp, SEED)
print("cpu counters", oc)
g = develop(sc, v,w,l,spp).astype(np.float64); c = develop(sc, ov,ow,ol,spp).astype(np.float64)
print("value sum gpu/cpu", v.sum(), ov.sum(), "light", l.sum(), ol.sum(), "weight", w.sum(), ow.sum())
print("rel L1", np.abs(g-c).sum()/np.abs(c).sum())
d = np.abs(g-c).sum(axis=2); idx = np.argsort(d.ravel())[::-1][:5]
for i in idx: print(divmod(i, sc.width), g.reshape(-1,g.shape[2])[i], c.reshape(-1,g.shape[2])[i])
