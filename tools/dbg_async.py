import sys, time, os
sys.path.insert(0, '/root/repo')
import torch
from wave_tracer_amd import Scene
from wave_tracer_amd.render import alloc_films
res = int(sys.argv[1]); K = int(sys.argv[2])
sc = Scene("cornell_box", res=res, mesh_detail=1)
sc.upload(0, res * res)
dev = torch.device("cuda", 0)
v, w, l = alloc_films(sc, dev)
st = torch.cuda.current_stream(dev).cuda_stream
sc.render_into(v, w, l, 0, 1, 1, st)
torch.cuda.synchronize()
sc.reset_counters()
t0 = time.time()
for s in range(K):
    t1 = time.time()
    sc.render_into(v, w, l, 1 + s, 2 + s, 1, st)
    print("enqueue ms", (time.time() - t1) * 1e3)
t2 = time.time()
torch.cuda.synchronize()
print("total ms", (time.time() - t0) * 1e3, "enqueue total", (t2 - t0) * 1e3)
print(sc.timings())
print(sc.counters())
