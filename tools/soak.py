import sys, time
sys.path.insert(0, '/root/repo')
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from wave_tracer_amd import Scene
from wave_tracer_amd.render import alloc_films
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sc = Scene("cornell_box", res=1440, mesh_detail=1)
sc.upload(0, 1440 * 1440)
dev = torch.device("cuda", 0)
v, w, l = alloc_films(sc, dev)
st = torch.cuda.current_stream(dev).cuda_stream
t = time.time()
sc.render_into(v, w, l, 0, spp, 99, st)     # one call, spp passes pipelined inside
torch.cuda.synchronize()
dt = time.time() - t
c = sc.counters()
print(f"{spp} spp in {dt:.2f}s = {1440*1440*spp/dt/1e6:.2f} Msamples/s")
print("finite:", bool(torch.isfinite(v).all() and torch.isfinite(w).all() and torch.isfinite(l).all()), "min weight", float(w.min()), "max weight", float(w.max()))
print({k: c[k] for k in ("samples", "walk_iteration_cap_hits", "fsd_pool_overflow", "fsd_edge_overflow", "edge_overflow", "cone_tri_overflow")})
img = (v / w.clamp_min(1e-30).unsqueeze(-1) + l / spp)
print("mean rgb", img.mean(dim=(0, 1)).tolist(), "max", float(img.max()))
