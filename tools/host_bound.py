"""Is a pass bound by the host's enqueue rate?  Times wtgpu_render_async (returns when everything is enqueued) against the pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wave_tracer_amd import Scene
from wave_tracer_amd.render import alloc_films
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1440
kw = dict(mesh_detail=2) if name != "cornell_box" else {}
sc = Scene(name, res=res, **kw)
sc.upload(0, sc.width * sc.height)
dev = torch.device("cuda", 0)
films = alloc_films(sc, dev)
st = torch.cuda.current_stream(dev).cuda_stream
for i in range(2):
    sc.render_into(*films, i, i + 1, 5, st)
torch.cuda.synchronize(dev)
te, tt = [], []
for i in range(2, 8):
    t0 = time.perf_counter()
    sc.render_async_into(*films, i, i + 1, 5, st)
    t1 = time.perf_counter()
    sc.join(st)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    te.append((t1 - t0) * 1e3)
    tt.append((t2 - t0) * 1e3)
print("%s %d: enqueue %.1f ms, pass %.1f ms (median of 6)  TIMING=%s STREAMS=%s" % (name, res, sorted(te)[3], sorted(tt)[3], os.environ.get("WTGPU_TIMING", "1"), os.environ.get("WTGPU_STREAMS", "default")))
