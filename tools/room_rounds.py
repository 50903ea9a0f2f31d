import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from wave_tracer_amd import Scene
from wave_tracer_amd.render import alloc_films
sc = Scene("bidir_room", res=1920, mesh_detail=2, polarimetric=1)
npix = sc.width * sc.height
sc.upload(0, npix * 2)
dev = torch.device("cuda", 0)
f = alloc_films(sc, dev)
st = torch.cuda.current_stream(dev).cuda_stream
for i in range(4):
    sc.reset_counters()
    t = time.time()
    sc.render_into(*f, 2 * i, 2 * i + 2, 1, st)
    torch.cuda.synchronize(dev)
    tm = sc.timings()
    print("step", i, "%.1f ms" % ((time.time() - t) * 1e3), {k: tm[k] for k in tm if "round" in k or k in ("batches",)}, "cap hits", sc.counters()["walk_iteration_cap_hits"])
