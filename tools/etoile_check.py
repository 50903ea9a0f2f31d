"""GPU: film sum and event counters of the full-size etoile render (tests/test_gpu_render.py::test_full_size_etoile_720) under the current environment knobs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wave_tracer_amd import Scene
from wave_tracer_amd.render import alloc_films
sc = Scene("etoile", res=720, mesh_detail=2)
sc.upload(0, 720 * 540)
dev = torch.device("cuda", 0)
f = alloc_films(sc, dev)
sc.reset_counters()
sc.render_into(*f, 0, 2, 5, torch.cuda.current_stream(dev).cuda_stream)
torch.cuda.synchronize(dev)
c = sc.counters()
print(sys.argv[1] if len(sys.argv) > 1 else "", "light sum %.9f" % f[2].sum().item(), {k: c[k] for k in ("segments", "fsd_interactions", "light_splats", "shadow_rays", "ray_queries", "cone_queries", "cone_tri_overflow", "traversal_stack_dropped")})
