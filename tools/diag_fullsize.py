#!/usr/bin/env python3
"""Sample-for-sample comparison of the GPU render with the CPU checker on every n-th 24x24 block of the full-size cornell film
(what tests/test_gpu_render.py::test_full_size_properties_1440 asserts), for A/B of device-code variants and knobs: prints the
fraction of compared pixels that agree to 1e-3.  The checker's tiles are cached in /tmp between calls of one gpurun session.
usage: diag_fullsize.py [label]"""
import os
import sys
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))   # the runtime's kernel-argument ring per stream (default 1 MiB: a full ring blocks the enqueueing thread)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from wave_tracer_amd import Scene
from wave_tracer_amd.render import alloc_films
from oracle_util import oracle_render_tiles
label = sys.argv[1] if len(sys.argv) > 1 else "run"
stride = int(os.environ.get("DIAG_STRIDE", "97"))
sc = Scene("cornell_box", res=1440, mesh_detail=1)
sc.upload(0, 1440 * 1440)
dev = torch.device("cuda", 0)
v, w, l = alloc_films(sc, dev)
sc.reset_counters()
sc.render_into(v, w, l, 0, 1, 5, torch.cuda.current_stream(dev).cuda_stream)
torch.cuda.synchronize(dev)
c = sc.counters()
cache = f"/tmp/diag_tiles_{stride}.npz"
if os.path.exists(cache):
    z = np.load(cache)
    ov, ow, mask = z["ov"], z["ow"], z["mask"]
else:
    ov, ow, ol, oc, n, mask = oracle_render_tiles(sc, 0, 1, 5, stride)
    np.savez(cache, ov=ov, ow=ow, mask=mask)
inner = mask.copy()
inner[1:, :] &= mask[:-1, :]
inner[:-1, :] &= mask[1:, :]
inner[:, 1:] &= mask[:, :-1]
inner[:, :-1] &= mask[:, 1:]
inner[0, :] = inner[-1, :] = inner[:, 0] = inner[:, -1] = False
gv, gw = v.cpu().numpy(), w.cpu().numpy()
rel = np.abs(gv[inner] - ov[inner]).sum() / np.abs(ov[inner]).sum()
same = np.abs(gv[inner] - ov[inner]).sum(axis=1) <= 1e-3 * np.abs(ov[inner]).sum(axis=1) + 1e-30
print(f"{label:24s} compared {int(inner.sum())} rel L1 {rel:.3e} frac_same {same.mean():.5f} differing {int((~same).sum())} segments/sample {c['segments'] / c['samples']:.4f} "
      f"cone_q {c['cone_queries'] / c['samples']:.4f} fsd {c['fsd_interactions']}")
if os.environ.get("DIAG_SAVE"):
    np.save(os.environ["DIAG_SAVE"], gv[inner])
