#!/usr/bin/env python3
"""A/B micro-benchmark of the ADS query kernels (per-lane vs 8-lane-group): Mqueries/s on the bench geometry + parity vs the CPU
checker.  usage (GPU box): python tools/bench_queries.py [n_rays]   — run twice, with and without WTGPU_RAYS_PER_LANE=1"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from wave_tracer_amd import Scene
from wave_tracer_amd.api import load_library, _check
from test_oracle import oracle_trace, random_rays

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
sc = Scene("cornell_box", res=16, mesh_detail=1, lut=(32, 32))
sc.upload(0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
# incoherent rays from random interior points (the pattern of shadow rays / bounced walks)
rays = random_rays(n, 5, -.02, .02)
rays[:, 1] += .01
lib = load_library()
d_rays = torch.from_numpy(rays).to(dev)
dist = torch.zeros(n, dtype=torch.float32, device=dev)
tuid = torch.zeros(n, dtype=torch.int32, device=dev)
bary = torch.zeros((n, 2), dtype=torch.float32, device=dev)
front = torch.zeros(n, dtype=torch.int32, device=dev)
for rep in range(3):
    torch.cuda.synchronize()
    t = time.time()
    _check(lib.wtgpu_trace_rays(sc.handle, None, d_rays.data_ptr(), n, dist.data_ptr(), tuid.data_ptr(), bary.data_ptr(), front.data_ptr()))
    torch.cuda.synchronize()
    dt = time.time() - t
    print(f"mode={'per-lane' if os.environ.get('WTGPU_RAYS_PER_LANE') else 'g8'} rep {rep}: {n / dt / 1e6:.1f} Mrays/s ({dt * 1e3:.1f} ms)")
m = 20000
od, ot, ob, of = oracle_trace(sc, rays[:m])
gd, gt = dist[:m].cpu().numpy(), tuid[:m].cpu().numpy().view(np.uint32)
hit = np.isfinite(od)
print("parity: same hit flag", (np.isfinite(gd) == hit).mean(), "same tri", (gt == ot)[hit].mean(), "max rel dist err",
      np.abs(gd[hit] - od[hit]).max() / np.abs(od[hit]).max(), "hit frac", hit.mean())
