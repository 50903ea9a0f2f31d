#!/usr/bin/env python3
"""What would ordering the walks of a round by where they are and where they point buy the per-lane traversal kernels?
The same set of incoherent queries (rays and cones from random interior points of the bench geometry) is traced in random order and
sorted by keys of increasing cost: Mqueries/s of each order.  usage (GPU box): python tools/bench_coherence.py [n]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
sc = Scene("cornell_box", res=16, mesh_detail=1, lut=(32, 32))
sc.upload(0)
dev = torch.device("cuda", 0)
lib = load_library()
rng = np.random.default_rng(7)
o = rng.uniform(-.009, .009, (n, 3)).astype(np.float32)   # (the box of the bench geometry: 2 cm wide, floor at y = 0)
o[:, 1] += .01
d = rng.normal(size=(n, 3))
d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)


def part1by2(x):
    x = x.astype(np.uint32) & 0x3FF
    x = (x | (x << 16)) & 0x030000FF
    x = (x | (x << 8)) & 0x0300F00F
    x = (x | (x << 4)) & 0x030C30C3
    x = (x | (x << 2)) & 0x09249249
    return x


def morton(q):
    return part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)


def keys(bits_o, bits_d):
    qo = np.clip(((o - o.min(0)) / (o.max(0) - o.min(0)) * (1 << bits_o)).astype(np.int64), 0, (1 << bits_o) - 1)
    ko = morton(qo).astype(np.uint64)
    octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << 1) | ((d[:, 2] < 0).astype(np.uint64) << 2))
    if bits_d:
        qd = np.clip(((np.abs(d)) * (1 << bits_d)).astype(np.int64), 0, (1 << bits_d) - 1)
        kd = (octant << np.uint64(3 * bits_d)) | morton(qd).astype(np.uint64)
        nd = 3 + 3 * bits_d
    else:
        kd, nd = octant, 3
    return ko, kd, 3 * bits_o, nd


orders = {"random": np.arange(n)}
for bo, bd in ((3, 0), (5, 0), (5, 2), (7, 3)):
    ko, kd, no, nd = keys(bo, bd)
    orders[f"origin{bo}b>dir{bd}b"] = np.argsort((ko << np.uint64(nd)) | kd, kind="stable")
    orders[f"dir{bd}b>origin{bo}b"] = np.argsort((kd << np.uint64(no)) | ko, kind="stable")


def time_rays(idx):
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 6], rays[:, 7] = o[idx], d[idx], 0, np.inf
    d_rays = torch.from_numpy(rays).to(dev)
    dist = torch.zeros(n, dtype=torch.float32, device=dev)
    tuid = torch.zeros(n, dtype=torch.int32, device=dev)
    bary = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    front = torch.zeros(n, dtype=torch.int32, device=dev)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.time()
        _check(lib.wtgpu_trace_rays(sc.handle, None, d_rays.data_ptr(), n, dist.data_ptr(), tuid.data_ptr(), bary.data_ptr(), front.data_ptr()))
        torch.cuda.synchronize()
        best = min(best, time.time() - t)
    return n / best / 1e6, float(torch.nan_to_num(dist, posinf=0.).double().sum().item())


def time_cones(idx, tan_alpha):
    c = np.zeros((n, 10), np.float32)
    c[:, :3], c[:, 3:6] = o[idx], d[idx]
    c[:, 6], c[:, 7], c[:, 8], c[:, 9] = tan_alpha, 1e-5, 0., 5.5e-7
    d_c = torch.from_numpy(c).to(dev)
    dist = torch.zeros(n, dtype=torch.float32, device=dev)
    flags = torch.zeros(n, dtype=torch.int32, device=dev)
    ntris = torch.zeros(n, dtype=torch.int32, device=dev)
    tris = torch.zeros((n, 64), dtype=torch.int32, device=dev)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.time()
        _check(lib.wtgpu_traverse_cones(sc.handle, None, d_c.data_ptr(), n, 64, dist.data_ptr(), flags.data_ptr(), ntris.data_ptr(), tris.data_ptr()))
        torch.cuda.synchronize()
        best = min(best, time.time() - t)
    return n / best / 1e6, float(ntris.double().sum().item())


print(f"{n} queries, {int(sc.info.n_tris)} triangles")
for name, idx in orders.items():
    r, chk = time_rays(idx)
    c1, k1 = time_cones(idx, 1e-3)
    c2, k2 = time_cones(idx, 2e-2)
    print(f"{name:22s} rays {r:8.1f} M/s   cones(tan 1e-3) {c1:7.1f} M/s   cones(tan 2e-2) {c2:7.1f} M/s   [checks {chk:.6g} {k1:.0f} {k2:.0f}]", flush=True)
