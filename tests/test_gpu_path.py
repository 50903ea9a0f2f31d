"""GPU parity tests of the plt_path integrator + UTD free-space diffraction: the HIP path (through the
C-ABI) against the CPU checker on identical seeded inputs, and against the closed forms of tests/test_path_oracle.py."""
import math

import numpy as np
import pytest

import parity
from test_gpu_render import _both, _rel_l1

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,res,spp,kw,tol", [
    # forward transport, point transmitter, UTD diffraction at the blocks' edges, NEE + sensing splats into the coverage sensor
    ("etoile", 96, 16, {"mesh_detail": 0}, 2e-2),
    ("etoile", 64, 8, {"mesh_detail": 1}, 2e-2),
    ("etoile", 64, 16, {"mesh_detail": 0, "fsd": 0}, 1e-2),
    ("etoile_open", 48, 32, {"mesh_detail": 0}, 1e-2),
    # backward transport: NEE with the power heuristic, emission, Russian roulette, block splats
    ("white_furnace_path", 24, 8, {}, 1e-2),
    ("furnace_path", 32, 4, {}, 1e-2),
    ("furnace_path", 24, 4, {"fsd": 1}, 2e-2),
    ("cornell_box_path", 24, 4, {"mesh_detail": 0, "crop_of": 1440}, 2e-2),
    # the dense bench geometry (283 K triangles) under plt_path: regions beyond the 64-triangle list take the primary from the axis hit
    # and their edge set from bvh_gather_edges (wt/path.h)
    ("cornell_box_path", 24, 2, {"mesh_detail": 1, "crop_of": 1440}, 3e-2),
])
def test_path_image_parity(built, name, res, spp, kw, tol):
    """Same Philox streams on both sides: the images agree sample for sample up to fp contraction / libm ulps (a handful of
    samples take a different discrete branch).  Tolerance: relative L1 of the developed image (fp32 arithmetic, f64 film)."""
    sc, gpu, cpu, gc, oc, gf, cf = _both(name, res, spp, 5, **kw)
    assert np.isfinite(gpu).all() and (gpu >= 0).all()
    assert cpu.sum() > 0
    assert np.allclose(gf[1], cf[1], rtol=1e-5, atol=1e-6)
    parity.check(f"gpu_path/{name}-{res}-{spp}-{sorted(kw.items())}", _rel_l1(gpu, cpu), tol)
    for key in ("segments", "connections", "surface_interactions", "null_interactions", "light_splats", "shadow_rays"):
        # (the bounded device lists — 64 triangles, 32 wedges per aperture — change a few apertures of the widest beams: DESIGN.md §5)
        assert abs(gc[key] - oc[key]) <= 1.5e-2 * max(100, oc[key]), (key, gc[key], oc[key])
    assert abs(gc["fsd_interactions"] - oc["fsd_interactions"]) <= 1e-2 * max(50, oc["fsd_interactions"])
    assert gc["walk_iteration_cap_hits"] == 0


def test_path_forward_free_space_closed_form_on_gpu(built):
    """E = I0 cos(theta) / (pi r^2) over bare ground (see tests/test_path_oracle.py), at a sample count the CPU checker would
    need minutes for."""
    from wave_tracer_amd import Scene, render, develop
    res, spp = 96, 512
    sc = Scene("etoile_open", res=res, mesh_detail=0)
    v, w, l = render(sc, spp, seed=3, device=0)
    img = develop(sc, v, w, l, spp).astype(np.float64)[..., 0]
    H, W = img.shape
    ex, ey = 840.0 / W, 630.0 / H
    tx, ty, h = 80.1, 193.8, 21.0 - 1e-3
    u = (np.arange(8) + .5) / 8
    expected = np.zeros_like(img)
    for y in range(H):
        for x in range(W):
            wx = -420.0 + (x + u[None, :]) * ex
            wy = 315.0 - (y + u[:, None]) * ey
            r2 = (wx - tx) ** 2 + (wy - ty) ** 2 + h * h
            expected[y, x] = np.mean(h / r2 ** 1.5) / math.pi
    assert abs(img.sum() / expected.sum() - 1) < 0.01
    ys, xs = np.mgrid[0:H, 0:W]
    rr = np.hypot(-420.0 + (xs + .5) * ex - tx, 315.0 - (ys + .5) * ey - ty)
    for lo, hi in [(0, 30), (30, 80), (80, 200), (200, 500)]:
        m = (rr >= lo) & (rr < hi)
        assert abs(img[m].sum() / expected[m].sum() - 1) < 0.03


def test_path_additivity_and_batching(built):
    """Samples are independent: rendering [0,8) in one call == [0,4) + [4,8) accumulated, also when the batch is split over
    several state slices (max_batch smaller than the film)."""
    import torch
    from wave_tracer_amd import Scene
    from wave_tracer_amd.render import alloc_films
    sc = Scene("etoile", res=64, mesh_detail=0)
    sc.upload(0, 1000)   # 3072 pixels: several batches per pass
    dev = torch.device("cuda", 0)
    a = alloc_films(sc, dev)
    b = alloc_films(sc, dev)
    sc.render_into(*a, 0, 8, 11)
    sc.render_into(*b, 0, 4, 11)
    sc.render_into(*b, 4, 8, 11)
    torch.cuda.synchronize()
    la, lb = a[2].cpu().numpy(), b[2].cpu().numpy()
    assert la.sum() > 0
    assert np.abs(la - lb).sum() <= 1e-9 * np.abs(la).sum()
