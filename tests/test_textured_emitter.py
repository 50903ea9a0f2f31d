"""Area emitters with a spatially varying radiance (the reference: include/wt/emitter/area.hpp:103-166, src/emitter/area.cpp:
109-130 and 153-271).  The radiance is scale x a bitmap texture at the surface's uv; positions are drawn from per-triangle tables of the
texture's luminance on a barycentric grid, and every density of a position (NEE, the MIS weights of both integrators) is read back from the
same tables.

(1) the tables the host builds == a numpy restatement of area.cpp:153-216 on the same texture;
(2) sampled densities == evaluated densities, cells are hit in proportion to their mass, black cells never;
(3) the estimators that use the tables (light subpaths, next-event estimation, their MIS weights) against the one that does not (backward
    path tracing without emitter sampling: the lamp is only ever HIT, its tables never read): the same pattern, and the factor the
    reference's density carries — CPU checker;
(4) GPU == CPU checker on the same random numbers, for plt_bdpt and both directions of plt_path."""
import ctypes as C
import math
import os
import sys

import numpy as np
import pytest

from oracle_util import load_oracle, oracle_render

HERE = os.path.dirname(os.path.abspath(__file__))
XML = os.path.join(HERE, "data", "xml", "textured_emitter.xml")
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
F = C.c_float
K = 2 * math.pi / 5.5e-4
f32 = np.float32


@pytest.fixture(scope="module")
def lib(built):
    lib = load_oracle()
    lib.kat_area_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
    lib.kat_area_table.restype = C.c_uint32
    lib.kat_area_samples.argtypes = [C.c_void_p, C.c_int, F, C.c_uint64, C.c_uint32, C.c_void_p]
    return lib


def _table(lib, sc, ei=0):
    n = lib.kat_area_table(C.c_void_p(sc.host_desc()), ei, None, 0)
    out = np.zeros(n, f32)
    assert lib.kat_area_table(C.c_void_p(sc.host_desc()), ei, out.ctypes.data_as(C.c_void_p), n) == n
    return out


def _roundf(x):
    """C's roundf on float32 (half away from zero)."""
    x = f32(x)
    return f32(np.floor(x + f32(.5))) if x >= 0 else -f32(np.floor(-x + f32(.5)))


def _nearest(img, u, v):
    """texture2d nearest lookup with repeat wrap, rows from the top, v up (wt/scene.h tex_bitmap)."""
    h, w = img.shape[:2]
    x = int(_roundf(f32(w) * f32(u) - f32(.5))) % w
    y = int(_roundf(f32(h) * (f32(1) - f32(v)) - f32(.5))) % h
    return img[y, x]


def _restated_tables(img, tris, res_cap=512.0):
    """src/emitter/area.cpp:153-216 in float32: per triangle (positions p[3], uvs uv[3]) the normalised cdf over the barycentric cells,
    {texels, 1 / texels, texel_to_area_density}, and the triangle distribution."""
    res = f32(min(max(img.shape[1], img.shape[0]), res_cap))
    out, powers = [], []
    for p, uv in tris:
        p, uv = np.asarray(p, f32), np.asarray(uv, f32)
        tarea = f32(.5) * f32(np.linalg.norm(np.cross(p[2] - p[0], p[1] - p[0]).astype(f32)))
        d = max(np.linalg.norm(uv[1] - uv[0]), np.linalg.norm(uv[2] - uv[0]), np.linalg.norm(uv[2] - uv[1]))
        texels = int(math.ceil(f32(res * f32(d))))
        step = f32(1) / f32(texels)
        lum = []
        for b in range(texels):
            for a in range(b + 1):
                alpha = f32(f32(a) * step + f32(.5) * step)
                beta = f32(f32(1) - f32(f32(b) * step + f32(.5) * step))
                g = f32(max(f32(0), f32(f32(f32(1) - alpha) - beta)))
                tuv = f32(alpha) * uv[0] + f32(beta) * uv[1] + g * uv[2]
                c = _nearest(img, tuv[0], tuv[1])
                lum.append(max(f32(0), f32(f32(f32(.2126) * c[0] + f32(.7152) * c[1]) + f32(.0722) * c[2])))
        lum = np.asarray(lum, f32)
        cdf = np.concatenate([[f32(0)], np.cumsum(lum, dtype=f32)]).astype(f32)
        I = cdf[-1]
        cdf = (cdf * (f32(1) / I)).astype(f32) if I > 0 else cdf
        out.append((texels, step, f32(1) / (step * step * tarea), cdf))
        powers.append(f32(I / f32(len(lum)) * tarea))
    tc = np.concatenate([[f32(0)], np.cumsum(np.asarray(powers, f32), dtype=f32)]).astype(f32)
    return (tc * (f32(1) / tc[-1])).astype(f32), out


LAMP_TRIS = None


def _lamp_tris():
    """the lamp rectangle of textured_emitter.xml as the reader tessellates it (host/scene_builder.cpp mesh_rectangle)."""
    p, x, y = np.array([-.5, 1.2, -.3]), np.array([1., 0, 0]), np.array([0, -.2, .6])
    v = [p, p + x, p + x + y, p + y]
    uv = [(0, 0), (1, 0), (1, 1), (0, 1)]
    return [([v[i] for i in t], [uv[i] for i in t]) for t in ((0, 1, 2), (2, 3, 0))]


def test_tables_are_the_reference_s_construction(lib):
    from make_lamp_fixture import lamp_texture
    from wave_tracer_amd import Scene
    sc = Scene.from_xml(XML, res=16)
    assert sc.info.n_emitters == 1
    tab = _table(lib, sc)
    tcdf, tris = _restated_tables(lamp_texture(), _lamp_tris())
    T = 2
    assert np.allclose(tab[:T + 1], tcdf, rtol=2e-6, atol=1e-7)
    assert 0.05 < tab[1] < 0.95   # both triangles carry light
    for i, (texels, step, dens, cdf) in enumerate(tris):
        h = tab[T + 1 + 4 * i:T + 1 + 4 * i + 4]
        assert texels == 12 and int(h[0]) == texels and h[1] == step          # ceil(8 texels x sqrt(2) of uv)
        assert np.isclose(h[2], dens, rtol=1e-5)
        off = int(h[3])
        mine = tab[off:off + len(cdf)]
        assert mine[0] == 0 and mine[-1] == 1 and np.all(np.diff(mine) >= 0)
        assert np.allclose(mine, cdf, rtol=2e-6, atol=2e-7), (i, np.abs(mine - cdf).max())
        assert (np.diff(mine) == 0).any() and (np.diff(mine) > 0).any()       # the black band: empty cells
    assert len(tab) == T + 1 + 4 * T + sum(len(t[3]) for t in tris)
    # a 2x finer uv tiling: the working resolution halves per uv unit (transform_t::resolution), the cell count stays
    sc2 = Scene.from_xml(XML, res=16, defines={"mscale": 2})
    t2 = _table(lib, sc2)
    assert int(t2[T + 1]) == 6


def test_sampled_positions_and_their_densities(lib):
    from wave_tracer_amd import Scene
    sc = Scene.from_xml(XML, res=16)
    tab = _table(lib, sc)
    n = 200000
    o = np.zeros((n, 8), f32)
    lib.kat_area_samples(C.c_void_p(sc.host_desc()), 0, F(K), 7, n, o.ctypes.data_as(C.c_void_p))
    tri, alpha, beta, ppd, pdf = o[:, 0].astype(int), o[:, 1], o[:, 2], o[:, 3], o[:, 4]
    assert ((alpha >= 0) & (beta >= 0) & (alpha + beta <= 1 + 1e-6)).all() and (ppd > 0).all()
    # the density evaluated at the sampled surface is the sampled one (a sample exactly on a cell border may read its neighbour)
    same = np.isclose(ppd, pdf, rtol=1e-5)
    assert same.mean() > 0.999, same.mean()
    # triangles and cells are hit in proportion to their mass
    T = 2
    for t in range(T):
        frac, mass = (tri == t).mean(), tab[t + 1] - tab[t]
        assert abs(frac - mass) < 4 * math.sqrt(mass * (1 - mass) / n), (t, frac, mass)
        texels, off = int(tab[T + 1 + 4 * t]), int(tab[T + 1 + 4 * t + 3])
        cells = texels * (texels + 1) // 2
        cell_pdf = np.diff(tab[off:off + cells + 1].astype(np.float64))
        m = tri == t
        a = np.clip(np.floor(alpha[m] * texels), 0, texels - 1).astype(int)
        b = np.clip(np.floor((1 - beta[m]) * texels), 0, texels - 1).astype(int)
        a = np.minimum(a, b)
        hist = np.bincount(b * (b + 1) // 2 + a, minlength=cells) / m.sum()
        assert hist[cell_pdf == 0].sum() < 2e-3                          # black cells: only border roundings
        sig = np.sqrt(np.maximum(cell_pdf * (1 - cell_pdf), 1e-12) / m.sum())
        assert (np.abs(hist - cell_pdf) < 5 * sig + 2e-3).all()
    # The reference's density of a position is TWICE the area density: texel_to_area_density = 1 / (step^2 tarea) (area.cpp:203), where a
    # barycentric cell of side `step` covers step^2 x 2 tarea of the triangle.  Kept, as every other number of the reference is: E[1 / ppd]
    # is half the lit area (the black band is never drawn), not the lit area.
    lit_area = (1.0 / ppd.astype(np.float64)).mean() / (1.0 * math.hypot(.2, .6))
    assert 0.40 < lit_area < 0.5, lit_area
    # ... and the radiance the sampled points see is the texture's (warm blob: mostly bright points are drawn)
    assert (o[:, 7] > 0).mean() > 0.97


def test_a_featureless_bitmap_gives_uniform_tables(lib, tmp_path):
    from wave_tracer_amd import Scene
    from wave_tracer_amd.imageio import write_pfm
    write_pfm(str(tmp_path / "flat.pfm"), np.full((2, 2, 3), .25, f32))
    sc = Scene.from_xml(XML, res=16, defines={"lamp": str(tmp_path / "flat.pfm")})
    o = np.zeros((2000, 8), f32)
    lib.kat_area_samples(C.c_void_p(sc.host_desc()), 0, F(K), 3, 2000, o.ctypes.data_as(C.c_void_p))
    # every cell alike: with n = ceil(2 texels x sqrt 2) = 3 cells a side the density is tpdf x 1 / (n (n + 1) / 2) x n^2 / tarea = 2 n / (n + 1) per
    # lamp area — the uniform density times the reference's factor (two, less the cells that straddle the diagonal)
    area, n = 1.0 * math.hypot(.2, .6), 3
    assert np.allclose(o[:, 3], 2 * n / (n + 1) / area, rtol=2e-5) and np.allclose(o[:, 4], o[:, 3], rtol=2e-5)
    assert np.allclose(o[:, 7], o[0, 7]) and o[0, 7] > 0


def test_the_reference_s_refusals(built):
    from wave_tracer_amd import Scene
    with pytest.raises(Exception, match="mean_spectrum"):
        Scene.from_xml(XML, res=16, defines={"checker": "true"})


def _film(sc, spp, seed):
    from wave_tracer_amd import develop
    v, w, l, c = oracle_render(sc, 0, spp, seed)
    return develop(sc, v, w, l, spp).astype(np.float64), c


def test_emitted_flux_carries_the_reference_s_factor(lib, tmp_path):
    """A one-channel radiance map is wavelength independent: the lamp emits pi x scale x area x mean texel (uv covers the map once, nearest
    lookups).  emitter_sample() weighs a sampled beam by 1 / (ppd dpd) with the tables' ppd — the reference's, (2 n / (n + 1)) x the area
    density on n cells a side — so the mean sampled flux is (n + 1) / (2 n) of that, plus the light the diagonal cells fold onto the edge."""
    from make_lamp_fixture import lamp_texture
    from wave_tracer_amd import Scene
    from wave_tracer_amd.imageio import write_pfm
    grey = lamp_texture().sum(-1).astype(f32)
    write_pfm(str(tmp_path / "grey.pfm"), grey)
    sc = Scene.from_xml(XML, res=16, defines={"lamp": str(tmp_path / "grey.pfm")})
    lib.kat_emitter_mean_flux.argtypes = [C.c_void_p, C.c_int, F, C.c_uint64, C.c_uint32, C.c_void_p]
    lib.kat_emitter_mean_flux.restype = C.c_double
    val = F(0)
    flux = lib.kat_emitter_mean_flux(C.c_void_p(sc.host_desc()), 0, F(K), 5, 400000, C.byref(val))
    emitted = math.pi * 2e-6 * math.hypot(.2, .6) * float(grey.mean())
    assert np.isclose(val.value, 2e-6 * grey.mean(), rtol=1e-5)          # the emitter's own spectrum: the mean texel, flat
    n = 12
    # (cells whose CENTRE falls on the black band are never drawn, whatever light their corners hold: a few per cent less)
    assert 0.88 * (n + 1) / (2 * n) < flux / emitted < 1.05 * (n + 1) / (2 * n), flux / emitted


def test_every_integrator_renders_the_lamp(built):
    """CPU checker: plt_bdpt and backward plt_path (camera), forward plt_path (the plane sensor over the floor) give finite pictures of the
    same pattern."""
    from wave_tracer_amd import Scene
    res = 12
    a, _ = _film(Scene.from_xml(XML, res=res, defines={"integrator": "plt_path", "direction": "backward"}), 256, 3)
    b, cb = _film(Scene.from_xml(XML, res=res, defines={"integrator": "plt_bdpt"}), 192, 5)
    assert np.isfinite(a).all() and np.isfinite(b).all() and a.sum() > 0 and cb["light_splats"] > 0
    assert 0.5 < b.sum() / a.sum() < 1.1
    assert np.corrcoef(a.reshape(res * res, -1).sum(-1), b.reshape(res * res, -1).sum(-1))[0, 1] > 0.9
    pa, _ = _film(Scene.from_xml(XML, res=res, defines={"integrator": "plt_path", "direction": "backward", "plane": "true"}), 128, 3)
    pf, cf = _film(Scene.from_xml(XML, res=res, defines={"integrator": "plt_path", "direction": "forward", "plane": "true"}), 256, 5)
    assert np.isfinite(pf).all() and pf.sum() > 0 and cf["light_splats"] > 0
    # the same broad pool of light under the lamp (4 x 4 blocks: single samples are noisy)
    blocks = lambda x: np.clip(x.reshape(res, res, -1).sum(-1), 0, np.percentile(x.reshape(res * res, -1).sum(-1), 97)).reshape(3, 4, 3, 4).sum((1, 3)).ravel()
    assert np.corrcoef(blocks(pa), blocks(pf))[0, 1] > 0.7


@pytest.mark.gpu
@pytest.mark.parametrize("defs", [{"integrator": "plt_bdpt"}, {"integrator": "plt_path", "direction": "forward", "plane": "true"},
                                  {"integrator": "plt_path", "direction": "backward"}, {"integrator": "plt_bdpt", "filter": "bicubic", "mscale": 2}],
                         ids=["bdpt", "forward", "backward", "bdpt-bicubic-tiled"])
def test_gpu_renders_like_the_checker(built, defs):
    import parity
    from wave_tracer_amd import Scene, develop, render
    sc = Scene.from_xml(XML, res=64, defines=defs)
    spp = 8
    v, w, l = render(sc, spp, seed=17)
    ov, ow, ol, oc = oracle_render(sc, 0, spp, 17)
    gi, oi = develop(sc, v, w, l, spp).astype(np.float64), develop(sc, ov, ow, ol, spp).astype(np.float64)
    assert oi.sum() > 0
    label = "-".join(f"{k}={v}" for k, v in sorted(defs.items()))
    parity.check(f"textured_emitter/{label}", np.abs(gi - oi).sum() / np.abs(oi).sum(), 1e-2)
    c = sc.counters()
    for key in ("segments", "vertices", "connections"):
        assert abs(c[key] - oc[key]) <= 5e-3 * oc[key], (key, c[key], oc[key])
