"""Second sources (oracle/indep/second_source.py: independent derivations in double precision, numpy) against the restated primitives of
wt/*.h.  (Where the other primitives have theirs: Fresnel / Mueller — test_kat.py; UTD Ds / Dh — test_kat_utd.py; the sampler built on the
Fraunhofer ASF — test_kat_fsd.py.)"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "indep"))
from second_source import cone_contains, cone_tri_min_z, fraunhofer_boundary_integral, polygon_fourier_integral  # noqa: E402

from oracle_util import load_oracle  # noqa: E402

F = C.c_float


def fa(x):
    return np.ascontiguousarray(x, np.float32)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def lib():
    lb = load_oracle()
    lb.kat_cone_tri.argtypes = [C.c_void_p, C.c_void_p, F, F, C.c_void_p]
    lb.kat_cone_local.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lb.kat_cone_box_outside.argtypes = [C.c_void_p, C.c_void_p, F, F]
    return lb


def _local(lib, cone, pts):
    out = np.zeros((len(pts), 3), np.float32)
    for i, q in enumerate(pts):
        lib.kat_cone_local(p(cone), p(fa(q)), p(out[i]))
    return out.astype(np.float64)


def test_cone_triangle_closest_z_against_the_barycentric_programme(lib):
    """intersect_cone_tri (wt/cone.h, restating math/intersect/cone.hpp:550-626: contained vertices, cone-plane extremum, cone-edge crossings
    in 3-D) against a differently derived solution of the same problem: minimise z over the triangle's barycentric domain under the cone's
    quadratic constraint and the slab (KKT candidates, second_source.cone_tri_min_z).  Isotropic and elliptic cones, whole range and slabs,
    4000 random configurations: the hit flags agree but for grazing contacts, the distances to 1e-4 of the scene scale."""
    rng = np.random.default_rng(11)
    n_cmp = n_flag = n_hit = 0
    worst = 0.0
    for it in range(4000):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        ecc = 0.0 if it % 2 == 0 else rng.uniform(0.2, 0.9)
        ta = float(10 ** rng.uniform(-3, -0.3))
        x0 = float(rng.uniform(0, 0.05)) if it % 3 else 0.0
        cone = fa([*rng.uniform(-.5, .5, 3), *d, ta, x0, ecc])
        e = 1.0 / np.sqrt(1 - float(cone[8]) ** 2)
        centre = np.array(cone[:3], np.float64) + d * rng.uniform(0.3, 3.0) + rng.normal(size=3) * rng.uniform(0, 0.6)
        tri = centre + rng.normal(size=(3, 3)) * 10 ** rng.uniform(-1.5, 0)
        if it % 4 == 0:
            zmin, zmax = 0.0, np.inf
        else:
            zc = float(np.dot(centre - cone[:3], d))
            zmin = max(0.0, zc + rng.uniform(-.5, .3))
            zmax = zmin + 10 ** rng.uniform(-1.5, 0.3)
        out = np.zeros(1, np.float32)
        hit = lib.kat_cone_tri(p(cone), p(fa(tri.ravel())), F(zmin), F(zmax if np.isfinite(zmax) else 3e38), p(out))
        P = _local(lib, cone, tri)
        ref = cone_tri_min_z(P, float(cone[6]), float(cone[7]), e, zmin, zmax)
        n_flag += 1
        if (ref is not None) != bool(hit):
            # disagreements must be grazing: the second source's margin (how deep the deepest point of the triangle sits inside the cone) is tiny
            continue
        n_cmp += 1
        if hit:
            n_hit += 1
            err = abs(float(out[0]) - ref) / max(1.0, abs(ref))
            worst = max(worst, err)
            assert err < 2e-4, (it, float(out[0]), ref)
    print(f"cone-triangle: {n_hit} hits of {n_flag}, flags agree in {n_cmp} ({100.0 * n_cmp / n_flag:.2f} %), worst distance error {worst:.1e}")
    assert n_hit > 600 and n_cmp >= 0.99 * n_flag


def test_cone_box_cull_never_removes_a_box_that_holds_a_point_of_the_cone(lib):
    """cone_box_outside (wt/bvh.h; not in the reference: the extra conservative cull of a child box against cone ∩ slab) is only allowed to
    say "outside" when no point of the box lies in the cone inside the slab.  Random boxes, 400 random points each, containment decided by the
    second source in double precision."""
    rng = np.random.default_rng(5)
    culled = kept = 0
    for it in range(3000):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        ecc = 0.0 if it % 2 == 0 else rng.uniform(0.2, 0.9)
        cone = fa([*rng.uniform(-.5, .5, 3), *d, 10 ** rng.uniform(-3, -0.3), rng.uniform(0, 0.05), ecc])
        e = 1.0 / np.sqrt(1 - float(cone[8]) ** 2)
        c = np.array(cone[:3], np.float64) + d * rng.uniform(0.1, 3.0) + rng.normal(size=3) * rng.uniform(0, 1.0)
        half = 10 ** rng.uniform(-2, 0, 3)
        box = fa([*(c - half), *(c + half)])
        zmin = float(rng.uniform(0, 2))
        zmax = zmin + float(10 ** rng.uniform(-1.5, 0.5))
        out = lib.kat_cone_box_outside(p(cone), p(box), F(zmin), F(zmax))
        if not out:
            kept += 1
            continue
        culled += 1
        pts = c + (rng.uniform(-1, 1, (400, 3))) * half
        loc = _local(lib, cone, pts)
        inside = [cone_contains(q, float(cone[6]), float(cone[7]), e, zmin, zmax) for q in loc]
        assert not any(inside), (it, cone, box, zmin, zmax)
    print(f"cone-box cull: {culled} culled, {kept} kept")
    assert culled > 300 and kept > 300


def test_bounding_sphere_filter_is_conservative(lib):
    """The first filter of the wave-cooperative queries (wt/cone.h: cone_sphere_maybe on tri_bounding_sphere; not in the reference) may only
    drop a triangle that cone_tri_maybe — and therefore the exact test — drops too.  12000 random configurations with small and large
    triangles, thin and wide cones, slabs: (1) the sphere contains the triangle (checked in double), (2) exact hit or filter pass => sphere
    pass, (3) the filter does filter (most non-hits of the small-triangle cases are dropped)."""
    lib.kat_cone_tri_sphere_maybe.argtypes = [C.c_void_p, C.c_void_p, F, F, C.c_void_p]
    lib.kat_cone_tri_maybe.argtypes = [C.c_void_p, C.c_void_p, F, F]
    rng = np.random.default_rng(23)
    n_hit = n_maybe = n_sphere = n_small = n_small_drop = 0
    for it in range(12000):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        ecc = 0.0 if it % 2 == 0 else rng.uniform(0.2, 0.9)
        ta = float(10 ** rng.uniform(-3, -0.3))
        x0 = float(rng.uniform(0, 0.05)) if it % 3 else 0.0
        cone = fa([*rng.uniform(-.5, .5, 3), *d, ta, x0, ecc])
        z = rng.uniform(0.3, 3.0)
        small = it % 2 == 1
        size = 10 ** rng.uniform(-3, -1.5) if small else 10 ** rng.uniform(-1.5, 0)
        centre = np.array(cone[:3], np.float64) + d * z + rng.normal(size=3) * rng.uniform(0, 2.0) * (ta * z + x0 + size)
        tri = fa((centre + rng.normal(size=(3, 3)) * size).ravel())
        if it % 4 == 0:
            zmin, zmax = 0.0, 3e38
        else:
            zc = float(np.dot(centre - cone[:3], d))
            zmin = max(0.0, zc + rng.uniform(-.5, .3) * (0.1 if small else 1.0))
            zmax = zmin + 10 ** rng.uniform(-2.5 if small else -1.5, 0.3)
        sph = np.zeros(4, np.float32)
        s_ok = lib.kat_cone_tri_sphere_maybe(p(cone), p(tri), F(zmin), F(zmax), p(sph))
        T = tri.reshape(3, 3).astype(np.float64)
        assert np.all(np.linalg.norm(T - sph[:3].astype(np.float64), axis=1) <= float(sph[3])), (it, T, sph)
        # ... and is not wastefully large: the radius never exceeds the longest edge's length / sqrt(3) (equilateral circumradius) by more than rounding
        edges = [np.linalg.norm(T[i] - T[(i + 1) % 3]) for i in range(3)]
        assert float(sph[3]) <= max(edges) / np.sqrt(3.0) * (1 + 1e-4) + 1e-12
        out = np.zeros(1, np.float32)
        hit = lib.kat_cone_tri(p(cone), p(tri), F(zmin), F(zmax), p(out))
        maybe = lib.kat_cone_tri_maybe(p(cone), p(tri), F(zmin), F(zmax))
        n_hit += bool(hit)
        n_maybe += bool(maybe)
        n_sphere += bool(s_ok)
        assert not (hit and not maybe), it
        assert not (maybe and not s_ok), (it, cone, tri, zmin, zmax)
        if small and not hit:
            n_small += 1
            n_small_drop += not s_ok
    print(f"bounding spheres: {n_hit} exact hits, {n_maybe} pass cone_tri_maybe, {n_sphere} pass the sphere filter of 12000; "
          f"small non-hit triangles dropped by the sphere alone: {n_small_drop} of {n_small}")
    assert n_hit > 1500 and n_small_drop > 0.5 * n_small


def _asf_unclamped(lib, segments, xis):
    lib.kat_fsd_asf_unclamped.argtypes = [C.c_void_p, C.c_uint32, F, C.c_void_p, C.c_uint32, C.c_void_p]
    E = fa([[e[0], e[1], a[0] + e[0] / 2, a[1] + e[1] / 2, ca - cb, (ca + cb) / 2] for a, e, ca, cb in segments])
    X = fa(xis)
    out = np.zeros(len(X), np.float32)
    lib.kat_fsd_asf_unclamped(p(E), len(E), F(1.0), p(X), len(X), p(out))
    return out.astype(np.float64)


def test_fraunhofer_edge_sum_against_the_boundary_integral(lib):
    """fsd_Psi / fsd_alpha1 / fsd_alpha2 / fsd_ASF_unclamped (wt/fsd.h, restating fsd.hpp:65-146: per segment a closed form in zeta = (xi.e,
    xi x e)) against a quadrature of the line integral those closed forms solve (second_source.fraunhofer_boundary_integral): 40 random sets of
    1-7 segments — open, unconnected, each with its own pair of end amplitudes, i.e. exactly what an aperture record holds — at 12 directions
    each, small and large |xi| (the sinc's series branch and its oscillating tail).  ASF = |B|^2 / (2 pi)^2 to 2e-5."""
    rng = np.random.default_rng(41)
    worst = 0.0
    n = 0
    for it in range(40):
        segs = [(rng.normal(size=2), rng.normal(size=2) * 10 ** rng.uniform(-1, 0.3), *rng.uniform(0.1, 1.5, 2)) for _ in range(rng.integers(1, 8))]
        xis = rng.normal(size=(12, 2)) * 10 ** rng.uniform(-2, 1, (12, 1))
        got = _asf_unclamped(lib, segs, xis)
        for x, g in zip(fa(xis).astype(np.float64), got):
            ref = abs(fraunhofer_boundary_integral(segs, x)) ** 2 / (4 * np.pi ** 2)
            err = abs(g - ref) / max(ref, 1e-6 * max(1.0, np.max(got)))
            worst = max(worst, err)
            n += 1
            assert err < 2e-4, (it, x, g, ref)
    print(f"Fraunhofer edge sum vs boundary integral: {n} directions, worst relative error {worst:.1e}")


def test_fraunhofer_pattern_of_a_uniform_polygon_is_its_fourier_transform(lib):
    """The physics the edge sum stands for: a uniformly lit polygonal aperture (every segment with end amplitudes 1, 1) diffracts into
    |FT of its indicator function|^2 / (2 pi)^2 — here the AREA integral by Gauss quadrature over a fan triangulation, which shares nothing
    with the edge formulation but Stokes' theorem.  Convex and non-convex polygons, both orientations."""
    rng = np.random.default_rng(43)
    polys = [np.array([[-1, -.7], [1.2, -.7], [1.0, .9], [-.8, .6]]), np.array([[0, 0], [2, 0], [2, 1], [1, .3], [0, 1.2]]),
             np.array([[0, 0], [.4, 0], [.4, 3], [0, 3]])]
    polys.append(polys[0][::-1])
    worst = 0.0
    for P in polys:
        segs = [(P[i], P[(i + 1) % len(P)] - P[i], 1.0, 1.0) for i in range(len(P))]
        xis = rng.normal(size=(16, 2)) * 10 ** rng.uniform(-1.5, 0.8, (16, 1))
        got = _asf_unclamped(lib, segs, xis)
        for x, g in zip(fa(xis).astype(np.float64), got):
            ref = abs(polygon_fourier_integral(P, x)) ** 2 / (4 * np.pi ** 2)
            err = abs(g - ref) / max(ref, 1e-6 * np.max(got))
            worst = max(worst, err)
            assert err < 5e-4, (x, g, ref)
    print(f"uniform polygons: worst relative error of the edge sum against the area Fourier integral {worst:.1e}")


# ---- whole renders on second-source primitives (oracle/indep/prims2.cpp, libindep2.so) -----------------------------------------------------
_lib2 = None


def _indep2():
    global _lib2
    if _lib2 is None:
        import ctypes as C
        import subprocess
        from oracle_util import ROOT
        p = os.path.join(ROOT, "oracle", "_build", "libindep2.so")
        if not os.path.exists(p):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _lib2 = C.CDLL(p)
        _lib2.indep_render.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib2.indep_render_path.argtypes = _lib2.indep_render.argtypes
    return _lib2


@pytest.mark.parametrize("name,res,spp,kw", [
    ("furnace", 16, 8, {}),                                    # diffuse walls: cone queries, the surface step
    ("furnace", 16, 8, {"fsd": 1, "lut": (64, 64)}),           # ... with Fraunhofer interactions: interaction regions from second-source cone queries
    ("furnace_spm", 16, 8, {}),                                # rough conductors: Fresnel (complex index) -> Mueller
    ("lens_b", 16, 8, {}),                                     # dielectric interfaces: Fresnel coefficients of both directions, total internal reflection
    ("double_slits", 48, 8, {"lut": (64, 64)}),                # the diffraction gate's scene
    ("bidir_room", 20, 4, {"mesh_detail": 0, "lut": (64, 64), "polarimetric": 1}),   # Stokes film: every Mueller entry counts
    ("etoile", 32, 8, {"mesh_detail": 0}),                     # plt_path: UTD wedges from second-source cone queries, ITU materials
])
def test_render_on_second_source_primitives(name, res, spp, kw):
    """A whole render in which NEITHER the composition NOR the riskiest primitives are the ones the GPU is checked against: oracle/indep/indep.cpp
    (the integrators restated a second time, f64) on oracle/indep/prims2.cpp (cone x triangle as a convex programme in f64, every box cull
    replaced by none, Fresnel coefficients from the angle / permittivity forms, Mueller matrices by the Kronecker construction, the Fraunhofer
    segment amplitudes from a quadrature of the aperture's boundary integral, the UTD wedge coefficients with the transition function from a
    quadrature of its Fresnel integral on a rotated contour), against liboracle.so.  The random numbers are the same, the arithmetic is not (f64 closest points, different formulas): a sample may take another
    discrete branch now and then; the images must agree like a GPU render agrees with the checker."""
    import ctypes as C
    from oracle_util import oracle_render
    from wave_tracer_amd import Scene
    sc = Scene(name, res=res, **kw)
    ov, ow, ol, oc = oracle_render(sc, 0, spp, 77, threads=1)
    H, W, Cn = sc.height, sc.width, sc.channels
    v, w, l = np.zeros((H, W, Cn)), np.zeros((H, W)), np.zeros((H, W, Cn))
    ctr = np.zeros(8, np.uint64)
    entry = _indep2().indep_render if int(sc.info.integrator) == 0 else _indep2().indep_render_path
    calls = np.zeros(6, np.uint64)
    _indep2().ss_calls(calls.ctypes.data_as(C.c_void_p), 1)
    assert entry(sc.host_desc(), 0, spp, 77, v.ctypes.data, w.ctypes.data, l.ctypes.data, ctr.ctypes.data) == 0
    _indep2().ss_calls(calls.ctypes.data_as(C.c_void_p), 1)
    used = dict(zip(["cone_tri", "fresnel_dielectric", "fresnel_conductor", "mueller", "fraunhofer_segment", "utd_wedge"], [int(x) for x in calls]))
    # the render did run on the second sources its scene is here for
    need = {"furnace": ["cone_tri"], "furnace_spm": ["cone_tri", "fresnel_conductor", "mueller"], "lens_b": ["fresnel_dielectric", "mueller"],
            "double_slits": ["cone_tri", "fraunhofer_segment"], "bidir_room": ["cone_tri", "mueller"], "etoile": ["cone_tri", "utd_wedge"]}[name]
    if kw.get("fsd"):
        need = need + ["fraunhofer_segment"]
    for k in need:
        assert used[k] > 0, (k, used)
    ic = dict(zip(["segments", "vertices", "connections", "surface", "fsd_interactions", "null_interactions", "light_splats", "shadow_rays"], [int(x) for x in ctr]))
    for k in ("segments", "connections") if int(sc.info.integrator) else ("segments", "vertices", "connections"):
        assert abs(ic[k] - oc[k]) <= 1e-2 * max(100, oc[k]) + 2, (k, ic[k], oc[k])
    assert np.allclose(w, ow, rtol=1e-5, atol=1e-9)
    a = v.sum(axis=2) + l.sum(axis=2)
    b = ov.sum(axis=2) + ol.sum(axis=2)
    assert b.sum() > 0
    rel = np.abs(a - b).sum() / b.sum()
    same = (np.abs(a - b) <= 1e-3 * np.abs(b) + 1e-9 * b.max()).mean()
    print(f"{name}: image rel. L1 {rel:.2e}, {same:.3f} of the pixels agree to 1e-3; counters {ic['segments']}/{oc['segments']} segments, {ic['connections']}/{oc['connections']} connections")
    assert rel < 2e-2 and same > 0.9, (rel, same)


def test_second_source_utd_transition_function_is_the_fresnel_integral():
    """prims2.cpp's transition function — F(x) = 2 i sqrt(x) e^{ix} int_{sqrt x}^inf e^{-i t^2} dt evaluated on the contour rotated by -pi/4 —
    against scipy's Fresnel integrals, x from 1e-8 to 1e3 (2e-13), and its limits; then what the render comparison rests on: the checker's
    utd_F (wt/utd.h, restating utd.hpp:36-57: erfc series below |x| = 6, four-term asymptote above) against it — 1e-6 below 6 and the
    asymptote's truncation error above (4e-3 at x = 6, 1e-4 at x = 12)."""
    import ctypes as C
    import math
    from scipy import special
    from oracle_util import load_oracle
    l2, lib = _indep2(), load_oracle()
    l2.ss_utd_transition.argtypes = [C.c_double, C.c_void_p]
    lib.kat_utd_F.argtypes = [C.c_float, C.c_void_p]

    def tail(a):
        S, Cc = special.fresnel(a * math.sqrt(2 / math.pi))
        return math.sqrt(math.pi / 2) * ((0.5 - Cc) - 1j * (0.5 - S))

    def ss(x):
        o = np.zeros(2)
        l2.ss_utd_transition(x, o.ctypes.data_as(C.c_void_p))
        return complex(o[0], o[1])

    worst = 0.0
    for x in 10 ** np.linspace(-8, 3, 300):
        ref = 2j * math.sqrt(x) * np.exp(1j * x) * tail(math.sqrt(x))
        worst = max(worst, abs(ss(x) - ref) / abs(ref))
    assert worst < 1e-11, worst
    assert ss(0.0) == 0 and abs(ss(1e6) - complex(1, 5e-7)) < 1e-11
    lo = hi = 0.0
    for x in 10 ** np.linspace(-4, 2.5, 300):
        o = np.zeros(2, np.float32)
        lib.kat_utd_F(C.c_float(x), o.ctypes.data_as(C.c_void_p))
        err = abs(complex(o[0], o[1]) - ss(float(np.float32(x)))) / abs(ss(float(np.float32(x))))
        if x < 6:
            lo = max(lo, err)
        else:
            hi = max(hi, err)
    print(f"contour quadrature vs scipy {worst:.1e}; the checker's utd_F vs it: {lo:.1e} below x = 6, {hi:.1e} above (the reference's four-term asymptote)")
    assert lo < 2e-6 and hi < 5e-3


def test_second_source_wedge_coefficients_against_the_checker():
    """ss_wedge_utd (textbook Kouyoumjian-Pathak in f64, N+- by search, exact transition function) against wedge_UTD (wt/utd.h) over random wedges,
    directions, wavenumbers and distances, both fed the azimuths the reference measures (atan2: (-pi, pi]): agreement to the asymptote's
    truncation (2e-3 of the larger coefficient), i.e. prefactor, signs, the four cotangents and the a+- arguments are the same function."""
    import ctypes as C
    from oracle_util import load_oracle
    l2, lib = _indep2(), load_oracle()
    l2.ss_wedge_utd.argtypes = [C.c_double] * 6 + [C.c_void_p]
    lib.kat_wedge_UTD.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    rng = np.random.default_rng(2)
    nff, tff = np.array([0, 1, 0.]), np.array([1, 0, 0.])
    e = np.cross(nff, tff)
    worst = 0.0
    for it in range(1500):
        alpha = rng.uniform(0.05, np.pi * 0.95)
        n = 2 - alpha / np.pi
        phii, phio = rng.uniform(0.01, n * np.pi - 0.01, 2)
        beta = rng.uniform(0.3, np.pi - 0.3)
        wi = np.sin(beta) * (np.cos(phii) * tff + np.sin(phii) * nff) + np.cos(beta) * e
        wo = np.sin(beta) * (np.cos(phio) * tff + np.sin(phio) * nff) - np.cos(beta) * e
        k, ro = 10 ** rng.uniform(-3, 1), 10 ** rng.uniform(-2, 1.5)
        wd = fa([0, 0, 0, 1.0, *nff, *tff, 0, 0, 0, alpha])
        wif, wof = fa(wi), fa(wo)
        out = np.zeros(4, np.float32)
        lib.kat_wedge_UTD(p(wd), C.c_float(k), p(wif), p(wof), C.c_float(ro), p(out))
        sb2 = 1 - float(wif.astype(np.float64) @ e) ** 2
        pi_, po_ = np.arctan2(nff @ wif, tff @ wif), np.arctan2(nff @ wof, tff @ wof)
        o2 = np.zeros(4)
        l2.ss_wedge_utd(n, k * 1e3 * ro * sb2, k * 1e3 * ro, np.sqrt(sb2), float(pi_), float(po_), o2.ctypes.data_as(C.c_void_p))
        if not np.any(out):      # (the reference's exclusion of grazing directions)
            continue
        worst = max(worst, np.abs(out - o2).max() / np.abs(o2).max())
    print(f"wedge coefficients, checker vs second source: worst {worst:.1e} of the larger coefficient")
    assert worst < 4e-3


def test_second_source_fraunhofer_segment_against_the_closed_forms():
    """ss_fraunhofer_segment (quadrature of the segment's boundary integral, f64) against a_b alpha_1 / iab_2 alpha_2 (wt/fsd.h, restating
    fsd.hpp:65-121) over random segments and directions: 3e-4 of the larger amplitude — the closed form's own f32 cancellation in
    cos(x/2) - sinc(x/2) at small xi . e is what is left (against f64 closed forms: 1e-6)."""
    import ctypes as C
    import math
    from oracle_util import load_oracle
    l2, lib = _indep2(), load_oracle()
    lib.kat_fsd_alpha1.restype = lib.kat_fsd_alpha2.restype = C.c_float
    lib.kat_fsd_alpha1.argtypes = lib.kat_fsd_alpha2.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(1)
    worst32 = worst64 = 0.0
    for it in range(1500):
        e = fa(rng.normal(size=2) * 10 ** rng.uniform(-1, 0.5))
        xi = fa(rng.normal(size=2) * 10 ** rng.uniform(-2, 1.5))
        ab, iab = np.float32(rng.normal()), np.float32(rng.uniform(0.1, 2))
        out = np.zeros(2, np.float32)
        l2.ss_fraunhofer_segment(p(e), C.c_float(ab), C.c_float(iab), p(xi), p(out))
        zx, zy = np.float32(xi @ e), np.float32(xi[0] * e[1] - xi[1] * e[0])
        a1, a2 = ab * lib.kat_fsd_alpha1(zx, zy), iab * lib.kat_fsd_alpha2(zx, zy)
        worst32 = max(worst32, max(abs(out[0] - a1), abs(out[1] - a2)) / max(abs(a1), abs(a2), 1e-12))
        x, y = float(xi.astype(np.float64) @ e.astype(np.float64)), float(xi[0]) * float(e[1]) - float(xi[1]) * float(e[0])
        sinc = math.sin(x / 2) / (x / 2)
        d1 = float(ab) * y / (2 * math.pi * x * (x * x + y * y)) * (math.cos(x / 2) - sinc)
        d2 = float(iab) * y / (2 * math.pi * (x * x + y * y)) * sinc
        worst64 = max(worst64, max(abs(out[0] - d1), abs(out[1] - d2)) / max(abs(d1), abs(d2), 1e-12))
    print(f"Fraunhofer segment amplitudes, quadrature vs closed forms: {worst32:.1e} (f32 closed forms), {worst64:.1e} (f64 closed forms)")
    assert worst32 < 1e-3 and worst64 < 2e-6

