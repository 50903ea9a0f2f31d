"""Dispatching BSDF wrappers on the hot path: `composite` (include/wt/bsdf/composite.hpp: per-wavenumber bins,
left-inclusive ranges, no BSDF outside the bins) and `mask` (src/bsdf/mask.cpp with a constant mask texture), evaluated PER SAMPLE at
the sample's wavenumber (wt/bsdf.h: material_resolve) — not resolved when the scene is baked.  Test scene: the furnace box under an
RGB sensor and a 6000 K blackbody area light, whose samples draw wavenumbers on BOTH sides of the 550 nm bin boundary."""
import numpy as np
import pytest

from oracle_util import oracle_render

RES, SPP = 20, 8


def _render(name, seed=5, spp=SPP, **kw):
    from wave_tracer_amd import Scene, develop
    sc = Scene(name, res=RES, **kw)
    v, w, l, c = oracle_render(sc, 0, spp, seed)
    return develop(sc, v, w, l, spp).astype(np.float64), c


def test_composite_with_identical_bins_equals_the_plain_material(built):
    """Dispatch on either side of the boundary reaches a BSDF that behaves like the plain one: same random numbers, same image bit for
    bit, same event counters."""
    a, ca = _render("furnace_wall_composite_same")
    b, cb = _render("furnace_wall_grey")
    assert np.array_equal(a, b)
    assert ca == cb


def test_composite_bins_select_by_wavenumber(built):
    """Two bins with different albedo (0.8 below 550 nm, 0.2 above) against ONE diffuse BSDF whose reflectance spectrum is that step:
    the same function of wavenumber expressed two ways (the table's step is smeared over one of its 1024 knots: < 0.5 % of the band)."""
    comp, cc = _render("furnace_wall_composite", spp=32)
    step, cs = _render("furnace_wall_step", spp=32)
    grey, _ = _render("furnace_wall_grey", spp=32)
    rel = np.abs(comp - step).sum() / step.sum()
    assert rel < 1e-2, rel
    assert abs(cc["vertices"] - cs["vertices"]) < 5e-3 * cs["vertices"]
    # ... and it is a spectral effect: short wavelengths (blue) bright, long (red) dark, unlike the grey box
    ratio = comp[..., 2].sum() / comp[..., 0].sum()
    assert ratio > 1.5 * grey[..., 2].sum() / grey[..., 0].sum(), ratio
    assert np.abs(comp - grey).sum() / grey.sum() > 0.1


def test_composite_has_no_bsdf_outside_its_bins(built):
    """A single bin below 550 nm: samples with longer wavelengths find no BSDF at the walls (composite.hpp:106-121: f = 0, no sample),
    i.e. the image of a wall that is black above 550 nm; their walks end at the first wall vertex (fewer vertices than the black-wall
    scene, where a zero-weight vertex is still appended)."""
    gap, cg = _render("furnace_wall_composite_gap", spp=32)
    step, cs = _render("furnace_wall_step_gap", spp=32)
    assert np.abs(gap - step).sum() / step.sum() < 1e-2
    assert gap[..., 0].sum() < 0.35 * gap[..., 2].sum()          # red is direct light only
    assert cg["vertices"] < cs["vertices"]


def test_mask_of_full_opacity_equals_the_nested_bsdf(built):
    a, ca = _render("furnace_wall_mask_one")
    b, cb = _render("furnace_wall_grey")
    assert np.array_equal(a, b) and ca == cb


def test_mask_scales_the_nested_bsdf_and_passes_the_rest_through(built):
    """mask(alpha = 0.6) over diffuse(0.8) on the walls of a closed box: what is not reflected leaves the box through the null lobe
    (wo = -wi, discrete) and is lost, so the converged image is that of diffuse(0.48) walls.  Different estimators (stochastic null
    lobe vs a darker BSDF): compared at 256 spp within Monte-Carlo noise."""
    m, cm = _render("furnace_wall_mask", spp=256, seed=11)
    e, ce = _render("furnace_wall_mask_equiv", spp=256, seed=12)
    e2, _ = _render("furnace_wall_mask_equiv", spp=256, seed=13)
    noise = np.abs(e - e2).sum() / e.sum()
    diff = np.abs(m - e).sum() / e.sum()
    print("mask vs equivalent diffuse: rel L1", diff, "noise floor", noise, "means", m.mean() / e.mean())
    assert abs(m.mean() / e.mean() - 1) < 0.02
    assert diff < 2.5 * noise
    plain, _ = _render("furnace_wall_grey", spp=64)
    assert abs(m.mean() / plain.mean() - 1) > 0.02                # and it is not the 0.5 box


def test_composite_bins_are_left_inclusive_in_wavenumber(built):
    """composite.hpp:33-36: ranges are left-inclusive in WAVENUMBER, i.e. a wavelength bin "300nm .. 550nm" is k in [2 pi / 550 nm,
    2 pi / 300 nm): exactly at 550 nm the short-wavelength bin (albedo 0.8) answers, one ulp below in k the long-wavelength one (0.2);
    outside 300 .. 800 nm no BSDF (no sample)."""
    import ctypes as C
    import math
    from oracle_util import load_oracle
    from wave_tracer_amd import Scene
    lib = load_oracle()
    F = C.c_float
    lib.kat_material_sample_consistency.argtypes = [C.c_void_p, C.c_int, C.c_void_p, F, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p]
    sc = Scene("furnace_wall_composite", res=8)
    h = C.c_void_p(sc.host_desc())
    wi = np.ascontiguousarray([0.3, 0.2, 0.9], np.float32)
    wi /= np.linalg.norm(wi)

    def albedo(k):
        o = np.zeros((8, 5), np.float32)
        lib.kat_material_sample_consistency(h, 2, wi.ctypes.data_as(C.c_void_p), F(k), 0, 3, 8, o.ctypes.data_as(C.c_void_p))
        return o[:, 2] / o[:, 0] if (o[:, 0] > 0).all() else None     # (weight x density) / density: a Lambertian sample's weight is its reflectance; None: no sample

    kb = np.float32(2 * math.pi / 550e-6)
    assert np.allclose(albedo(float(kb)), 0.8)
    assert np.allclose(albedo(float(np.nextafter(kb, np.float32(0)))), 0.2)
    assert np.allclose(albedo(2 * math.pi / 400e-6), 0.8) and np.allclose(albedo(2 * math.pi / 700e-6), 0.2)
    assert albedo(2 * math.pi / 900e-6) is None and albedo(2 * math.pi / 250e-6) is None
