"""Known-answer tests of the Gaussian surface profile (include/wt/interaction/surface_profile/gaussian.hpp:25-255; SURVEY.md §8
row a10) as restated in wave_tracer_amd/csrc/wt/bsdf.h.  CPU only."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle_util import load_oracle

F = C.c_float


@pytest.fixture(scope="module")
def lib(built):
    lib = load_oracle()
    lib.kat_gaussian.restype = F
    lib.kat_gaussian.argtypes = [C.c_int, F, F, F, C.c_void_p, C.c_void_p]
    lib.kat_gaussian_sample.argtypes = [F, F, F, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    return lib


def fa(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


K = 2 * math.pi / 5.5e-4   # 550 nm in 1/mm


def test_gaussian_psd_formula_and_normalisation(lib):
    sigma = 0.25 * K      # explicit rms: s2 = sigma^2/k^2 = 1/16
    s2 = (sigma / K) ** 2
    wi = fa([0, 0, 1])
    rng = np.random.default_rng(1)
    for _ in range(50):
        xy = rng.uniform(-.7, .7, 2)
        wo = fa([xy[0], xy[1], math.sqrt(1 - xy @ xy)])
        z2 = K * K * (xy @ xy)
        norm = 1 / (1 - math.exp(-K * K / 2 / sigma ** 2))
        ref = norm / (2 * math.pi * sigma ** 2) * K * K * math.exp(-z2 / 2 / sigma ** 2)
        got = lib.kat_gaussian(0, 0.0, sigma, K, p(wi), p(wo))
        assert abs(got - ref) <= 2e-5 * ref + 1e-12
    # normal incidence: the PSD integrates to one over the unit disk of outgoing projected directions
    n = 400
    g = (np.arange(n) + .5) / n * 2 - 1
    X, Y = np.meshgrid(g, g)
    m = X * X + Y * Y < 1
    vals = [lib.kat_gaussian(0, 0.0, sigma, K, p(wi), p(fa([x, y, math.sqrt(max(0, 1 - x * x - y * y))]))) for x, y in zip(X[m][::7], Y[m][::7])]
    assert abs(np.mean(vals) * math.pi - 1) < 0.02
    # roughness-parametrised variant: sigma^2 = 1/T of the fractal profile's roughness -> T mapping; specular fraction
    a = lib.kat_gaussian(2, 3e-4, 0.0, K, p(wi), p(wi))
    assert abs(a - math.exp(-(2 * K) ** 2 * (3e-4 / 9) ** 2)) < 1e-4


@pytest.mark.parametrize("theta_i", [0.0, 0.5, 1.1])
def test_gaussian_sampler_matches_pdf(lib, theta_i):
    sigma = 0.2 * K
    s2 = (sigma / K) ** 2
    wi = fa([math.sin(theta_i), 0, math.cos(theta_i)])
    n = 200000
    o = np.zeros((n, 5), np.float32)
    lib.kat_gaussian_sample(0.0, sigma, K, p(wi), 5, n, p(o))
    assert np.allclose(np.linalg.norm(o[:, :3], axis=1), 1, atol=1e-3) and (o[:, 2] >= 0).all() and (o[:, 3] > 0).all()
    # returned pdf == pdf(wo) evaluated afterwards
    for row in o[:200]:
        pd = lib.kat_gaussian(1, 0.0, sigma, K, p(wi), p(fa(row[:3])))
        assert abs(pd - row[3]) <= 5e-3 * row[3] + 1e-6
    # The sampler draws r from the Gaussian truncated at r <= 1 + sin(theta_i) and the azimuth uniformly on the arc that stays
    # inside the unit disk; its true density over the disk is pdf / (cos(theta_i) (1 - s)), s = exp(-(1+sin)^2 / 2 s2): the
    # reference's pdf carries the cos(theta_i) factor and omits the truncation mass (gaussian.hpp:48, kept verbatim).  Hence
    # E[psd/pdf] = Int_disk psd dA / (cos(theta_i) (1 - s)); the integral is evaluated on a grid from the closed-form PSD.
    l = math.sin(theta_i)
    sm = math.exp(-.5 * (1 + l) ** 2 / s2)
    n_g = 1200
    g = (np.arange(n_g) + .5) / n_g * 2 - 1
    X, Y = np.meshgrid(g, g)
    inside = X * X + Y * Y < 1
    norm = 1 / (1 - math.exp(-1 / (2 * s2)))
    psd_grid = norm / (2 * math.pi * s2) * np.exp(-((X + l) ** 2 + Y ** 2) / (2 * s2))     # per unit area of the projected disk
    integral = psd_grid[inside].sum() * (2 / n_g) ** 2
    est = np.mean(o[:, 4].astype(np.float64) / o[:, 3].astype(np.float64))
    assert abs(est / (integral / (math.cos(theta_i) * (1 - sm))) - 1) < 0.01, (est, integral)


def test_gaussian_profile_in_a_render(built):
    """'furnace_spm': the furnace room with rough-conductor occluders (Gaussian profile, both parametrisations, and fractal): finite,
    non-negative, and — conductors absorb — darker than, but close to, the all-diffuse furnace."""
    from wave_tracer_amd import Scene, develop
    from oracle_util import oracle_render
    imgs = {}
    for name in ("furnace", "furnace_spm"):
        sc = Scene(name, res=32)
        v, w, l, c = oracle_render(sc, 0, 8, 3)
        imgs[name] = develop(sc, v, w, l, 8).astype(np.float64)
        assert np.isfinite(imgs[name]).all() and (imgs[name] >= 0).all()
    r = imgs["furnace_spm"].mean() / imgs["furnace"].mean()
    assert 0.8 < r < 1.2, r
