"""SURVEY.md §8c K10.  The balance-heuristic weights of plt_bdpt (plt_bdpt_detail.hpp:604-720) sum to one over the strategies of a
path exactly when every density a random walk STORES at sampling time equals the density the MIS code EVALUATES for the same
transition from the other side (both are then the same numbers p_i in w_s = p_s / sum_i p_i).  These tests pin that property for
every sampler on the hot path: BSDFs, the sensors, the emitters.  (The complementary statistical check — all strategies together
reproduce the closed-form white furnace — is in tests/test_oracle.py.)  CPU only."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle_util import load_oracle

F = C.c_float


@pytest.fixture(scope="module")
def lib(built):
    lib = load_oracle()
    lib.kat_material_sample_consistency.argtypes = [C.c_void_p, C.c_int, C.c_void_p, F, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p]
    lib.kat_sensor_sample_consistency.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, F, C.c_uint64, C.c_uint32, C.c_void_p]
    lib.kat_emitter_sample_consistency.argtypes = [C.c_void_p, C.c_int, F, C.c_uint64, C.c_uint32, C.c_void_p]
    return lib


def fa(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


K = 2 * math.pi / 5.5e-4


@pytest.mark.parametrize("scene,kw,mats", [("furnace_spm", {}, [0, 2, 3, 4]), ("cornell_box", {"mesh_detail": 0, "lut": (32, 32)}, [0, 1, 2]),
                                           # the dispatching wrappers: a two-bin composite (material 2; queried at its bin boundary), a
                                           # two-sided mask of opacity 0.6 over a diffuse BSDF (material 1): nested density x alpha
                                           ("furnace_wall_composite", {}, [2]), ("furnace_wall_mask", {}, [1])])
@pytest.mark.parametrize("transport", [0, 1])
def test_bsdf_sampled_density_equals_evaluated_density(lib, scene, kw, mats, transport):
    """Diffuse BSDFs: sampled density == evaluated density and weight == f/pdf, exactly.  Rough conductors (surface_spm without
    transmission): the reference's pdf() always applies the reflection/transmission split computed from Re(eta)
    (src/bsdf/surface_spm.cpp:190-196) while sample() applies it only when the IOR transmits (:108-111), so the evaluated density
    is the sampled one times (1 - T(Re eta, wi)) — a factor that does not depend on wo.  Kept verbatim; pinned here."""
    from wave_tracer_amd import Scene
    sc = Scene(scene, res=16, **kw)
    h = C.c_void_p(sc.host_desc())
    lib.kat_spectrum.restype = F
    lib.kat_spectrum.argtypes = [C.c_void_p, C.c_int, F, C.c_void_p]
    lib.kat_material_ior_spec.argtypes = [C.c_void_p, C.c_int]
    n = 400
    checked = 0
    for mat in mats:
        ior_spec = lib.kat_material_ior_spec(h, mat)
        eta = None
        if ior_spec >= 0:
            im = C.c_float()
            re = lib.kat_spectrum(h, ior_spec, F(K), C.byref(im))
            eta = complex(re, im.value)
        for th in (0.2, 0.9, 1.3):
            for up in (1.0, -1.0):
                wi = fa([math.sin(th) * .8, math.sin(th) * .6, up * math.cos(th)])
                o = np.zeros((n, 5), np.float32)
                lib.kat_material_sample_consistency(h, mat, p(wi), F(K), transport, 7, n, p(o))
                cont = (o[:, 0] > 0) & (o[:, 1] > 0)    # continuous lobes (discrete ones are stored negative: delta, skipped by MIS)
                if not cont.any():
                    continue
                ratio = o[cont, 0].astype(np.float64) / o[cont, 1].astype(np.float64)
                assert ratio.std() <= 3e-3 * ratio.mean(), (scene, mat, th, up)          # independent of the sampled direction
                conductor = eta is not None and (eta.imag ** 2) / abs(eta) ** 2 > 1e-2
                if conductor:
                    out = np.zeros(10, np.float32)
                    lib.kat_fresnel(F((1 / eta).real), p(fa([wi[0], wi[1], abs(wi[2])])), p(out))   # eta_12 = eta_exterior / eta_interior
                    expected = 1.0 / (1.0 - (out[4] + out[5]) / 2)
                    assert abs(ratio.mean() / expected - 1) < 5e-3, (scene, mat, th, ratio.mean(), expected)
                else:
                    assert abs(ratio.mean() - 1) < 3e-3, (scene, mat, th, up, ratio.mean())
                checked += int(cont.sum())
                if eta is None:
                    assert np.allclose(o[cont, 2], o[cont, 3], rtol=1e-4, atol=1e-9)   # Lambertian: weight * pdf == f
    assert checked > (1000 if len(mats) > 1 else 300)


def test_sensor_and_emitter_sampled_densities_equal_evaluated_densities(lib):
    from wave_tracer_amd import Scene
    n = 300
    for scene, kw in (("furnace", {"lut": (32, 32)}), ("double_slits", {"lut": (32, 32)}), ("cornell_box", {"mesh_detail": 0, "lut": (32, 32)}),
                      ("etoile", {"mesh_detail": 0}), ("sunlit", {})):
        sc = Scene(scene, res=32, **kw)
        h = C.c_void_p(sc.host_desc())
        k = K if scene not in ("double_slits", "etoile") else (2 * math.pi / .05 if scene == "double_slits" else 2 * math.pi / 29.9792458)
        o = np.zeros((n, 4), np.float32)
        lib.kat_sensor_sample_consistency(h, sc.width // 3, sc.height // 2, F(k), 9, n, p(o))
        assert np.allclose(o[:, 0], o[:, 1], rtol=2e-3, atol=1e-12), scene        # direction density
        assert np.allclose(o[:, 2], o[:, 3], rtol=1e-5), scene                    # position density (or discrete mass)
        for ei in range(sc.info.n_emitters):
            e = np.zeros((n, 4), np.float32)
            lib.kat_emitter_sample_consistency(h, ei, F(k), 11, n, p(e))
            if scene == "sunlit":
                # infinite emitter: direction is a delta (mass 1), its positions have no density (vertex.hpp:557-558) — the
                # MIS code uses pdf_target_position instead (pinned in tests/test_emitters.py)
                assert np.allclose(e[:, 0], e[:, 1]) and (e[:, 3] == 0).all()
                continue
            assert np.allclose(e[:, 0], e[:, 1], rtol=2e-3, atol=1e-12), (scene, ei)
            assert np.allclose(e[:, 2], e[:, 3], rtol=1e-5), (scene, ei)


def test_fraunhofer_fsd_sampled_density_vs_evaluated_density(lib):
    """Double-slit aperture under the spot's beam: the sampled direction's density is f(xi)/I (fsd_sampler.cpp:72-110,
    free_space_diffraction.hpp:83-99) and pdf() evaluates the same function — but clamps: densities >= 100 sr^-1 evaluate to ZERO
    (free_space_diffraction.hpp:133).  With a millimetre-sized beam at 50 um most of the diffracted lobe is that peaked, so the
    reverse densities the MIS code sees for such vertices vanish.  Reference behaviour, kept verbatim; pinned here."""
    from wave_tracer_amd import Scene
    lib.kat_fsd_sample_consistency.restype = C.c_uint32
    lib.kat_fsd_sample_consistency.argtypes = [C.c_void_p, C.c_void_p, F, F, F, F, C.c_uint64, C.c_uint32, C.c_void_p]
    sc = Scene("double_slits", res=64, lut=(128, 128))
    k = 2 * math.pi / .05
    cone = fa([0, 0, -0.5, 0, 0, 1])          # from the spot's position towards the slits (screen at z = -15 mm)
    n = 4000
    o = np.zeros((n, 6), np.float32)
    n_seg = lib.kat_fsd_sample_consistency(C.c_void_p(sc.host_desc()), p(cone), F(0.002), F(1e-4), F(0.485), F(k), 3, n, p(o))
    assert n_seg >= 8
    ok = o[:, 3] > 0
    assert ok.mean() > 0.9 and (o[ok, 5] == 1).all()                       # rejection sampling: weight 1
    assert np.allclose(np.linalg.norm(o[ok, 0:3], axis=1), 1, atol=1e-4) and (o[ok, 2] > 0).all()
    low = ok & (o[:, 3] < 99.0)
    high = ok & (o[:, 3] > 101.0)
    assert low.sum() > 100 and high.sum() > 100
    assert np.allclose(o[low, 3], o[low, 4], rtol=2e-3)
    assert (o[high, 4] == 0).all()


def test_dielectric_and_conductor_energy_balance(lib):
    """E[sample weight] over the lobe choice: a lossless dielectric interface conserves energy in forward transport (R + T = 1,
    dielectric.cpp:26-72; backward transport carries the eta^2 radiance scaling on transmission), a smooth conductor reflects
    less than one, a Lambertian reflects its albedo."""
    from wave_tracer_amd import Scene
    sc = Scene("cornell_box", res=16, mesh_detail=0, lut=(32, 32))
    h = C.c_void_p(sc.host_desc())
    n = 20000
    for th in (0.1, 0.7, 1.2):
        for up in (1.0, -1.0):
            wi = fa([math.sin(th), 0, up * math.cos(th)])
            o = np.zeros((n, 5), np.float32)
            lib.kat_material_sample_consistency(h, 4, p(wi), F(K), 0, 5, n, p(o))       # SF5 dielectric, forward transport
            w = np.where(o[:, 0] != 0, o[:, 2] / np.where(o[:, 0] != 0, o[:, 0], 1), 0)
            assert abs(w.mean() - 1) < 0.02, (th, up, w.mean())
            if up > 0:
                lib.kat_material_sample_consistency(h, 3, p(wi), F(K), 0, 5, n, p(o))   # gold, Dirac profile, lit from outside
                w = np.where(o[:, 0] != 0, o[:, 2] / np.where(o[:, 0] != 0, o[:, 0], 1), 0)
                assert 0.3 < w.mean() < 1.0
    wi = fa([.3, .2, math.sqrt(1 - .13)])
    o = np.zeros((n, 5), np.float32)
    lib.kat_material_sample_consistency(h, 1, p(wi), F(K), 0, 5, n, p(o))               # right wall: diffuse .6
    assert abs((o[:, 2] / o[:, 0]).mean() - .6) < 1e-3
