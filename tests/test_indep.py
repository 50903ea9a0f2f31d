"""The independent composition checker (oracle/indep/: vertices, area-measure densities, Russian roulette, the (s,t) strategies, MIS and
beam integration restated a second time from the reference, in double precision, sharing no header with wave_tracer_amd/csrc/wt/) against
liboracle.so (the shared-header composition the GPU tests are checked against).  Both consume the same Philox streams in the reference's
order, so they must agree SAMPLE FOR SAMPLE: event counters exactly, films to float rounding.  A bug in either composition — a wrong
density conversion, a missing cosine, a MIS special case — shows up here; a bug in a primitive does not (those are pinned by the KATs)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle_util import ROOT, oracle_render

_lib = None


def indep():
    global _lib
    if _lib is None:
        p = os.path.join(ROOT, "oracle", "_build", "libindep.so")
        if not os.path.exists(p):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _lib = C.CDLL(p)
        _lib.indep_render.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def indep_render(sc, b, e, seed):
    H, W, Cn = sc.height, sc.width, sc.channels
    v, w, l = np.zeros((H, W, Cn)), np.zeros((H, W)), np.zeros((H, W, Cn))
    ctr = np.zeros(8, np.uint64)
    assert indep().indep_render(sc.host_desc(), b, e, seed, v.ctypes.data, w.ctypes.data, l.ctypes.data, ctr.ctypes.data) == 0
    return v, w, l, dict(zip(["segments", "vertices", "connections", "surface", "fsd_interactions", "null_interactions", "light_splats", "shadow_rays"],
                             [int(x) for x in ctr]))


CASES = [
    # scene, res, spp, kwargs — every strategy class: s=0 (emitter hit), t=0 (virtual sensor), s=1 / t=1 (direct sampling), interior
    ("furnace", 16, 4, {}),
    ("white_furnace", 12, 6, {}),
    ("furnace", 16, 4, {"fsd": 1, "lut": (64, 64)}),
    ("furnace_spm", 16, 4, {}),
    ("furnace", 12, 4, {"mis": 0}),                      # the non-MIS weight 1 / ((s+t+1) k_density), plt_bdpt.cpp:126
    ("furnace", 12, 4, {"rr": 0}),
    ("double_slits", 48, 4, {"lut": (64, 64)}),          # virtual_plane sensor: t = 0 strategy, Fraunhofer vertices, spot emitter
    ("sunlit", 16, 4, {}),                               # directional (infinite) emitter
    ("cornell_box", 16, 2, {"mesh_detail": 0, "lut": (64, 64), "crop_of": 1440}),
    ("lens_b", 16, 4, {}),                               # dielectric (delta) vertices
    ("bidir_room", 20, 2, {"mesh_detail": 0, "lut": (64, 64), "polarimetric": 1}),   # Stokes film: polarised beam integration
    # subpaths far beyond the 18 vertices rounds 1-2 were compiled for (the reference's max_depth is a plain integer, plt_bdpt.cpp:111,169;
    # kitchen / sponza / veach_mis / munich ask for 24..128): 82 vertices per sample, strategies up to (s,t) = (41,41)
    ("white_furnace", 8, 4, {"max_depth": 40, "rr": 0}),
    ("furnace", 12, 4, {"max_depth": 32, "rr": 0, "fsd": 1, "lut": (64, 64)}),
]


@pytest.mark.parametrize("name,res,spp,kw", CASES)
def test_independent_composition_matches_the_checker(built, name, res, spp, kw):
    from wave_tracer_amd import Scene
    sc = Scene(name, res=res, **kw)
    ov, ow, ol, oc = oracle_render(sc, 0, spp, 77, threads=1)
    iv, iw, il, ic = indep_render(sc, 0, spp, 77)
    for k in ("segments", "vertices", "connections", "fsd_interactions", "null_interactions", "light_splats", "shadow_rays"):
        assert ic[k] == oc[k], (k, ic[k], oc[k])
    assert np.allclose(iw, ow, rtol=1e-6, atol=1e-12)
    # Pixel by pixel: float rounding, EXCEPT single samples through a Fraunhofer vertex whose aperture holds an edge that is exactly
    # axis-aligned in the beam frame: alpha_1 / alpha_2 are defined as 0 at zeta.x == 0 (fsd.hpp:66-75) although their limit there is
    # not, so psi_0^2 (the 8-point average of free_space_diffraction.cpp:106-118) jumps by 30x with the last bit of the beam frame —
    # an instability of the formula (found with this test: the two compositions differ in FMA contraction only), also behind the
    # rare discrete GPU-vs-CPU divergences.  Such samples change only MIS weights: <= 3 % of the pixels, image < 2 %.
    tot_a = iv.sum(axis=2) + il.sum(axis=2)
    tot_b = ov.sum(axis=2) + ol.sum(axis=2)
    assert tot_b.sum() > 0
    same = np.abs(tot_a - tot_b) <= 1e-4 * np.abs(tot_b) + 1e-12 * tot_b.max()
    print(f"{name}: {same.mean():.4f} of the pixels agree to 1e-4, image rel. L1 {np.abs(tot_a - tot_b).sum() / tot_b.sum():.2e}")
    fsd_scene = oc["fsd_interactions"] > 0
    assert same.mean() >= (0.97 if fsd_scene else 1.0), same.mean()
    assert np.abs(tot_a - tot_b).sum() <= (2e-2 if fsd_scene else 1e-5) * tot_b.sum()


XML_CASES = [("textured.xml", {"variant": 1}, 4), ("textured.xml", {"variant": 3}, 4), ("textured.xml", {"variant": 5}, 4), ("objects.xml", {}, 2),
             ("single_slit.xml", {}, 4),
             # an area emitter with a bitmap radiance: the densities of its positions come from per-triangle tables AT the point's surface
             ("textured_emitter.xml", {"res": 16}, 6), ("textured_emitter.xml", {"res": 16, "filter": "bicubic", "mscale": 2}, 4)]


@pytest.mark.parametrize("file,defines,spp", XML_CASES)
def test_independent_composition_on_scene_files(built, file, defines, spp):
    """The same on scenes read by the XML reader (tests/data/xml/): textures, the mask wrapper, a textured scale factor, the procedural
    shapes with dielectric / conductor materials, a slit in front of a virtual-plane sensor."""
    from wave_tracer_amd import Scene
    sc = Scene.from_xml(os.path.join(ROOT, "tests", "data", "xml", file), defines=defines, lut=(32, 32))
    if sc.info.integrator != 0:
        pytest.skip("oracle/indep restates plt_bdpt only")
    ov, ow, ol, oc = oracle_render(sc, 0, spp, 41, threads=1)
    iv, iw, il, ic = indep_render(sc, 0, spp, 41)
    for k in ("segments", "vertices", "connections", "fsd_interactions", "null_interactions", "light_splats", "shadow_rays"):
        assert ic[k] == oc[k], (k, ic[k], oc[k])
    tot_a, tot_b = iv.sum(axis=2) + il.sum(axis=2), ov.sum(axis=2) + ol.sum(axis=2)
    assert tot_b.sum() > 0
    fsd_scene = oc["fsd_interactions"] > 0
    assert np.abs(tot_a - tot_b).sum() <= (2e-2 if fsd_scene else 1e-5) * tot_b.sum()


def indep_render_path(sc, b, e, seed):
    lib = indep()
    lib.indep_render_path.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    H, W, Cn = sc.height, sc.width, sc.channels
    v, w, l = np.zeros((H, W, Cn)), np.zeros((H, W)), np.zeros((H, W, Cn))
    ctr = np.zeros(8, np.uint64)
    assert lib.indep_render_path(sc.host_desc(), b, e, seed, v.ctypes.data, w.ctypes.data, l.ctypes.data, ctr.ctypes.data) == 0
    return v, w, l, dict(zip(["segments", "-", "connections", "surface_interactions", "fsd_interactions", "null_interactions", "light_splats", "edge_queries"],
                             [int(x) for x in ctr]))


def test_independent_plt_path_on_a_textured_emitter(built):
    """Backward plt_path on tests/data/xml/textured_emitter.xml: next-event estimation against, and emission weighted by, the table densities of
    an area emitter with a bitmap radiance (read at the hit surface) — second composition against the checker, sample for sample."""
    from wave_tracer_amd import Scene, develop
    sc = Scene.from_xml(os.path.join(ROOT, "tests", "data", "xml", "textured_emitter.xml"), defines={"res": 16, "integrator": "plt_path", "direction": "backward"})
    spp = 8
    ov, ow, ol, oc = oracle_render(sc, 0, spp, 9, threads=1)
    iv, iw, il, ic = indep_render_path(sc, 0, spp, 9)
    for k in ("segments", "connections", "surface_interactions", "null_interactions", "light_splats"):
        assert ic[k] == oc[k], (k, ic[k], oc[k])
    a, b = develop(sc, ov, ow, ol, spp).astype(np.float64), develop(sc, iv, iw, il, spp).astype(np.float64)
    assert a.sum() > 0 and oc["connections"] > 100
    assert np.abs(a - b).max() <= 1e-4 * a.max() and np.abs(a - b).sum() <= 1e-5 * a.sum()


PATH_CASES = [
    # forward plt_path from a point transmitter over the etoile stand-in: UTD apertures, do_fsd (Fermat points, two shadow rays per wedge,
    # coherent sum), nee_forward, sensing on the virtual coverage sensor
    ("etoile", 32, 4, {"mesh_detail": 0}),
    ("etoile", 48, 4, {"mesh_detail": 1}),
    # backward plt_path: nee_backward with the power heuristic, emission, Russian roulette
    ("white_furnace_path", 12, 8, {}),
    ("furnace_path", 12, 8, {}),
    ("cornell_box_path", 16, 4, {"mesh_detail": 0}),
    ("sunlit_path", 12, 8, {}),          # directional emitter under backward transport
]


@pytest.mark.parametrize("name,res,spp,kw", PATH_CASES)
def test_independent_plt_path_matches_the_checker(built, name, res, spp, kw):
    """plt_path restated a second time (oracle/indep/indep.cpp: path_random_walk, from plt_path_detail.hpp:33-828 — recursion, optionals and
    std::vector like the reference, double-precision bookkeeping, single-purpose primitives) against wt/path.h's explicit walk state through
    liboracle.so: every event counter identical, every pixel to 1e-4 of the image scale."""
    from wave_tracer_amd import Scene, develop
    sc = Scene(name, res=res, **kw)
    ov, ow, ol, oc = oracle_render(sc, 0, spp, 9, threads=1)
    iv, iw, il, ic = indep_render_path(sc, 0, spp, 9)
    for k in ("segments", "connections", "surface_interactions", "fsd_interactions", "null_interactions", "light_splats"):
        assert ic[k] == oc[k], (k, ic[k], oc[k])
    a = develop(sc, ov, ow, ol, spp).astype(np.float64)
    b = develop(sc, iv, iw, il, spp).astype(np.float64)
    assert a.sum() > 0 and np.allclose(iw, ow, rtol=1e-6, atol=1e-12)
    if name == "cornell_box_path":
        # UTD at OPTICAL wavelengths: the coherent sum over wedges carries phases k (r_o + r_i) ~ 1e7 rad, which the checker forms from f32
        # lengths like the reference and this restatement from doubles; single samples differ by ~1e-3 of their own value (with fsd = 0 the two
        # agree to 1e-7).  Which pixels such a sample lands in depends on the BVH (knife-edge samples also flip with the tree's tie order):
        # all but 1 % of the pixels to 1e-4, the rest to 5e-3, image 5e-4.
        assert (np.abs(a - b) > 1e-4 * a.max()).mean() < 1e-2
        assert np.abs(a - b).max() <= 5e-3 * a.max(), np.abs(a - b).max() / a.max()
        assert np.abs(a - b).sum() <= 5e-4 * a.sum()
        return
    assert np.abs(a - b).max() <= 1e-4 * a.max(), np.abs(a - b).max() / a.max()
    assert np.abs(a - b).sum() <= 1e-5 * a.sum()
