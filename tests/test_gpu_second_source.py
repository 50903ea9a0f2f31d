"""The HIP path against the SECOND composition on SECOND-SOURCE primitives (oracle/indep/indep.cpp + prims2.cpp -> libindep2.so: the integrators
restated a second time in f64 without the wt/ headers; cone x triangle as a convex programme, no box culls, Fresnel coefficients from the angle /
permittivity forms, Mueller matrices by the Kronecker construction, Fraunhofer segment amplitudes and UTD wedge coefficients from quadratures).

Every other GPU parity test compares libwtgpu.so with liboracle.so, which is compiled from the SAME wt/*.h the kernels include: a formula restated
wrongly there is invisible to them.  Here the chain is closed on the device: same scene, same counter-based random numbers, nothing else shared
but the scene description.  The arithmetic differs (f64 closest points, other formulas), so single samples take another discrete branch now and
then; tolerances are at most 10 x what was measured on the MI355X (the measured value is in each assertion's message)."""
import ctypes as C

import numpy as np
import pytest

import parity
from test_second_source import _indep2

pytestmark = pytest.mark.gpu

# name, res, spp, scene keywords, (sanity bound of the image rel. L1 — the tolerance proper is 10 x the committed measurement, tests/parity.py —,
# minimum share of pixels equal to 1e-3 [measured on the MI355X in brackets], counter tolerance)
CASES = [
    ("furnace", 16, 8, {"fsd": 1, "lut": (64, 64)}, (2e-2, 0.95, 5e-3)),        # [8.9e-4, 0.973: rejection decisions at the threshold flip single samples]
    ("furnace_spm", 16, 8, {}, (1e-4, 0.999, 1e-3)),                            # [4.7e-7, 1.0]
    ("lens_b", 16, 8, {}, (1e-4, 0.999, 1e-3)),                                 # [1.7e-8, 1.0]
    ("double_slits", 48, 8, {"lut": (64, 64)}, (1e-3, 0.97, 2e-3)),             # [1.1e-5, 0.984]
    ("bidir_room", 20, 4, {"mesh_detail": 0, "lut": (64, 64), "polarimetric": 1}, (1e-3, 0.999, 2e-3)),   # [1.1e-6, 1.0]
    ("etoile", 32, 8, {"mesh_detail": 0}, (5e-3, 0.98, 5e-3)),                  # [3.9e-4, 0.995: the reference's four-term asymptote of the UTD transition function]
]


@pytest.mark.parametrize("name,res,spp,kw,tol", CASES)
def test_gpu_film_equals_the_second_composition_on_second_source_primitives(built, name, res, spp, kw, tol):
    from wave_tracer_amd import Scene
    _compare(Scene(name, res=res, **kw), name, spp, tol)


@pytest.mark.parametrize("defs", [{"integrator": "plt_bdpt"}, {"integrator": "plt_path", "direction": "backward"}], ids=["bdpt", "backward"])
def test_gpu_textured_emitter_equals_the_second_composition(built, defs):
    """The area emitter with a bitmap radiance (tests/data/xml/textured_emitter.xml): positions from per-triangle texel tables, their densities read
    back at the hit surface by the MIS weights of both integrators — HIP film against the second composition."""
    import os
    from wave_tracer_amd import Scene
    xml = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "xml", "textured_emitter.xml")
    _compare(Scene.from_xml(xml, res=24, defines=defs), "textured_emitter-" + defs["integrator"], 8, (1e-3, 0.999, 2e-3))


def _compare(sc, name, spp, tol):
    from wave_tracer_amd import render
    H, W, Cn = sc.height, sc.width, sc.channels
    gv, gw, gl = render(sc, spp, seed=77, device=0)
    gc = sc.counters()
    v, w, l = np.zeros((H, W, Cn)), np.zeros((H, W)), np.zeros((H, W, Cn))
    ctr = np.zeros(8, np.uint64)
    lib = _indep2()
    entry = lib.indep_render if int(sc.info.integrator) == 0 else lib.indep_render_path
    calls = np.zeros(6, np.uint64)
    lib.ss_calls(calls.ctypes.data_as(C.c_void_p), 1)
    assert entry(sc.host_desc(), 0, spp, 77, v.ctypes.data, w.ctypes.data, l.ctypes.data, ctr.ctypes.data) == 0
    lib.ss_calls(calls.ctypes.data_as(C.c_void_p), 1)
    assert int(calls.sum()) > 0   # (the render did run on second-source primitives)
    ic = dict(zip(["segments", "vertices", "connections", "surface", "fsd_interactions", "null_interactions", "light_splats", "shadow_rays"], [int(x) for x in ctr]))
    rel_tol, same_min, ctr_tol = tol
    for k in ("segments", "connections") if int(sc.info.integrator) else ("segments", "vertices", "connections"):
        assert abs(gc[k] - ic[k]) <= ctr_tol * max(100, ic[k]) + 2, (k, gc[k], ic[k])
    assert np.allclose(gw, w, rtol=1e-5, atol=1e-9)
    # every plane of the film (Stokes components included), value + light
    a = gv.reshape(H, W, -1) + gl.reshape(H, W, -1)
    b = v.reshape(H, W, -1) + l.reshape(H, W, -1)
    assert np.abs(b).sum() > 0
    rel = np.abs(a - b).sum() / np.abs(b).sum()
    same = (np.abs(a - b).sum(axis=2) <= 1e-3 * np.abs(b).sum(axis=2) + 1e-9 * np.abs(b).max()).mean()
    print(f"{name}: GPU vs second composition on second-source primitives: rel. L1 {rel:.3e}, {same:.4f} of the pixels equal to 1e-3; "
          f"segments {gc['segments']}/{ic['segments']}, connections {gc['connections']}/{ic['connections']}")
    parity.check(f"second_source/{name}", rel, rel_tol)
    assert same >= same_min, f"pixels equal to 1e-3: {same:.4f} (minimum {same_min})"
