"""GPU parity of the scale wrapper's non-constant factors (material_t::scale_spec / scale_tex, wt/bsdf.h: material_scale_factor), which were
added after the last GPU session of round 2: the same Philox streams on both sides, relative L1 of the developed image < 1e-2 like
tests/test_gpu_render.py::test_image_parity_small.  (Last in the alphabet on purpose: it runs after every other GPU test.)"""
import os

import numpy as np

import parity
import pytest

from oracle_util import oracle_render

HERE = os.path.dirname(os.path.abspath(__file__))

BOX = '''<scene version="0.1.0">
  <integrator type="plt_bdpt"><integer name="max_depth" value="5"/><boolean name="FSD" value="false"/></integrator>
  <sensor type="perspective"><quantity name="fov" value="60°"/>
    <transform name="to_world"><lookat origin="0m, 0m, .9m" target="0m, 0m, 0m" up="0, 1, 0"/></transform>
    <film type="array"><integer name="width" value="24"/><integer name="height" value="24"/>
      <response type="RGB"><string name="white_point" value="E"/></response></film></sensor>
  <shape type="cube"><quantity name="length" value="2m"/><bsdf type="twosided"><bsdf type="scale"><spectrum name="scale" rgb=".8, .4, .2"/>
    <bsdf type="diffuse"><spectrum name="reflectance" constant="1"/></bsdf></bsdf></bsdf></shape>
  <shape type="rectangle"><point name="p" x="-.25m" y=".95m" z="-.25m"/><point name="x" x="0m" y="0m" z=".5m"/><point name="y" x=".5m" y="0m" z="0m"/>
    <bsdf type="diffuse"><spectrum name="reflectance" constant="0"/></bsdf>
    <emitter type="area"><spectrum name="radiance" blackbody="6000K"><float name="scale" value="1e-6"/></spectrum></emitter></shape>
</scene>'''


def _parity(sc, spp, seed, label):
    from wave_tracer_amd import render, develop
    v, w, l = render(sc, spp, seed=seed)
    ov, ow, ol, oc = oracle_render(sc, 0, spp, seed)
    g, c = develop(sc, v, w, l, spp).astype(np.float64), develop(sc, ov, ow, ol, spp).astype(np.float64)
    assert np.isfinite(g).all() and c.sum() > 0
    assert np.allclose(w, ow, rtol=1e-5, atol=1e-6)
    rel = np.abs(g - c).sum() / np.abs(c).sum()
    parity.check(f"scale_wrappers/{label}", rel, 1e-2)
    gc = sc.counters()
    for key in ("segments", "vertices", "connections", "surface_interactions"):
        assert abs(gc[key] - oc[key]) <= 2e-3 * max(1, oc[key]), (key, gc[key], oc[key])
    return g


@pytest.mark.gpu
def test_textured_scale_factor_gpu_parity(built):
    from wave_tracer_amd import Scene
    g = _parity(Scene.from_xml(os.path.join(HERE, "data", "xml", "textured.xml"), defines={"variant": 5}), 8, 5, "textured_scale_factor")
    assert g.max() > 2 * np.median(g[g > 0])      # the checks are visible


@pytest.mark.gpu
def test_spectral_scale_factor_gpu_parity(built, tmp_path):
    from wave_tracer_amd import Scene
    f = tmp_path / "box.xml"
    f.write_text(BOX)
    g = _parity(Scene.from_xml(str(f), lut=(32, 32)), 8, 5, "spectral_scale_factor")
    assert g[..., 0].sum() > 1.5 * g[..., 2].sum()     # reddish walls


@pytest.mark.gpu
def test_rgb_bitmap_uplift_gpu_parity(built, tmp_path):
    """An RGB bitmap as the walls' reflectance: uplifted per lookup on the device (wt/scene.h: rgb_uplift, an out-of-line device function)."""
    from wave_tracer_amd import Scene
    from wave_tracer_amd.imageio import write_pfm
    write_pfm(str(tmp_path / "c.pfm"), np.array([[[.8, .4, .2], [.2, .4, .8]], [[.3, .7, .3], [.6, .6, .6]]], np.float32))
    walls = (f'<bsdf type="twosided"><bsdf type="diffuse"><texture name="reflectance" type="bitmap"><path value="{tmp_path / "c.pfm"}"/>'
             '<string name="filter_type" value="bilinear"/></texture></bsdf></bsdf>')
    start = BOX.index('<bsdf type="twosided">')
    end = BOX.index('</bsdf></bsdf></bsdf>') + len('</bsdf></bsdf></bsdf>')
    f = tmp_path / "box.xml"
    f.write_text(BOX[:start] + walls + BOX[end:])
    _parity(Scene.from_xml(str(f), lut=(32, 32)), 8, 5, "rgb_bitmap_uplift")
