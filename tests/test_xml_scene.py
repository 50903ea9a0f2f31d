"""The minimal reader of the reference's XML scene format (wtgpu_scene_create_from_xml, csrc/host/xml_scene.cpp).
(1) The reference's own scenes/diffraction_simple/double_slits.xml (+ bits/geometry.xml), read where /root/reference exists, must
    bake to the SAME flattened scene, byte for byte, as the hand-written builder host/scenes.cpp:build_double_slits that every
    double-slit test of this repository uses — for the pattern sensor and for -Doptical_overview=true.
(2) A scene file written for this repository (tests/data/xml/) exercises the reader's features without the reference: defines and
    overrides, expressions with units, includes, disabled elements, composite materials, both sensor kinds, error reporting."""
import os

import numpy as np
import pytest

from oracle_util import oracle_render

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/scenes/diffraction_simple"
OWN = os.path.join(HERE, "data", "xml", "single_slit.xml")
needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "double_slits.xml")), reason="the reference checkout is not present on this machine")


@needs_reference
@pytest.mark.parametrize("res,lut", [(96, (64, 64)), (360, (0, 0)), (1440, (32, 32))])
def test_reference_double_slits_xml_bakes_to_the_builders_scene(built, res, lut):
    from wave_tracer_amd import Scene
    a = Scene.from_xml(os.path.join(REF, "double_slits.xml"), res=res, lut=lut)
    b = Scene("double_slits", res=res, lut=lut)
    assert (a.width, a.height) == (res, res // 4) and a.info.n_tris == 10 and a.info.n_emitters == 1 and a.info.max_depth == 16
    assert a.first_difference(b) == ""


@needs_reference
def test_reference_double_slits_overview_define(built):
    from wave_tracer_amd import Scene
    a = Scene.from_xml(os.path.join(REF, "double_slits.xml"), defines={"optical_overview": "true", "res": 64})
    b = Scene("double_slits_overview", res=64)
    assert (a.width, a.height, a.channels) == (64, 64, 3) and a.info.n_emitters == 2 and a.info.sensor_type == 0
    assert a.first_difference(b) == ""
    # screen=false removes the three screen rectangles (bits/geometry.xml: enabled = $screen)
    c = Scene.from_xml(os.path.join(REF, "double_slits.xml"), defines={"screen": "false"}, res=64, lut=(32, 32))
    assert c.info.n_tris == 4


@needs_reference
def test_reference_reflectors_variant_loads_as_forward_plt_path(built):
    from wave_tracer_amd import Scene
    c = Scene.from_xml(os.path.join(REF, "double_slits_and_reflectors.xml"), defines={"reflectors": "true", "res": 128})
    assert (c.width, c.height) == (128, 32) and c.info.integrator == 1 and c.info.n_tris == 14 and c.info.n_materials == 4
    v, w, l, ctr = oracle_render(c, 0, 2, 5)
    assert np.isfinite(v).all() and np.isfinite(l).all() and ctr["segments"] > 0


def test_own_scene_defaults(built):
    from wave_tracer_amd import Scene
    s = Scene.from_xml(OWN, lut=(32, 32))
    assert (s.width, s.height, s.channels) == (64, 16, 1) and s.info.sensor_type == 1 and s.info.integrator == 0
    assert s.info.max_depth == 8
    assert s.info.n_tris == 6 and s.info.n_shapes == 3 and s.info.n_materials == 2
    assert s.info.n_emitters == 1          # the blackbody emitter has no line overlap with the 50 um sensor
    v, w, l, c = oracle_render(s, 0, 4, 3)
    assert c["fsd_interactions"] > 0 and (v.sum() + l.sum()) > 0     # light diffracts through the slit onto the wall sensor


def test_own_scene_define_overrides_and_disabled_elements(built):
    from wave_tracer_amd import Scene
    s = Scene.from_xml(OWN, defines={"res": 128, "backdrop": "false", "half": "10"}, lut=(32, 32))
    assert (s.width, s.height) == (128, 32) and s.info.n_tris == 4
    t = Scene.from_xml(OWN, res=32, lut=(32, 32))                    # params.res stands in for -Dres
    assert (t.width, t.height) == (32, 8)
    cam = Scene.from_xml(OWN, defines={"camera": "true", "res": 48})
    assert (cam.width, cam.height, cam.channels) == (48, 48, 3) and cam.info.sensor_type == 0
    assert cam.info.n_emitters == 1        # now the far-infrared line is the one without overlap, the blackbody stays
    # the composite paint resolves to its optical bin for the camera and to its radio bin for the wall sensor: different scenes
    assert cam.first_difference(Scene.from_xml(OWN, defines={"res": 48})) != ""
    # max_depth override through the parameter struct (the CLI's integrator overrides)
    assert Scene.from_xml(OWN, max_depth=5, lut=(32, 32)).info.max_depth == 5


def test_equivalent_spellings_bake_identically(built):
    """Units and expressions are evaluated, not pattern-matched: the same quantities spelled differently give the same scene."""
    from wave_tracer_amd import Scene
    a = Scene.from_xml(OWN, defines={"wall_z": "40", "slit": "0.4"}, lut=(32, 32))
    b = Scene.from_xml(OWN, defines={"wall_z": "(2*20)", "slit": "(4/10)"}, lut=(32, 32))
    assert a.first_difference(b) == ""


def test_reader_reports_errors(built, tmp_path):
    from wave_tracer_amd import Scene
    from wave_tracer_amd.api import WtgpuError
    with pytest.raises(WtgpuError, match="cannot open"):
        Scene.from_xml(str(tmp_path / "missing.xml"))
    undef = tmp_path / "undef.xml"
    undef.write_text(open(OWN).read().replace('value="$spp"', 'value="$nope"'))
    with pytest.raises(WtgpuError, match="undefined \\$nope"):
        Scene.from_xml(str(undef))
    with pytest.raises(WtgpuError, match="unknown unit"):
        Scene.from_xml(OWN, defines={"wl": "50 parsec"})
    with pytest.raises(WtgpuError, match="more than one enabled sensor|no enabled sensor"):
        bad = tmp_path / "two.xml"
        bad.write_text(open(OWN).read().replace('value="$camera"', 'value="true"').replace('value="!$camera"', 'value="true"'))
        os.makedirs(tmp_path / "parts", exist_ok=True)
        (tmp_path / "parts" / "slit_geometry.xml").write_text(open(os.path.join(HERE, "data", "xml", "parts", "slit_geometry.xml")).read())
        Scene.from_xml(str(bad))
    broken = tmp_path / "broken.xml"
    broken.write_text("<scene>\n  <integrator type='plt_bdpt'>\n</scene>\n")
    with pytest.raises(WtgpuError, match="broken.xml:3"):
        Scene.from_xml(str(broken))
    sphere = tmp_path / "sphere.xml"
    sphere.write_text(open(OWN).read().replace('<include path="parts/slit_geometry.xml"/>', '<shape type="sphere"><ref id="metal"/></shape>'))
    with pytest.raises(WtgpuError, match="not supported by the minimal reader"):
        Scene.from_xml(str(sphere))
    loop = tmp_path / "loop.xml"
    loop.write_text('<scene><include path="loop_part.xml"/></scene>')
    (tmp_path / "loop_part.xml").write_text('<include path="loop_part.xml"/>')
    with pytest.raises(WtgpuError, match="nested deeper"):
        Scene.from_xml(str(loop))
    nodir = tmp_path / "nodir.xml"
    nodir.write_text("<scene><integrator type='plt_path'/></scene>")
    with pytest.raises(WtgpuError, match="direction"):
        Scene.from_xml(str(nodir))
