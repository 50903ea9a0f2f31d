"""Host-side tonemapping / image writers (include/wt/sensor/response/tonemap/tonemap.hpp, src/sensor/response/tonemap.cpp)."""
import numpy as np

from wave_tracer_amd.imageio import colourmap, srgb_from_linear, tonemap, tonemap_db, write_pfm, write_ppm


def test_db_operator_follows_the_reference_formula():
    x = np.array([0.0, 1e-9, 1e-6, 1e-3, 1.0, 10.0])
    got = tonemap_db(x, -60.0, 0.0)
    assert got[0] == 0.0                                        # zero stays zero (tonemap.cpp:63-64)
    assert np.allclose(got[1:], np.clip((10 * np.log10(x[1:]) + 60) / 60, 0, 1))
    assert got[1] == 0.0 and got[4] == 1.0 and got[5] == 1.0   # clamped


def test_srgb_transfer_and_modes():
    assert np.isclose(srgb_from_linear(np.array(.0031308)), 12.92 * .0031308)
    assert np.isclose(srgb_from_linear(np.array(.5)), 1.055 * .5 ** (1 / 2.4) - .055)
    assert np.isclose(srgb_from_linear(np.array(2.0)), 1.0) and srgb_from_linear(np.array(-1.0)) == 0.0
    rgb = np.random.default_rng(0).uniform(0, 1, (4, 5, 3))
    out = tonemap(rgb, op="sRGB")                                # polychromatic + select: per channel
    assert np.allclose(out, srgb_from_linear(rgb))
    mono = rgb[..., :1]
    cm = tonemap(mono, op="linear", cmap="grey")                 # monochrome + select: colour map
    assert cm.shape == (4, 5, 3) and np.allclose(cm[..., 0], mono[..., 0])
    db = tonemap(rgb, op="dB", db_range=(-30, 0), cmap="grey")    # dB always maps the luminance through the colour map
    lum = np.maximum(0, rgb @ np.array([.2126, .7152, .0722]))
    assert np.allclose(db[..., 1], tonemap_db(lum, -30, 0))
    t = colourmap(np.linspace(0, 1, 64))
    assert t.shape == (64, 3) and (t >= 0).all() and (t <= 1).all() and t[8, 2] > t[8, 0] and t[-4, 0] > t[-4, 2]   # blue -> red


def test_image_writers_round_trip(tmp_path):
    img = np.random.default_rng(1).uniform(0, 1, (6, 7, 3))
    write_ppm(tmp_path / "a.ppm", img)
    raw = (tmp_path / "a.ppm").read_bytes()
    assert raw.startswith(b"P6\n7 6\n255\n") and len(raw) == len(b"P6\n7 6\n255\n") + 6 * 7 * 3
    write_pfm(tmp_path / "a.pfm", img[..., 0])
    raw = (tmp_path / "a.pfm").read_bytes()
    hdr = b"Pf\n7 6\n-1.0\n"
    assert raw.startswith(hdr)
    back = np.frombuffer(raw[len(hdr):], np.float32).reshape(6, 7)[::-1]
    assert np.allclose(back, img[..., 0].astype(np.float32))
