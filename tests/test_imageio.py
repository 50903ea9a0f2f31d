"""Host-side tonemapping / image writers (include/wt/sensor/response/tonemap/tonemap.hpp, src/sensor/response/tonemap.cpp)."""
import numpy as np

from wave_tracer_amd.imageio import colourmap, read_exr, srgb_from_linear, tonemap, tonemap_db, write_developed, write_exr, write_pfm, write_ppm


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


def test_exr_round_trip_and_header_layout(tmp_path):
    """Uncompressed scanline OpenEXR: the writer's files read back bit for bit, the header is the documented layout (magic 20000630,
    version 2, channels in alphabetical order, one chunk per scanline) and carries the reference's string attributes."""
    import struct
    rng = np.random.default_rng(1)
    rgb = rng.uniform(0, 5, (7, 11, 3)).astype(np.float32)
    p = tmp_path / "a.exr"
    write_exr(str(p), rgb, {"renderer": "wave_tracer_amd", "scene": "cornell_box", "sensor": "camera", "samples": "4096"})
    img, names, attrs = read_exr(str(p))
    assert names == ["B", "G", "R"]                                            # stored alphabetically
    assert np.array_equal(img[..., 2], rgb[..., 0]) and np.array_equal(img[..., 1], rgb[..., 1]) and np.array_equal(img[..., 0], rgb[..., 2])
    assert attrs == {"renderer": "wave_tracer_amd", "scene": "cornell_box", "sensor": "camera", "samples": "4096"}
    raw = p.read_bytes()
    assert struct.unpack_from("<ii", raw, 0) == (20000630, 2)
    assert b"channels\0chlist\0" in raw and b"compression\0compression\0\x01\0\0\0\0" in raw and b"dataWindow\0box2i\0" in raw
    # file size = header + offset table + H x (8 + 4 W C)
    hdr_end = raw.index(b"screenWindowWidth")
    assert len(raw) > hdr_end + 8 * 7 + 7 * (8 + 4 * 11 * 3)
    mono = rng.uniform(0, 1, (5, 4)).astype(np.float32)
    write_exr(str(tmp_path / "m.exr"), mono)
    img, names, _ = read_exr(str(tmp_path / "m.exr"))
    assert names == ["Y"] and np.array_equal(img[..., 0], mono)


def test_developed_films_are_written_like_the_reference_names_them(tmp_path):
    """src/main.cpp:331-430: `<sensor>.exr` for intensity films, `<sensor>_I/_Q/_U/_V.exr` for polarimetric sensors; attributes
    renderer / scene / sensor / samples on every file."""
    import os
    rng = np.random.default_rng(2)
    film = rng.uniform(0, 1, (6, 8, 3)).astype(np.float32)
    (p,) = write_developed(str(tmp_path), "camera", "box", film, 64)
    assert os.path.basename(p) == "camera.exr"
    img, names, attrs = read_exr(p)
    assert attrs["samples"] == "64" and attrs["sensor"] == "camera" and attrs["scene"] == "box" and "renderer" in attrs
    stokes = rng.normal(size=(6, 8, 3, 4)).astype(np.float32)                   # [H][W][C][S]
    paths = write_developed(str(tmp_path), "pol", "room", stokes.reshape(6, 8, 12), 8, stokes=4)
    assert [os.path.basename(q) for q in paths] == ["pol_I.exr", "pol_Q.exr", "pol_U.exr", "pol_V.exr"]
    for s, q in enumerate(paths):
        img, names, attrs = read_exr(q)
        assert names == ["B", "G", "R"] and np.array_equal(img[..., ::-1], stokes[..., s])      # Q, U, V are signed: no clamping
