"""Textures on the hot path (SURVEY.md §8 row a10 / §2 `texture`): constant, checkerboard (texture/checkerboard.hpp), bitmap (nearest /
bilinear filtering, wrap modes: bitmap/texture2d.hpp, texture2d_storage.hpp) with the transform and scale wrappers folded in, feeding a
diffuse reflectance (src/bsdf/diffuse.cpp:25-31), the mask wrapper's opacity (src/bsdf/mask.cpp:24-92) and the normalmap wrapper's
shading frame (bsdf/normalmap.hpp:48-62).  Evaluated per interaction at the surface's uv on the device — the LFS bitmap assets of the
shipped scenes are absent, so the bitmaps here are generated.  Scene: a sunlit ground plane (uv in [0,1]^2, 4 m x 4 m) from above."""
import numpy as np
import pytest

from oracle_util import oracle_render

RES = 32


def _render(name, spp=4, seed=3, **kw):
    from wave_tracer_amd import Scene, develop
    sc = Scene(name, res=RES, **kw)
    v, w, l, c = oracle_render(sc, 0, spp, seed)
    return develop(sc, v, w, l, spp).astype(np.float64), c


def test_neutral_textures_change_nothing(built):
    """A constant texture of 1, a bilinear bitmap of equal texels (x its scale wrapper) and the flat normal map (0.5, 0.5, 1) give the
    untextured scene bit for bit (same random numbers)."""
    plain, cp = _render("tex_plain")
    for name in ("tex_const", "tex_bilinear_flat", "tex_normal_flat"):
        img, c = _render(name)
        assert np.array_equal(img, plain), name
        assert c == cp, name


def test_checkerboard_equals_the_same_pattern_as_a_nearest_bitmap(built):
    """checkerboard(0.8, 0.2) under a x4 uv transform against a 4 x 4 nearest-filtered, repeating bitmap of the same pattern (rows from
    the image's top: v flipped) — two texture kinds, one function of uv: identical images."""
    a, ca = _render("tex_checker")
    b, cb = _render("tex_bitmap")
    assert np.array_equal(a, b) and ca == cb


def test_checkerboard_modulates_the_reflectance(built):
    """Single scattering (max_depth 2), no Russian roulette: the textured render follows the SAME paths as the plain one (albedo 0.5)
    and every contribution carries the reflectance at its one surface vertex: per pixel the ratio is reflectance(uv) / 0.5 = 1.6 on
    the bright checks, 0.4 on the dark ones (pixels straddling a check boundary lie in between)."""
    tex, _ = _render("tex_checker", spp=8, rr=0, max_depth=2)
    plain, _ = _render("tex_plain", spp=8, rr=0, max_depth=2)
    ratio = tex.sum(axis=2) / plain.sum(axis=2)
    bright, dark = ratio > 1.0, ratio <= 1.0
    # interior pixels of a check: the whole 5 x 5 neighbourhood (3 x 3 reconstruction filter + sub-pixel jitter) is on the same side
    def interior(mask):
        m = mask.copy()
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                m &= np.roll(np.roll(mask, dy, axis=0), dx, axis=1)
        m[:2, :] = m[-2:, :] = False
        m[:, :2] = m[:, -2:] = False
        return m
    ib, idk = interior(bright), interior(dark)
    assert ib.sum() > 150 and idk.sum() > 150                           # four checks are in view, two of each kind
    assert np.abs(ratio[ib] - 1.6).max() < 1e-3 and np.abs(ratio[idk] - 0.4).max() < 1e-3
    # the pattern alternates: diagonal quadrants agree, neighbours differ
    q = lambda y, x: ratio[y, x] > 1.0
    assert q(8, 8) == q(24, 24) and q(8, 24) == q(24, 8) and q(8, 8) != q(8, 24)


def test_bilinear_bitmap_ramp(built):
    """2 x 1 texels (0.2 | 0.8), bilinear, clamped: reflectance(u) = 0.2 for u < 0.25, 0.8 for u > 0.75, linear in between
    (texture2d.hpp:287-303: texel centres at (i + 0.5) / width)."""
    tex, _ = _render("tex_bilinear_ramp", spp=8, rr=0, max_depth=2)
    plain, _ = _render("tex_plain", spp=8, rr=0, max_depth=2)
    ratio = (tex.sum(axis=2) / plain.sum(axis=2)).mean(axis=0)          # per column
    # one film column spans du = 2 * 3 tan(20 deg) / RES / 4 of the plane's u: a straight line of slope 1.2 du / 0.5 per column between the
    # clamped ends (the view covers u = 0.23 .. 0.77)
    cols = np.arange(3, RES - 3)
    slope, icpt = np.polyfit(cols, ratio[cols], 1)
    du = 2 * 3.0 * np.tan(np.radians(20.0)) / RES / 4
    assert abs(abs(slope) - 1.2 * du / 0.5) < 0.05 * 1.2 * du / 0.5, (slope, 1.2 * du / 0.5)   # (the film maps pixel i to i / (RES - 1): 3 %)
    assert np.abs(ratio[cols] - (slope * cols + icpt)).max() < 0.01
    assert abs(ratio.max() - 1.6) < 0.02 and abs(ratio.min() - 0.4) < 0.03


def test_mask_texture_punches_holes(built):
    """mask wrapper with a checkerboard opacity (1 / 0): opaque checks render like the plain plane, transparent ones pass every beam
    through (null lobe, wo = -wi) into empty space: black."""
    m, cm = _render("tex_mask", spp=8, rr=0, max_depth=2)
    plain, _ = _render("tex_plain", spp=8, rr=0, max_depth=2)
    ratio = m.sum(axis=2) / plain.sum(axis=2)
    opaque, hole = ratio > 0.5, ratio <= 0.5
    def interior(mask):
        mm = mask.copy()
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                mm &= np.roll(np.roll(mask, dy, axis=0), dx, axis=1)
        mm[:2, :] = mm[-2:, :] = False
        mm[:, :2] = mm[:, -2:] = False
        return mm
    io, ih = interior(opaque), interior(hole)
    assert io.sum() > 150 and ih.sum() > 150
    assert np.abs(ratio[io] - 1).max() < 1e-3 and ratio[ih].max() < 1e-6
    assert cm["surface_interactions"] > 0


def test_normal_map_equals_tilted_shading_normals(built):
    """A constant normal map (0.3, 0, 1)/|.| is the same shading frame as a mesh whose vertex normals are tilted that way
    (normalmap.hpp:54-61: the mapped normal is given in the unperturbed shading frame); `flip` mirrors x and y back."""
    nm, _ = _render("tex_normal_tilt", spp=8)
    mesh, _ = _render("tex_tilt_mesh", spp=8)
    plain, _ = _render("tex_plain", spp=8)
    assert np.abs(nm - mesh).max() < 1e-5 * mesh.max()
    assert np.abs(nm - plain).sum() > 0.05 * plain.sum()                 # towards the sun: brighter
    assert nm.sum() > plain.sum()
    fl, _ = _render("tex_normal_tilt_flipped", spp=8)
    assert np.array_equal(fl, nm)


def test_wrap_modes_and_texel_addressing():
    """The addressing arithmetic against a direct numpy restatement of texture2d_storage.hpp:80-97 (wrap_coord)."""
    def ref(mode, c, dim):
        if 0 <= c < dim:
            return c
        return {"black": -1, "white": -1, "clamp": min(max(c, 0), max(1, dim) - 1), "repeat": c % dim,
                "mirror": (lambda m2: 2 * dim - 1 - m2 if m2 >= dim else m2)(c % (2 * dim))}[mode]
    import ctypes as C
    from oracle_util import load_oracle
    lib = load_oracle()
    lib.kat_tex_wrap.restype = C.c_int
    lib.kat_tex_wrap.argtypes = [C.c_uint32, C.c_int, C.c_int]
    for mi, mode in enumerate(["black", "white", "clamp", "repeat", "mirror"]):
        for dim in (1, 2, 5):
            for c in range(-12, 13):
                assert lib.kat_tex_wrap(mi, c, dim) == ref(mode, c, dim), (mode, c, dim)


def test_bicubic_filter_is_the_reference_s_catmull_rom(built):
    """texture2d_t::bicubic_native (include/wt/bitmap/texture2d.hpp:316-343), the reference's DEFAULT bitmap filter (texture2d_storage.hpp:73): the 4 x 4
    texels around the sample, rows filtered first, with p1 + x/2 (p2 - p0) + x^2/2 (2 p0 - 5 p1 + 4 p2 - p3) + x^3/2 (-p0 + 3 p1 - 3 p2 + p3), wrapped
    per axis, negative lobes clamped to 0 — restated here in numpy; it interpolates the texels and reproduces a linear ramp exactly."""
    import ctypes as C
    from oracle_util import load_oracle
    lib = load_oracle()
    lib.kat_tex_bitmap.restype = C.c_float
    lib.kat_tex_bitmap.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float]
    lib.kat_tex_wrap.restype = C.c_int
    rng = np.random.default_rng(4)
    W, H = 7, 5
    tex = rng.uniform(0.1, 1.0, (H, W)).astype(np.float32)

    def cubic(x, p0, p1, p2, p3):
        return p1 + .5 * x * (-p0 + p2) + .5 * x * x * (2 * p0 - 5 * p1 + 4 * p2 - p3) + .5 * x * x * x * (-p0 + 3 * p1 - 3 * p2 + p3)

    def texel(x, y, uw, vw):
        x, y = lib.kat_tex_wrap(uw, int(x), W), lib.kat_tex_wrap(vw, int(y), H)
        if x < 0 or y < 0:
            return 0.0 if (uw if x < 0 else vw) == 0 else 1.0
        return float(tex[y, x])

    for uw, vw in ((3, 3), (2, 4), (0, 1)):   # repeat / clamp, mirror / black, white
        for _ in range(200):
            u, v = rng.uniform(-0.3, 1.3, 2)
            got = lib.kat_tex_bitmap(tex.ctypes.data, W, H, 2, uw, vw, u, v)
            x, y = np.float32(W) * np.float32(u) - np.float32(.5), np.float32(H) * (np.float32(1) - np.float32(v)) - np.float32(.5)   # (v is flipped: texture2d.hpp:368)
            ix, iy = int(np.floor(x)), int(np.floor(y))
            fx, fy = float(x - np.floor(x)), float(y - np.floor(y))
            rows = [cubic(fx, *(texel(ix + dx, iy + dy, uw, vw) for dx in (-1, 0, 1, 2))) for dy in (-1, 0, 1, 2)]
            want = max(0.0, cubic(fy, *rows))
            assert abs(got - want) < 2e-5 * max(1.0, abs(want)), (uw, vw, u, v, got, want)
    # at texel centres the filter returns the texel; on a linear ramp it is exact between them
    for x in range(1, W - 2):
        assert abs(lib.kat_tex_bitmap(tex.ctypes.data, W, H, 2, 2, 2, (x + .5) / W, 1 - 2.5 / H) - tex[2, x]) < 1e-6
    ramp = (0.1 + 0.1 * np.arange(8, dtype=np.float32))[None, :].repeat(4, 0).copy()
    for u in np.linspace(0.2, 0.8, 13):
        assert abs(lib.kat_tex_bitmap(ramp.ctypes.data, 8, 4, 2, 2, 2, u, 0.5) - (0.1 + 0.1 * (8 * u - .5))) < 1e-6
