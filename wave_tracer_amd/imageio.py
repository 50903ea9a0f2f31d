"""Host-side film post-processing: the tonemapping operators of the reference's sensor responses and two
dependency-free image writers, so that developed films can be looked at.  Pure numpy: nothing here is on the hot path.

Reference: include/wt/sensor/response/tonemap/tonemap.hpp:36-231, src/sensor/response/tonemap.cpp:29-106 (operators and modes),
include/wt/spectrum/colourspace/RGB/RGB.hpp:152-200 (BT.709 luminance, sRGB transfer).  Colour maps: the reference takes them
from tinycolormap (a third-party header that is absent from the checkout); `turbo` here is the published polynomial fit of the
Turbo map, `grey` is the identity — tables are not reproduced."""
import numpy as np


def luminance(rgb):
    """BT.709 luminance of a linear RGB image (RGB.hpp:152-157)."""
    return np.maximum(0.0, rgb[..., 0] * .2126 + rgb[..., 1] * .7152 + rgb[..., 2] * .0722)


def srgb_from_linear(x):
    """sRGB transfer function (RGB.hpp:191-199); input clamped to [0,1] like tonemap.cpp:59."""
    x = np.clip(x, 0.0, 1.0)
    return np.where(x <= .0031308, np.maximum(0.0, 12.92 * x), 1.055 * np.power(np.maximum(x, 1e-30), 1 / 2.4) - .055)


def tonemap_gamma(x, gamma):
    """x^(1/gamma) of the clamped value (tonemap.cpp:57)."""
    return np.power(np.clip(x, 0.0, 1.0), 1.0 / gamma)


def tonemap_db(x, db_min, db_max):
    """Decibel mapping (tonemap.cpp:61-68): 0 stays 0; otherwise clamp01((10 log10 x - db_min) / (db_max - db_min))."""
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        db = 10.0 * np.log10(x)
        out = np.clip((db - db_min) / (db_max - db_min), 0.0, 1.0)
    return np.where(x == 0, 0.0, out)


def colourmap(v, name="turbo"):
    """Maps v in [0,1] to RGB.  'grey' or 'turbo' (polynomial approximation of Google's Turbo map)."""
    v = np.clip(np.asarray(v, dtype=np.float64), 0.0, 1.0)
    if name == "grey":
        return np.stack([v, v, v], axis=-1)
    if name != "turbo":
        raise ValueError(f"unknown colour map {name}")
    v4 = np.stack([np.ones_like(v), v, v * v, v * v * v], axis=-1)
    v2 = np.stack([v4[..., 2] * v4[..., 2], v4[..., 3] * v4[..., 2]], axis=-1)
    r4, g4, b4 = (.13572138, 4.61539260, -42.66032258, 132.13108234), (.09140261, 2.19418839, 4.84296658, -14.18503333), (.10667330, 12.64194608, -60.58204836, 110.36276771)
    r2, g2, b2 = (-152.94239396, 59.28637943), (4.27729857, 2.82956604), (-89.90310912, 27.34824973)
    rgb = [v4 @ np.array(c4) + v2 @ np.array(c2) for c4, c2 in ((r4, r2), (g4, g2), (b4, b2))]
    return np.clip(np.stack(rgb, axis=-1), 0.0, 1.0)


def tonemap(img, op="sRGB", mode="select", gamma=2.2, db_range=(-60.0, 0.0), cmap="turbo"):
    """tonemap_t::operator() on a developed film [H,W,1] or [H,W,3] -> RGB in [0,1].
    op: linear | gamma | sRGB | dB;  mode: select (colour map for monochrome, per channel for RGB) | normal | colourmap."""
    img = np.asarray(img, dtype=np.float64)
    if img.ndim == 2:
        img = img[..., None]
    if img.shape[-1] not in (1, 3):
        raise ValueError("single-channel or RGB film expected")
    f = {"linear": lambda x: x, "gamma": lambda x: tonemap_gamma(x, gamma), "sRGB": srgb_from_linear,
         "dB": lambda x: tonemap_db(x, db_range[0], db_range[1])}[op]
    if op == "dB":
        mode = "colourmap"   # tonemap_t::create_dB
    mono = img.shape[-1] == 1
    use_cm = mode == "colourmap" or (mode == "select" and mono)
    if use_cm:
        return colourmap(f(img[..., 0] if mono else luminance(img)), cmap)
    out = f(img)
    return np.repeat(out, 3, axis=-1) if mono else out


def write_ppm(path, rgb01):
    """8-bit binary PPM."""
    a = (np.clip(rgb01, 0.0, 1.0) * 255.0 + .5).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(np.ascontiguousarray(a).tobytes())


def write_pfm(path, img):
    """32-bit float PFM (linear data; 1 or 3 channels), bottom row first, little endian."""
    a = np.asarray(img, dtype=np.float32)
    if a.ndim == 2:
        a = a[..., None]
    with open(path, "wb") as f:
        f.write((b"PF" if a.shape[-1] == 3 else b"Pf") + b"\n%d %d\n-1.0\n" % (a.shape[1], a.shape[0]))
        f.write(np.ascontiguousarray(a[::-1]).tobytes())
