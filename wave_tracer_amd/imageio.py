"""Host-side film post-processing: the tonemapping operators of the reference's sensor responses and two
dependency-free image writers, so that developed films can be looked at.  Pure numpy: nothing here is on the hot path.

Reference: include/wt/sensor/response/tonemap/tonemap.hpp:36-231, src/sensor/response/tonemap.cpp:29-106 (operators and modes),
include/wt/spectrum/colourspace/RGB/RGB.hpp:152-200 (BT.709 luminance, sRGB transfer).  Colour maps: the reference takes them
from tinycolormap (a third-party header that is absent from the checkout); `turbo` here is the published polynomial fit of the
Turbo map, `grey` is the identity — tables are not reproduced.  Output files: PPM / PFM and OpenEXR (uncompressed scanline profile
written directly; the reference goes through the OpenEXR library, src/bitmap/write2d.cpp) with the reference's attributes and
per-Stokes file naming (src/main.cpp:243-255, 395-430)."""
import struct

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


# ---- OpenEXR (scanline, uncompressed, 32-bit float channels) ------------------------------------------------------------------------
# The reference writes its developed films as EXR with string attributes renderer / scene / sensor / samples (src/main.cpp:243-255,
# src/bitmap/write2d.cpp).  The OpenEXR library is not available here; the format's simplest profile is a few lines: magic, version,
# attribute list, line-offset table, one chunk per scanline (y, byte count, channels in alphabetical order, each a row of floats).
def _exr_attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def write_exr(path, img, attributes=None, channel_names=None):
    """img: H x W or H x W x C float array (linear).  Channels are named Y (1), R G B (3) or R G B A (4) unless `channel_names` is given.
    attributes: dict of string attributes (the reference sets renderer, scene, sensor, samples)."""
    a = np.asarray(img, dtype=np.float32)
    if a.ndim == 2:
        a = a[..., None]
    H, W, C = a.shape
    names = channel_names or {1: ["Y"], 3: ["R", "G", "B"], 4: ["R", "G", "B", "A"]}.get(C) or [f"C{i}" for i in range(C)]
    assert len(names) == C
    order = sorted(range(C), key=lambda i: names[i])            # channels are stored in alphabetical order
    chlist = b"".join(names[i].encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for i in order) + b"\0"   # pixel type 2 = FLOAT
    box = struct.pack("<iiii", 0, 0, W - 1, H - 1)
    hdr = struct.pack("<ii", 20000630, 2)
    hdr += _exr_attr("channels", "chlist", chlist)
    hdr += _exr_attr("compression", "compression", b"\0")
    hdr += _exr_attr("dataWindow", "box2i", box) + _exr_attr("displayWindow", "box2i", box)
    hdr += _exr_attr("lineOrder", "lineOrder", b"\0")
    hdr += _exr_attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += _exr_attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0))
    hdr += _exr_attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    for k, v in (attributes or {}).items():
        hdr += _exr_attr(k, "string", str(v).encode())
    hdr += b"\0"
    row_bytes = 4 * W * C
    first = len(hdr) + 8 * H
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(struct.pack("<%dQ" % H, *[first + y * (8 + row_bytes) for y in range(H)]))
        for y in range(H):
            f.write(struct.pack("<ii", y, row_bytes))
            for i in order:
                f.write(np.ascontiguousarray(a[y, :, i]).tobytes())


def read_exr(path):
    """Reads back what write_exr writes (uncompressed scanline files with FLOAT channels): (image H x W x C in the file's channel
    order, channel names, string attributes)."""
    b = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", b, 0)
    assert magic == 20000630 and (version & 0xff) == 2 and not (version & 0x200), "scanline OpenEXR expected"
    p = 8
    attrs, names, box = {}, [], None
    while b[p] != 0:
        e = b.index(b"\0", p)
        name = b[p:e].decode()
        p = e + 1
        e = b.index(b"\0", p)
        typ = b[p:e].decode()
        p = e + 1
        (n,) = struct.unpack_from("<i", b, p)
        p += 4
        val = b[p:p + n]
        p += n
        if typ == "chlist":
            q = 0
            while val[q] != 0:
                e2 = val.index(b"\0", q)
                names.append(val[q:e2].decode())
                assert struct.unpack_from("<i", val, e2 + 1)[0] == 2, "FLOAT channels expected"
                q = e2 + 1 + 16
        elif typ == "box2i" and name == "dataWindow":
            box = struct.unpack("<iiii", val)
        elif typ == "compression":
            assert val == b"\0", "uncompressed files only"
        elif typ == "string":
            attrs[name] = val.decode()
    p += 1
    W, H, C = box[2] - box[0] + 1, box[3] - box[1] + 1, len(names)
    offs = struct.unpack_from("<%dQ" % H, b, p)
    img = np.zeros((H, W, C), np.float32)
    for y in range(H):
        yy, nbytes = struct.unpack_from("<ii", b, offs[y])
        assert nbytes == 4 * W * C
        row = np.frombuffer(b, np.float32, W * C, offs[y] + 8).reshape(C, W)
        img[yy - box[1]] = row.T
    return img, names, attrs


def write_developed(out_dir, sensor_name, scene_name, developed, spe, stokes=1, renderer="wave_tracer_amd"):
    """Writes a developed film the way the reference's CLI names its outputs (src/main.cpp:243-255, 331-430): `<sensor>.exr`, or one
    file per Stokes component `<sensor>_I.exr`, `_Q`, `_U`, `_V` for polarimetric sensors (film planes [H][W][C][S], S = 4), each with
    the attributes renderer / scene / sensor / samples.  Returns the paths."""
    import os
    a = np.asarray(developed, dtype=np.float32)
    H, W = a.shape[:2]
    attrs = {"renderer": renderer, "scene": scene_name, "sensor": sensor_name, "samples": str(int(spe))}
    paths = []
    if stokes == 1:
        paths.append(os.path.join(out_dir, sensor_name + ".exr"))
        write_exr(paths[-1], a.reshape(H, W, -1), attrs)
    else:
        planes = a.reshape(H, W, -1, stokes)
        for s, suffix in enumerate(["I", "Q", "U", "V"][:stokes]):
            paths.append(os.path.join(out_dir, f"{sensor_name}_{suffix}.exr"))
            write_exr(paths[-1], planes[..., s], attrs)
    return paths
