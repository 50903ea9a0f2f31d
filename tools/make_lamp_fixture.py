"""Writes tests/data/xml/parts/lamp.pfm: the 8 x 6 RGB radiance texture of tests/data/xml/textured_emitter.xml (a bright warm blob, a dim blue
corner and a black band, so the sampling tables hold empty, dim and bright cells).  Deterministic; the file is committed."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wave_tracer_amd.imageio import write_pfm  # noqa: E402


def lamp_texture():
    h, w = 6, 8
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    blob = np.exp(-((x - 5.0) ** 2 + (y - 2.0) ** 2) / 3.0).astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    img[..., 0] = 1.0 * blob
    img[..., 1] = 0.7 * blob
    img[..., 2] = 0.3 * blob
    img[4:, :3, 2] += 0.25   # a dim blue corner
    img[:, 1, :] = 0.0       # a black band
    return img


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "data", "xml", "parts", "lamp.pfm")
    write_pfm(out, lamp_texture())
    print(out, os.path.getsize(out), "bytes")
