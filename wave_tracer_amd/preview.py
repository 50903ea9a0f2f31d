"""Interactive preview in the `tev` image viewer: the reference pushes partially rendered films to tev over its TCP
protocol (include/wt/util/preview/preview_tev.hpp:22-47, src/util/preview_tev.cpp:33-262: CreateImage once per sensor, then
UpdateImageV3 packets with planar R, G, B channels; polarimetric films show their intensity plane; updates are rate-limited to one
per preview_update_interval() = 2 s, preview_interface.hpp:23-41).  Host-side only: nothing here is on the hot path.

Packet layout (tev's IPC, little endian): uint32 total length (including these 4 bytes), char type, then
  CreateImage   (4): bool grabFocus, cstring name, int32 width, int32 height, int32 nChannels, nChannels cstrings
  UpdateImageV3 (6): bool grabFocus, cstring name, int32 nChannels, nChannels cstrings, int32 x, y, width, height,
                     nChannels int64 offsets, nChannels int64 strides, float32 data[]"""
import socket
import struct
import time

import numpy as np

CREATE_IMAGE, UPDATE_IMAGE_V3 = 4, 6


def _cstr(s):
    return s.encode() + b"\0"


def _packet(ptype, body):
    payload = struct.pack("<b", ptype) + body
    return struct.pack("<I", 4 + len(payload)) + payload


def create_image_packet(name, width, height, channels=("R", "G", "B"), grab_focus=False):
    return _packet(CREATE_IMAGE, struct.pack("<?", grab_focus) + _cstr(name) + struct.pack("<iii", width, height, len(channels)) + b"".join(_cstr(c) for c in channels))


def update_image_packet(name, planes, x=0, y=0, channels=("R", "G", "B"), grab_focus=False):
    """planes: C x H x W float32 (planar, like the reference sends it: channel c at offset c * pixels, stride 1)."""
    planes = np.ascontiguousarray(planes, dtype=np.float32)
    C, H, W = planes.shape
    assert C == len(channels)
    n = H * W
    body = struct.pack("<?", grab_focus) + _cstr(name) + struct.pack("<i", C) + b"".join(_cstr(c) for c in channels)
    body += struct.pack("<iiii", x, y, W, H) + struct.pack("<%dq" % C, *[c * n for c in range(C)]) + struct.pack("<%dq" % C, *([1] * C))
    return _packet(UPDATE_IMAGE_V3, body + planes.tobytes())


class TevPreview:
    """preview_tev_t: connects lazily, creates one image per preview id and size, drops the connection on a failed write (the render goes
    on), honours the reference's 2 s update interval."""

    def __init__(self, host="127.0.0.1", port=14158, min_interval_s=2.0, timeout_s=10.0):
        self.addr, self.min_interval_s, self.timeout_s = (host, port), min_interval_s, timeout_s
        self._sock, self._images, self._last = None, {}, {}

    def _connect(self):
        if self._sock is None:
            try:
                self._sock = socket.create_connection(self.addr, timeout=self.timeout_s)
            except OSError:
                self._sock = None
        return self._sock is not None

    def _send(self, data):
        try:
            self._sock.sendall(data)
            return True
        except OSError:
            self.close()
            return False

    def close(self):
        if self._sock is not None:
            try:
                self._sock.close()
            finally:
                self._sock, self._images = None, {}

    def available(self, preview_id):
        return time.monotonic() - self._last.get(preview_id, -1e9) >= self.min_interval_s

    def update(self, preview_id, image, force=False):
        """image: H x W x 3 (or H x W x 1 / H x W: shown grey; H x W x C x 4 polarimetric films: the intensity plane).  Returns True if
        the viewer received it."""
        if not force and not self.available(preview_id):
            return False
        a = np.asarray(image, dtype=np.float32)
        if a.ndim == 4:
            a = a[..., 0]            # Stokes I (preview_tev.hpp:39-44)
        if a.ndim == 2:
            a = a[..., None]
        if a.shape[-1] == 1:
            a = np.repeat(a, 3, axis=-1)
        a = a[..., :3]
        H, W = a.shape[:2]
        name = f"wave_tracer '{preview_id}'"
        if not self._connect():
            return False
        if self._images.get(preview_id) != (W, H):
            if not self._send(create_image_packet(name, W, H)):
                return False
            self._images[preview_id] = (W, H)
        ok = self._send(update_image_packet(name, np.moveaxis(a, -1, 0)))
        if ok:
            self._last[preview_id] = time.monotonic()
        return ok


def render_with_preview(scene, spp, preview, seed=1, chunk_spp=8, preview_id="sensor", device=0):
    """Renders `spp` samples per element in chunks (wtgpu_render_progressive) and pushes the developed partial film to `preview` as the
    render advances (src/scene/render.cpp:306-368).  Needs a GPU.  Returns (value, weight, light) numpy films."""
    import torch
    from .render import alloc_films, develop
    if scene.device is None:
        scene.upload(device)
    dev = torch.device("cuda", device)
    films = alloc_films(scene, dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def on_progress(done, total):
        if preview.available(preview_id) or done == total:
            torch.cuda.synchronize(dev)
            v, w, l = (t.cpu().numpy() for t in films)
            per_elem = max(1, done // (scene.width * scene.height))
            img = develop(scene, v, w, l, per_elem)
            stokes = int(scene.info.stokes)
            # polarimetric films are [H][W][channels][4 Stokes components]: the viewer shows the intensity plane, like the reference's preview
            img = img.reshape(scene.height, scene.width, -1, stokes)[..., 0] if stokes > 1 else img.reshape(scene.height, scene.width, -1)
            preview.update(preview_id, img, force=done == total)
        return 0

    scene.render_progressive(*films, 0, spp, seed, chunk_spp=chunk_spp, progress=on_progress, stream=stream)
    torch.cuda.synchronize(dev)
    return tuple(t.cpu().numpy() for t in films)
