"""tev preview (src/util/preview_tev.cpp:33-262): packet layout of tev's IPC protocol against a byte-level parse, and
the client's behaviour (one CreateImage per image, UpdateImageV3 with planar channels, rate limiting, survival of a dead viewer)
against a local TCP server standing in for tev."""
import socket
import struct
import threading
import time

import numpy as np

from wave_tracer_amd.preview import CREATE_IMAGE, UPDATE_IMAGE_V3, TevPreview, create_image_packet, update_image_packet


def _parse(buf):
    """Splits a byte stream into tev packets and decodes the two kinds the client sends."""
    out, p = [], 0
    while p + 4 <= len(buf):
        (n,) = struct.unpack_from("<I", buf, p)
        if p + n > len(buf):
            break
        body, q = buf[p + 4:p + n], 0
        ptype = body[0]
        q = 1
        grab = body[q] != 0
        q += 1

        def cstr():
            nonlocal q
            e = body.index(b"\0", q)
            s = body[q:e].decode()
            q = e + 1
            return s
        name = cstr()
        if ptype == CREATE_IMAGE:
            w, h, c = struct.unpack_from("<iii", body, q)
            q += 12
            names = [cstr() for _ in range(c)]
            assert q == len(body)
            out.append(("create", name, grab, w, h, names))
        elif ptype == UPDATE_IMAGE_V3:
            (c,) = struct.unpack_from("<i", body, q)
            q += 4
            names = [cstr() for _ in range(c)]
            x, y, w, h = struct.unpack_from("<iiii", body, q)
            q += 16
            offs = struct.unpack_from("<%dq" % c, body, q)
            q += 8 * c
            strides = struct.unpack_from("<%dq" % c, body, q)
            q += 8 * c
            data = np.frombuffer(body, np.float32, (len(body) - q) // 4, q)
            assert len(data) == max(o + (w * h - 1) * s + 1 for o, s in zip(offs, strides))      # preview_tev.cpp:101-109
            out.append(("update", name, grab, names, (x, y, w, h), offs, strides, data))
        else:
            raise AssertionError(ptype)
        p += n
    return out


def test_packet_layout():
    c = create_image_packet("wave_tracer 'cam'", 7, 5)
    (kind, name, grab, w, h, names), = _parse(c)
    assert (kind, name, grab, w, h, names) == ("create", "wave_tracer 'cam'", False, 7, 5, ["R", "G", "B"])
    assert struct.unpack_from("<I", c, 0)[0] == len(c) and c[4] == 4
    planes = np.arange(3 * 5 * 7, dtype=np.float32).reshape(3, 5, 7)
    u = update_image_packet("wave_tracer 'cam'", planes)
    (kind, name, grab, names, box, offs, strides, data), = _parse(u)
    assert kind == "update" and names == ["R", "G", "B"] and box == (0, 0, 7, 5)
    assert offs == (0, 35, 70) and strides == (1, 1, 1)                     # planar: channel c at c * pixels (preview_tev.cpp:236-244)
    assert np.array_equal(data.reshape(3, 5, 7), planes) and u[4] == 6


class _FakeTev(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.srv = socket.socket()
        self.srv.bind(("127.0.0.1", 0))
        self.srv.listen(1)
        self.port = self.srv.getsockname()[1]
        self.data = b""
        self.start()

    def run(self):
        conn, _ = self.srv.accept()
        conn.settimeout(2.0)
        try:
            while True:
                b = conn.recv(1 << 16)
                if not b:
                    break
                self.data += b
        except OSError:
            pass
        conn.close()


def test_client_creates_once_updates_and_rate_limits():
    srv = _FakeTev()
    pv = TevPreview("127.0.0.1", srv.port, min_interval_s=0.2)
    rgb = np.random.default_rng(0).uniform(0, 1, (6, 9, 3)).astype(np.float32)
    assert pv.update("camera", rgb)
    assert not pv.update("camera", 2 * rgb)                       # too soon (preview_update_interval)
    time.sleep(0.25)
    assert pv.update("camera", 2 * rgb)
    mono = np.ones((6, 9, 1), np.float32)
    assert pv.update("pattern", mono)                             # a second sensor: its own image, grey
    stokes = np.zeros((6, 9, 3, 4), np.float32)
    stokes[..., 0] = rgb
    assert pv.update("pol", stokes)                               # polarimetric film: the intensity plane
    pv.close()
    srv.join(3.0)
    pk = _parse(srv.data)
    kinds = [(k[0], k[1]) for k in pk]
    assert kinds == [("create", "wave_tracer 'camera'"), ("update", "wave_tracer 'camera'"), ("update", "wave_tracer 'camera'"),
                     ("create", "wave_tracer 'pattern'"), ("update", "wave_tracer 'pattern'"),
                     ("create", "wave_tracer 'pol'"), ("update", "wave_tracer 'pol'")]
    assert pk[0][3:5] == (9, 6)
    assert np.array_equal(pk[1][7].reshape(3, 6, 9), np.moveaxis(rgb, -1, 0))
    assert np.array_equal(pk[2][7].reshape(3, 6, 9), np.moveaxis(2 * rgb, -1, 0))
    assert np.array_equal(pk[4][7].reshape(3, 6, 9), np.ones((3, 6, 9), np.float32))
    assert np.array_equal(pk[6][7].reshape(3, 6, 9), np.moveaxis(rgb, -1, 0))


def test_client_survives_without_a_viewer():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]                                 # (closed again: nothing listens there)
    pv = TevPreview("127.0.0.1", port, timeout_s=0.5)
    assert pv.update("camera", np.zeros((2, 2, 3), np.float32)) is False       # render goes on (preview_tev.cpp:211-216)
