#!/usr/bin/env python3
"""CPU only: the scene reader's file loaders (host/{exr,png,ply,obj}_loader.cpp, the PFM reader) under AddressSanitizer + UBSan on mutated files —
a damaged or hostile asset must end in the loader's error message, never in a read or write outside a buffer.  Builds a small harness with
g++ -fsanitize=address,undefined, writes valid seed files of every kind the tests cover (the writers of tests/test_xml_scene.py), flips 1-5 bytes /
truncates, and loads every mutant.   usage: fuzz_loaders.py [mutants per seed, default 500]     (round 5: 10,200 mutants, none reported)"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wave_tracer_amd", "csrc")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
d = tempfile.mkdtemp(prefix="wtgpu_fuzz_")
harness = os.path.join(d, "harness")
subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-Wno-unknown-pragmas", "-I" + CSRC, "-o", harness, "-x", "c++", "-",
                       *[os.path.join(CSRC, "host", f) for f in ("exr_loader.cpp", "png_loader.cpp", "ply_loader.cpp", "obj_loader.cpp")], "-lz"], check=True, input=b'''
#include "host/scene_builder.h"
#include <cstdio>
#include <cstring>
int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        uint32_t w, h, c;
        const char* e = std::strrchr(argv[i], '.');
        try {
            if (!std::strcmp(e, ".exr")) wth::load_exr(argv[i], w, h, c);
            else if (!std::strcmp(e, ".png")) wth::load_png(argv[i], w, h, c, 0, 2.2);
            else if (!std::strcmp(e, ".pfm")) wth::load_pfm(argv[i], w, h, c);
            else if (!std::strcmp(e, ".ply")) wth::load_ply(argv[i], false, 1.0);
            else { const std::string mt = "red"; wth::load_obj(argv[i], false, 1.0, (i & 1) ? &mt : nullptr); }
            std::printf("loaded\\n");
        } catch (const std::exception& ex) { std::printf("refused\\n"); }
    }
}
''')
src = open(os.path.join(ROOT, "tests", "test_xml_scene.py")).read()
ns = {"np": np}
exec("import zlib, struct\n" + src[src.index("def _write_png"):src.index("def test_png_bitmaps")] + src[src.index("def _write_exr"):src.index("def test_exr_bitmaps")], ns)
rng = np.random.default_rng(11)
seeds = []


def seed(name, writer=None, data=None):
    p = os.path.join(d, name)
    if data is not None:
        open(p, "wb").write(data)
    else:
        writer(p)
    seeds.append(p)


rgb = rng.uniform(0, 1, (19, 7, 3)).astype(np.float32)
rgb[3:9] = 0.25
for comp in (0, 1, 2, 3):
    seed(f"c{comp}.exr", lambda p: ns["_write_exr"](p, {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "half")}, compression=comp))
seed("g8.png", lambda p: ns["_write_png"](p, rng.integers(0, 256, (7, 6))))
seed("c16.png", lambda p: ns["_write_png"](p, rng.integers(0, 65536, (5, 4, 3)), depth=16))
seed("pal.png", lambda p: ns["_write_png"](p, rng.integers(0, 4, (6, 5)), palette=rng.integers(0, 256, (4, 3))))
seed("a.ply", data=b"ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\n"
     b"property float nz\nelement face 2\nproperty list uchar int vertex_indices\nend_header\n0 0 0 0 0 1\n1 0 0 0 0 1\n1 1 0 0 0 1\n0 1 0 0 0 1\n3 0 1 2\n3 0 2 3\n")
seed("b.ply", data=b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty float s\nproperty float t\n"
     b"element face 2\nproperty list uchar uint vertex_indices\nend_header\n" + b"".join(struct.pack("<5f", *v) for v in [(0, 0, 0, 0, 0), (1, 0, 0, 1, 0), (1, 1, 0, 1, 1), (0, 1, 0, 0, 1)])
     + struct.pack("<B3I", 3, 0, 1, 2) + struct.pack("<B3I", 3, 0, 2, 3))
open(os.path.join(d, "two.mtl"), "w").write("newmtl red\nnewmtl green\n")
seed("a.obj", data=b"mtllib two.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nusemtl red\nf 1/1/1 2/2/1 3/3/1 4/1/1\nusemtl green\n"
     b"f -4/1/1 -3/2/1 -1/3/1\nf 1/1/1 2/2/1 3/3/1\n")
seed("a.pfm", data=b"PF\n4 3\n-1.0\n" + rng.uniform(0, 1, (3, 4, 3)).astype("<f4").tobytes())
total = loaded = reports = 0
for base in seeds:
    b = bytearray(open(base, "rb").read())
    ext = os.path.splitext(base)[1]
    text = ext == ".obj" or (ext == ".ply" and b"ascii" in b[:40])
    batch = []
    for it in range(N):
        m = bytearray(b)
        for _ in range(rng.integers(1, 6)):
            m[int(rng.integers(0, len(m)))] = int(rng.choice(list(b" 0123456789-/.\nfvtn"))) if text else int(rng.integers(0, 256))
        if rng.random() < 0.2:
            m = m[:int(rng.integers(4, len(m)))]
        q = os.path.join(d, f"m{total}{ext}")
        open(q, "wb").write(bytes(m))
        batch.append(q)
        total += 1
    r = subprocess.run([harness] + batch, capture_output=True)
    loaded += r.stdout.count(b"loaded")
    print(f"  {os.path.basename(base):10s} {r.stdout.count(b'loaded'):5d} of {len(batch)} mutants loaded")
    err = r.stderr.decode("latin1")
    if r.returncode != 0 or "ERROR" in err or "runtime error" in err:
        reports += 1
        print(base, err[-2000:])
print(f"{total} mutants of {len(seeds)} seed files: {loaded} loaded, {total - loaded} refused with a message, {reports} sanitizer reports")
sys.exit(1 if reports else 0)
