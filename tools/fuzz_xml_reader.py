#!/usr/bin/env python3
"""CPU only: the XML scene reader (csrc/host/xml_scene.cpp with its own XML parser, expression evaluator and function-texture compiler, on top of
scene_builder.cpp / scenes.cpp / the file loaders) under AddressSanitizer + UBSan on mutated scene files: characters replaced by syntax tokens,
stretches deleted, numbers replaced by extreme ones, lines duplicated or dropped.  Every mutant must load or be refused with a message.
usage: fuzz_xml_reader.py [seed] [mutants per scene file, default 200]
(round 5: 3,000 mutants of the four scene files of tests/data/xml, no sanitizer report; one finding: a sphere of tessellation 10^6 never finished —
the reader now bounds it with a message)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wave_tracer_amd", "csrc")
XML = os.path.join(ROOT, "tests", "data", "xml")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d = tempfile.mkdtemp(prefix="wtgpu_fuzz_xml_")
harness = os.path.join(d, "harness")
subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-Wno-unknown-pragmas", "-I" + CSRC, "-o", harness, "-x", "c++", "-",
                *[os.path.join(CSRC, "host", f) for f in ("scene_builder.cpp", "scenes.cpp", "xml_scene.cpp", "ply_loader.cpp", "obj_loader.cpp", "spectrum_db.cpp",
                                                          "png_loader.cpp", "exr_loader.cpp")], "-lz"], check=True, input=b'''
#include "host/scene_builder.h"
#include <cstdio>
int main(int argc, char** argv) {
    int loaded = 0, refused = 0;
    for (int i = 1; i < argc; ++i) {
        try {
            wth::scene_builder_t b;
            wth::scene_params_t p{};
            p.res = 8; p.max_depth = -1; p.fsd = -1; p.mis = p.rr = -1; p.mesh_detail = 0; p.lut_n_theta = 16; p.lut_m = 16; p.polarimetric = -1;
            wth::build_scene_from_xml(argv[i], {}, p, b);
            ++loaded;
        } catch (const std::exception& e) { ++refused; }
    }
    std::printf("%d loaded, %d refused with a message\\n", loaded, refused);
}
''')
shutil.copytree(os.path.join(XML, "parts"), os.path.join(d, "parts"))
tokens = ['"', "<", ">", "/", "$", "(", ")", "*", "+", "-", ",", " ", "0", "9", "e", "1e400", "nan", "°", "cm", "inf", "\n", "=", "{", "}", "%", "^", "."]
total = reports = 0
for name in ("objects.xml", "textured.xml", "single_slit.xml", "function_textures.xml"):
    t = open(os.path.join(XML, name), encoding="utf-8").read()
    batch = []
    for it in range(N):
        m = t
        for _ in range(rng.integers(1, 5)):
            k, pos = rng.random(), int(rng.integers(0, len(m)))
            if k < 0.4:
                m = m[:pos] + str(rng.choice(tokens)) + m[pos + 1:]
            elif k < 0.6:
                m = m[:pos] + m[pos + int(rng.integers(1, 30)):]
            elif k < 0.8:
                nums = list(re.finditer(r"-?\d+\.?\d*", m))
                if nums:
                    x = nums[int(rng.integers(0, len(nums)))]
                    m = m[:x.start()] + str(rng.choice(["0", "-1", "1e30", "1e-30", "99999999999", "-0.0", "3", "64", "1000000"])) + m[x.end():]
            else:
                ls = m.split("\n")
                a = int(rng.integers(0, len(ls)))
                if rng.random() < 0.5:
                    ls.insert(a, ls[int(rng.integers(0, len(ls)))])
                else:
                    del ls[a]
                m = "\n".join(ls)
        q = os.path.join(d, f"m{total}.xml")
        open(q, "w", encoding="utf-8").write(m)
        batch.append(q)
        total += 1
    try:
        r = subprocess.run([harness] + batch, capture_output=True, timeout=20 * N)
        out, err, rc = r.stdout.decode("latin1").strip().split("\n")[-1], r.stderr.decode("latin1"), r.returncode
    except subprocess.TimeoutExpired:
        out, err, rc = "a mutant did not finish", "", 1
    print(f"  {name:24s} {out}")
    if rc != 0 or "ERROR" in err or "runtime error" in err:
        reports += 1
        k = min([x for x in (err.find("ERROR"), err.find("runtime error")) if x >= 0] or [0])
        print(err[max(0, k - 300):k + 2500])
print(f"{total} mutants, {reports} batches with a sanitizer report or a crash")
sys.exit(1 if reports else 0)
