#!/usr/bin/env python3
"""Mutation fuzzing of the scene-file readers for memory safety: every mutated XML / PNG input must either load or be rejected with a
WtgpuError — never crash the process.  usage: fuzz_readers.py xml|png <seed> <count>"""
import sys, os, random, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from wave_tracer_amd import Scene
from wave_tracer_amd.api import WtgpuError
mode=sys.argv[1]; seed=int(sys.argv[2]); n=int(sys.argv[3])
rng=random.Random(seed)
tmp=tempfile.mkdtemp()
if mode=="xml":
    srcs=[open(os.path.join(ROOT, "tests", "data", "xml", f)).read() for f in ("single_slit.xml","objects.xml","textured.xml")]
    ok=err=0
    for i in range(n):
        s=rng.choice(srcs)
        b=bytearray(s.encode())
        for _ in range(rng.randint(1,4)):
            op=rng.randint(0,3); p=rng.randrange(len(b))
            if op==0: del b[p:p+rng.randint(1,30)]
            elif op==1: b[p]=rng.randrange(32,127)
            elif op==2: b[p:p]=bytes(rng.choice([b'<',b'>',b'"',b'$x',b'(',b')',b'&',b'/>',b'<bsdf type="scale">',b'1e999',b'-']))
            else: b=b[:p]
        path=os.path.join(tmp,"f.xml"); open(path,"wb").write(bytes(b))
        # includes / assets resolve relative to tmp: copy parts
        os.makedirs(os.path.join(tmp,"parts"),exist_ok=True)
        for q in os.listdir(os.path.join(ROOT, "tests", "data", "xml", "parts")):
            open(os.path.join(tmp,"parts",q),"wb").write(open(os.path.join(ROOT, "tests", "data", "xml", "parts", q),"rb").read())
        try:
            Scene.from_xml(path, lut=(16,16)); ok+=1
        except WtgpuError: err+=1
    print("xml",ok,err)
else:
    import numpy as np
    import test_xml_scene as T
    r=np.random.default_rng(seed)
    base=os.path.join(tmp,"b.png"); T._write_png(base, r.integers(0,256,(9,7,3)))
    data=open(base,"rb").read()
    TEX=T.TEX
    ok=err=0
    for i in range(n):
        b=bytearray(data)
        for _ in range(rng.randint(1,3)):
            if len(b) <= 9:
                break
            p=rng.randrange(8,len(b))
            op=rng.randint(0,2)
            if op==0: b[p]=rng.randrange(256)
            elif op==1: b=b[:p]
            else: del b[p:p+rng.randint(1,8)]
        path=os.path.join(tmp,"m.png"); open(path,"wb").write(bytes(b))
        try:
            Scene.from_xml(TEX, defines={"variant":2,"bitmap":path}); ok+=1
        except WtgpuError: err+=1
    print("png",ok,err)
