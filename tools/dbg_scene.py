import sys, ctypes as C, numpy as np, hashlib
sys.path.insert(0, '/root/repo')
from wave_tracer_amd import Scene
for md in (0, 1):
    sc = Scene("cornell_box", res=16, mesh_detail=md, lut=(32, 32))
    print("mesh_detail", md, "n_tris", sc.info.n_tris, "n_edges", sc.info.n_edges)
    d = sc.host_desc()
    try:
        dd = d.contents if hasattr(d, 'contents') else d
        n = dd.n_tris; 
        tg = np.ctypeslib.as_array(C.cast(dd.tri_geo, C.POINTER(C.c_float)), shape=(n*16,)) if hasattr(dd,'tri_geo') else None
        tm = dd.tri_meta if hasattr(dd,'tri_meta') else None
        print([f for f,_ in dd._fields_][:40])
    except Exception as e: print("desc err", e)
