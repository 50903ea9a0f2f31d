"""CPU study (the checker's instrumented policy, oracle_profile_heavy): which walks' cone queries exceed the per-lane work budget, and what could have
told beforehand.  Prints, for the headline workload, how well `cone radius at the axis hit / bounding-sphere radius of the axis-hit triangle` separates them."""
import sys, os, ctypes as C, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from wave_tracer_amd.api import Scene
from oracle_util import load_oracle
budget = int(sys.argv[1]) if len(sys.argv) > 1 else 96
sc = Scene("cornell_box", res=1440, mesh_detail=1)
lib = load_oracle()
lib.oracle_profile_heavy.restype = C.c_uint64
lib.oracle_profile_heavy.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64]
cap = 2_000_000
out = np.zeros((cap, 8), np.float32)
n = lib.oracle_profile_heavy(sc.host_desc(), 7, 97, out.ctypes.data, cap)
o = out[:min(n, cap)]
units, rc, rt, ad, ta, x0, em, nq = o.T
heavy = units > budget
print("calls", len(o), "heavy (max query >", budget, "units):", heavy.mean(), " share of all cone work units in heavy calls (lower bound, max query only):", units[heavy].sum() / units.sum())
hit = rc >= 0
print("axis miss:", (~hit).mean(), "heavy among axis-miss", heavy[~hit].mean() if (~hit).any() else 0)
ratio = np.where(hit, rc / np.maximum(rt, 1e-30), np.inf)
for thr in (0.5, 1, 2, 3, 4, 6, 8, 12, 16):
    pred = ratio > thr
    tp = (pred & heavy).sum(); fp = (pred & ~heavy).sum(); fn = (~pred & heavy).sum()
    # wasted per-lane units: heavy walks not predicted spend `budget` units before the hand-over; light walks predicted heavy go to a wavefront for nothing
    print(f"ratio > {thr:5}: predicted {pred.mean():.3%}  recall {tp / max(1, heavy.sum()):.3f}  precision {tp / max(1, pred.sum()):.3f}  light walks sent to a wavefront {fp / len(o):.3%} (their mean units {units[pred & ~heavy].mean() if fp else 0:.0f})  heavy walks missed {fn / len(o):.3%}")
np.save("/tmp/prof_heavy.npy", o)
