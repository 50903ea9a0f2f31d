#!/usr/bin/env python3
"""Bring-up aid for a kernel that does not end: renders a plt_path scene with a library built with -DWTGPU_FSD_WATCH, whose k_path_fsd (and the ray
traversal it calls) write their progress into a HOST-MAPPED buffer; a second thread prints the buffer after a few seconds and ends the process.
usage: WTGPU_LIB=.../libwtgpu_watch.so python tools/watch_path.py [scene] [res] [spp] [seconds]"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from wave_tracer_amd import Scene, render  # noqa: E402
from wave_tracer_amd.api import load_library  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sunlit_path"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 12.0
torch.cuda.init()
buf = torch.zeros(64 * 16, dtype=torch.int32).pin_memory()
lib = load_library()
lib.wtgpu_debug_set_watch.argtypes = [C.c_void_p]
print("set_watch rc", lib.wtgpu_debug_set_watch(C.c_void_p(buf.data_ptr())), flush=True)
done = []


def work():
    sc = Scene(name, res=res)
    v, w, l = render(sc, spp, seed=5, device=0)
    done.append(float(v.sum()))


t = threading.Thread(target=work, daemon=True)
t.start()
t.join(secs)
names = ["item", "n", "walk", "n_edges", "round", "i_begin", "i_after_f_edge", "edge_offset", "node_steps", "stack_size", "leaf_steps", "leaf_count", "loop_iters", "i_lane0", "n_edges_seen", "before_shadow"]
print("render finished:" if done else "render still running after %.0f s:" % secs, done, flush=True)
b = buf.view(64, 16)
for r in range(64):
    row = b[r].tolist()
    if any(row):
        print("block %2d: " % r + " ".join("%s=%d" % (n, x & 0xFFFFFFFF) for n, x in zip(names, row)), flush=True)
time.sleep(1.0)
b2 = buf.view(64, 16)
print("one second later (counters that still move = a loop that is running):", flush=True)
for r in range(64):
    row = b2[r].tolist()
    if any(row):
        print("block %2d: node_steps=%d leaf_steps=%d stack=%d leaf_count=%d i=%d loop_iters=%d before_shadow=%d" % (r, row[8] & 0xFFFFFFFF, row[10] & 0xFFFFFFFF, row[9], row[11] & 0xFFFFFFFF, row[5], row[12] & 0xFFFFFFFF, row[15] & 0xFFFFFFFF), flush=True)
os._exit(0)
