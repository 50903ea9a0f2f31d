#!/usr/bin/env python3
"""Static instruction mix per kernel from the device assembly tools/kernel_resources.sh leaves in /tmp/wtgpu_dev.s:
global / scratch / LDS loads and stores, VALU / SALU counts.  usage: kernel_isa_stats.py [asm] [kernel substring ...]"""
import re
import sys
asm = sys.argv[1] if len(sys.argv) > 1 else "/tmp/wtgpu_dev.s"
want = sys.argv[2:]
cur = None
stats = {}
for line in open(asm):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        stats[cur] = {}
        continue
    if cur is None:
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        cur = None
        continue
    t = line.strip().split()
    if not t or t[0].startswith((".", ";")):
        continue
    op = t[0]
    key = None
    if op.startswith("global_load"): key = "gload"
    elif op.startswith("global_store"): key = "gstore"
    elif op.startswith("global_atomic"): key = "gatomic"
    elif op.startswith("scratch_load"): key = "sload"
    elif op.startswith("scratch_store"): key = "sstore"
    elif op.startswith("buffer_load"): key = "bload"
    elif op.startswith("buffer_store"): key = "bstore"
    elif op.startswith("ds_"): key = "lds"
    elif op.startswith("s_load") or op.startswith("s_buffer_load"): key = "smem"
    elif op.startswith("v_"): key = "valu"
    elif op.startswith("s_"): key = "salu"
    if key:
        stats[cur][key] = stats[cur].get(key, 0) + 1
        if key in ("gload", "gstore", "sload", "sstore", "bload", "bstore"):
            w = {"dwordx4": 4, "dwordx3": 3, "dwordx2": 2, "dword": 1, "b128": 4, "b96": 3, "b64": 2, "b32": 1}
            for k, v in w.items():
                if op.endswith(k):
                    stats[cur][key + "_dw"] = stats[cur].get(key + "_dw", 0) + v
                    break
cols = ["valu", "salu", "smem", "gload", "gload_dw", "gstore", "gstore_dw", "gatomic", "sload", "sload_dw", "sstore", "sstore_dw", "lds"]
print("%-22s" % "kernel" + "".join("%10s" % c for c in cols))
for k, s in stats.items():
    m = re.search(r"\d+(k_\w+?)E", k)
    name = m.group(1) if m else k[:22]
    if want and not any(x in name for x in want):
        continue
    if not s:
        continue
    print("%-22s" % name + "".join("%10d" % s.get(c, 0) for c in cols))
