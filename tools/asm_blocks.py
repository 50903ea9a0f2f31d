#!/usr/bin/env python3
"""Basic blocks of one kernel in the device assembly (tools/kernel_resources.sh leaves it in /tmp/wtgpu_dev.s): per block the number of
instructions, VALU, scratch / global / LDS accesses and its backward branches — where the loops are and what they carry to memory.
usage: asm_blocks.py <kernel substring> [asm] [--all]"""
import re
import sys

want = sys.argv[1]
asm = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "/tmp/wtgpu_dev.s"
show_all = "--all" in sys.argv
lines = open(asm).read().split("\n")
start = end = None
for i, l in enumerate(lines):
    if start is None and re.match(r"^_Z\w*%s\w*:" % re.escape(want), l):
        start = i
    elif start is not None and l.startswith(".Lfunc_end"):
        end = i
        break
if start is None:
    sys.exit("kernel not found: " + want)
body = lines[start:end]
blocks = []
cur = {"name": "entry", "ins": []}
blocks.append(cur)
for l in body[1:]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = {"name": m.group(1), "ins": []}
        blocks.append(cur)
        continue
    t = l.strip()
    if not t or t.startswith((".", ";")):
        continue
    cur["ins"].append(t)
idx = {b["name"]: i for i, b in enumerate(blocks)}
tot = {}
print("%5s %-12s %6s %6s %5s %5s %5s %5s %5s  %s" % ("#", "block", "instr", "valu", "s_ld", "s_st", "g_ld", "g_st", "lds", "backward branches (target #)"))
for i, b in enumerate(blocks):
    c = lambda p: sum(1 for x in b["ins"] if x.startswith(p))
    tg = [x.split()[-1] for x in b["ins"] if x.startswith(("s_cbranch", "s_branch"))]
    back = [(t, idx[t]) for t in tg if t in idx and idx[t] <= i]
    row = (len(b["ins"]), c("v_"), c("scratch_load"), c("scratch_store"), c("global_load"), c("global_store"), c("ds_"))
    for k, v in zip(("instr", "valu", "s_ld", "s_st", "g_ld", "g_st", "lds"), row):
        tot[k] = tot.get(k, 0) + v
    if show_all or row[2] or row[3] or back or row[0] >= 150:
        print("%5d %-12s %6d %6d %5d %5d %5d %5d %5d  %s" % ((i, b["name"]) + row + (" ".join("%s(#%d)" % t for t in back),)))
print("total", tot, "blocks", len(blocks))
