#!/usr/bin/env python3
"""Checks the device assembly for the failure of round 5 (wtgpu_kernels.h: wave_grab0): a queue grab whose pieces — the convergent
marker in front, lane 0's atomic on the queue head, the readfirstlane that hands its result to the wavefront — the compiler has put into DIFFERENT
loops (the atomic in an outer loop, the read in an inner one that 63 lanes then go round alone).  For every grab in every kernel: the innermost loop
of the opening marker, of the atomic, of every readfirstlane and of the closing marker must be the same one.
usage: check_persistent_loops.py [asm]      (default /tmp/wtgpu_dev.s, which tools/kernel_resources.sh leaves behind)"""
import re
import sys

asm = sys.argv[1] if len(sys.argv) > 1 else "/tmp/wtgpu_dev.s"
lines = open(asm).read().split("\n")
kernel = None
loop = None          # innermost loop header of the current block
pending = None       # a label whose loop annotation continues on the following comment lines
sites, bad = [], []
grab = None          # state of the grab being read: dict(kernel, line, marker_loop, atomic_loop)
n_kernels = 0
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kernel, loop, grab = m.group(1), None, None
        n_kernels += 1
        continue
    if l.startswith(".Lfunc_end"):
        kernel = None
        continue
    if kernel is None:
        continue
    m = re.match(r"^(\.LBB\d+_\d+):(.*)$", l) or re.match(r"^; %bb\.\d+:(.*)$", l)
    if m:
        label = m.group(1) if l.startswith(".L") else None
        comment = m.group(m.lastindex)
        loop = None
        pending = label
        mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", comment)
        if mm:
            loop = mm.group(1)
        elif "This Loop Header" in comment or "This Inner Loop Header" in comment:
            loop = label[2:] if label else None
        continue
    if pending is not None and re.match(r"^\s*;", l) and "; wave barrier" not in l:      # continuation of a header's annotation
        if "This Loop Header" in l or "This Inner Loop Header" in l:
            loop = pending[2:] if pending else loop
        continue
    pending = None
    if "; wave barrier" in l:
        if grab is not None and grab.get("read_loop") is not None and i - grab["read_line"] <= 24:       # the closing marker of wave_grab0: right behind its readfirstlane
            # (a barrier of a one-wavefront block leaves the same marker; an atomic and the optimiser's readfirstlane between two of THOSE are not a grab)
            grab["close_loop"] = loop
            sites.append(grab)
            if len({grab["marker_loop"], grab["atomic_loop"], grab["close_loop"]} | grab["read_loops"]) != 1:
                bad.append(grab)
            grab = None
        else:
            grab = {"kernel": kernel, "line": i + 1, "marker_loop": loop, "atomic_loop": None, "read_loop": None, "read_loops": set()}
        continue
    if grab is not None:
        if i + 1 - grab["line"] > 160:       # not a grab (a marker of wave_bcast0 or of a barrier): forget it
            grab = None
        elif re.search(r"\b(global|flat)_atomic_add\b", l) and grab["atomic_loop"] is None:
            grab["atomic_loop"] = loop
        elif "v_readfirstlane_b32" in l and grab["atomic_loop"] is not None:      # the optimiser's own and the helper's: all of them
            grab["read_loop"] = loop
            grab["read_line"] = i
            grab["read_loops"].add(loop)
per = {}
for s in sites:
    per[s["kernel"]] = per.get(s["kernel"], 0) + 1
print(f"{n_kernels} functions, {len(sites)} queue grabs (marker / lane-0 atomic / readfirstlane / marker) in {len(per)} kernels")
bad_kernels = {b["kernel"] for b in bad}
for k in sorted(per):
    name = re.search(r"\d+(k_\w+?)E", k)
    print(f"  {name.group(1) if name else k:28s} {per[k]} grab(s), " + ("SPLIT over two loops" if k in bad_kernels else "each inside one loop"))
for b in bad:
    print(f"SPLIT GRAB in {b['kernel']} at line {b['line']}: marker in loop {b['marker_loop']}, atomic in {b['atomic_loop']}, readfirstlane in {sorted(map(str, b['read_loops']))}, closing marker in {b['close_loop']}")
sys.exit(1 if bad or not sites else 0)
