"""Rules the device sources keep by hand, checked on the text (CPU suite).

Round 5's compiler finding (wtgpu_kernels.h: wave_grab0): a persistent loop of a one-wavefront block that BEGINS with
`if (threadIdx.x == 0) item = atomicAdd(head, 1)` can have that branch threaded with a branch on the same condition at the END of its body; the
63 other lanes then go round an inner loop of their own, for ever.  The rule: whatever lane 0 alone fetches for its wavefront goes through
wave_grab0 / wave_grab / wave_grab_item, or has a convergent operation (a barrier, a ballot, a shuffle) in the lines right in front of the branch.
"""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wave_tracer_amd", "csrc")
LANE0_FETCH = re.compile(r"if \((?:threadIdx\.x(?: & 63u?)?|tid|lane)\s*==\s*(?:0u?|leader)\)\s*(?:\w+\s*=\s*)[^;]*(?:atomicAdd|_alloc)\(")
CONVERGENT = re.compile(r"__syncthreads\(\)|__ballot\(|__shfl|wave_barrier\(\)|wave_bcast0\(|wave_grab")


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "kernels_*.hip")) + [os.path.join(CSRC, "wtgpu_kernels.h")] + glob.glob(os.path.join(CSRC, "wt", "coop*.h")))


def test_lane0_fetches_sit_behind_a_convergent_operation():
    found, bad = 0, []
    for path in _sources():
        lines = open(path).read().split("\n")
        for i, l in enumerate(lines):
            if l.lstrip().startswith("//") or not LANE0_FETCH.search(l):
                continue
            found += 1
            before = "\n".join(x for x in lines[max(0, i - 6):i] if not x.lstrip().startswith("//"))
            if not CONVERGENT.search(before):
                bad.append(f"{os.path.relpath(path, ROOT)}:{i + 1}: {l.strip()}")
    assert found >= 4, "the pattern no longer matches the sources: update this test"
    assert not bad, "lane-0 fetch without a convergent operation in front of its branch:\n" + "\n".join(bad)


def test_one_wavefront_kernels_grab_through_the_helper():
    """No one-wavefront kernel hands its queue item through a __shared__ word any more (the 256-thread k_interact_c_hard does, between real barriers,
    the first of them in front of the branch)."""
    for path in _sources():
        src = open(path).read()
        for m in re.finditer(r"if \((?:threadIdx\.x|tid) == 0\) s_item = ", src):
            before = src[:m.start()].rstrip().split("\n")[-1]
            assert "__syncthreads();" in before, f"{os.path.relpath(path, ROOT)}: a shared queue word written without a barrier in front of the branch"
    hdr = open(os.path.join(CSRC, "wtgpu_kernels.h")).read()
    body = hdr[hdr.index("WT_D uint32_t wave_grab0("):]
    body = body[:body.index("}") + 1]
    assert body.index("wave_barrier()") < body.index("if (wave_first_lane())") < body.index("readfirstlane")


def _innermost_loops(asm_lines):
    """For every line of a function's assembly the header of the innermost loop its basic block belongs to (None outside loops), from the
    compiler's own block annotations."""
    out, loop, pending = [], None, None
    for l in asm_lines:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", l) or re.match(r"^; %bb\.\d+:(.*)$", l)
        if m:
            label = m.group(1) if l.startswith(".L") else None
            comment = m.group(m.lastindex)
            loop, pending = None, label
            mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=\d+", comment)
            if mm:
                loop = mm.group(1)
            elif "Loop Header" in comment and label:
                loop = label[2:]
        elif pending is not None and re.match(r"^\s*;", l) and "; wave barrier" not in l:
            if "Loop Header" in l:
                loop = pending[2:]
        else:
            pending = None
        out.append(loop)
    return out


def test_minimal_reproducer_of_the_split_persistent_loop(tmp_path):
    """tools/repro_persistent_loop.hip, compiled for gfx950 here (no GPU needed): in k_new — the form of wave_grab0 — the marker, lane 0's atomic and
    the readfirstlane that hands the item over sit in ONE loop.  What the compiler makes of k_old, the idiom of rounds 2-5, is reported: with ROCm 7.2
    the atomic is in the outer of two loops and the load of the shared word in the header of the inner one — the hang of round 5."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("hipcc not found")
    asm = tmp_path / "repro.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", str(asm), os.path.join(ROOT, "tools", "repro_persistent_loop.hip")],
                          stderr=subprocess.DEVNULL)
    text = open(asm).read().split("\n")

    def body(name):
        a = next(i for i, l in enumerate(text) if l.startswith(name + ":"))
        b = next(i for i in range(a, len(text)) if text[i].startswith(".Lfunc_end"))
        return text[a:b]

    new = body("k_new")
    loops = _innermost_loops(new)
    at = [i for i, l in enumerate(new) if "global_atomic_add" in l]
    marks = [i for i, l in enumerate(new) if "; wave barrier" in l]
    assert len(at) == 1 and len(marks) >= 2
    opening = max(i for i in marks if i < at[0])
    closing = min(i for i in marks if i > at[0])
    reads = [i for i in range(at[0], closing + 12) if "v_readfirstlane_b32" in new[i]]
    assert reads, "no readfirstlane behind the atomic"
    involved = {loops[opening], loops[at[0]], loops[closing]} | {loops[i] for i in reads}
    assert len(involved) == 1 and None not in involved, f"k_new: the queue grab is spread over loops {involved}"
    old = body("k_old")
    lo = _innermost_loops(old)
    a0 = next(i for i, l in enumerate(old) if "global_atomic_add" in l)
    r0 = next(i for i, l in enumerate(old) if "ds_read_b32" in l)
    print(f"k_old with this compiler: atomic in loop {lo[a0]}, load of the shared item in loop {lo[r0]} -> " + ("SPLIT (the round-5 hang)" if lo[a0] != lo[r0] else "one loop"))
