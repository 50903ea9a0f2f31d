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
