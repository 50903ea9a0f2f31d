#!/usr/bin/env python3
"""Generates the regression fixtures in this directory FROM THE ORACLE (oracle/_build/liboracle.so):

    python tests/golden/make_golden.py

The reference cannot be built or run in this environment (C++23 + vcpkg dependencies, no tests, LFS data: SURVEY.md F7),
so these are NOT reference outputs: they freeze the oracle's answers on small seeded cases at the commit at which the
oracle passed its closed-form / brute-force / KAT pins (tests/test_oracle.py, tests/test_kat.py), so that later changes to
the shared headers cannot move both the oracle and the HIP path silently.  Each .npz holds data only: the developed image
(f32), the raw film sums and the event counters; traversal fixtures hold the query arrays and the expected answers."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

CASES = {
    "furnace_r16": dict(scene="furnace", res=16, spp=4, seed=11, kw={}),
    "furnace_fsd_r16": dict(scene="furnace", res=16, spp=4, seed=12, kw={"fsd": 1, "lut": (128, 128)}),
    "white_furnace_r12": dict(scene="white_furnace", res=12, spp=8, seed=13, kw={}),
    "double_slits_r96": dict(scene="double_slits", res=96, spp=8, seed=14, kw={"lut": (128, 128)}),
    # central 24x24 crop of the 1440x1440 film: the same pixel pitch (beam footprints) as the headline workload
    "cornell_box_r12": dict(scene="cornell_box", res=24, spp=2, seed=15, kw={"mesh_detail": 0, "lut": (128, 128), "crop_of": 1440}),
    # plt_path (forward: point transmitter + UTD diffraction, coverage sensor; backward: NEE + MIS), directional emitter, Stokes film
    "etoile_r48": dict(scene="etoile", res=48, spp=8, seed=16, kw={"mesh_detail": 0}),
    "white_furnace_path_r12": dict(scene="white_furnace_path", res=12, spp=8, seed=17, kw={}),
    "sunlit_r16": dict(scene="sunlit", res=16, spp=4, seed=18, kw={}),
    "cornell_box_stokes_r12": dict(scene="cornell_box", res=12, spp=2, seed=19, kw={"mesh_detail": 0, "lut": (128, 128), "crop_of": 1440, "polarimetric": 1}),
}
COUNTER_KEYS = ["segments", "vertices", "connections", "surface_interactions", "fsd_interactions", "light_splats"]


def make_scene(case):
    from wave_tracer_amd import Scene
    return Scene(case["scene"], res=case["res"], **case["kw"])


def run_case(case, renderer=None):
    """renderer(scene, begin, end, seed) -> (value, weight, light, counters); default: the oracle."""
    from wave_tracer_amd import develop
    from oracle_util import oracle_render
    sc = make_scene(case)
    v, w, l, c = (renderer or oracle_render)(sc, 0, case["spp"], case["seed"])
    img = develop(sc, v, w, l, case["spp"]).astype(np.float64)
    return img, {k: int(c[k]) for k in COUNTER_KEYS}


def traversal_queries():
    from test_oracle import random_rays, random_cones
    rays = random_rays(512, 21, -.02, .02)
    rays[:, 1] += .01
    cones = random_cones(256, 22, -.015, .015)
    cones[:, 1] += .01
    return rays, cones


def main():
    from test_oracle import oracle_trace, oracle_cones
    from wave_tracer_amd import Scene
    for name, case in CASES.items():
        img, counters = run_case(case)
        meta = {"case": {k: (v if k != "kw" else {a: list(b) if isinstance(b, tuple) else b for a, b in v.items()}) for k, v in case.items()},
                "counters": counters, "source": "oracle (parity unpinned vs reference)"}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img.astype(np.float32), meta=json.dumps(meta))
        print(name, img.shape, float(img.mean()), counters)
    sc = Scene("cornell_box", res=16, mesh_detail=0, lut=(32, 32))
    rays, cones = traversal_queries()
    dist, tuid, bary, front = oracle_trace(sc, rays)
    cdist, cflags, cntris, ctris = oracle_cones(sc, cones)
    np.savez_compressed(os.path.join(HERE, "cornell_traversal.npz"), rays=rays, dist=dist, tuid=tuid, bary=bary, front=front, cones=cones,
                        cdist=cdist, cflags=cflags, cntris=cntris, ctris=ctris)
    print("cornell_traversal", len(rays), len(cones))


if __name__ == "__main__":
    main()
