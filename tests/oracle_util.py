"""Loader for the CPU checker (oracle/_build/liboracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def load_oracle():
    global _lib
    if _lib is None:
        p = os.environ.get("WT_ORACLE_LIB") or os.path.join(ROOT, "oracle", "_build", "liboracle.so")   # (override: A/B builds of the checker)
        if not os.path.exists(p):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        lib = C.CDLL(p)
        lib.oracle_render.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.oracle_render_tiles.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_uint32, C.c_uint32, C.c_void_p]
        lib.oracle_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_traverse_cones.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


ORACLE_COUNTERS = ["segments", "ray_queries", "cone_queries", "vertices", "connections", "shadow_rays", "cone_tri_overflow", "edge_overflow",
                   "fsd_edge_overflow", "fsd_pool_overflow", "fsd_interactions", "null_interactions", "surface_interactions", "light_splats"]


def oracle_render(scene, sample_begin, sample_end, seed, threads=0):
    lib = load_oracle()
    H, W, Cn = scene.height, scene.width, scene.channels
    value = np.zeros((H, W, Cn), np.float64)
    weight = np.zeros((H, W), np.float64)
    light = np.zeros((H, W, Cn), np.float64)
    ctr = np.zeros(lib.oracle_counters_count(), np.uint64)
    rc = lib.oracle_render(scene.host_desc(), sample_begin, sample_end, seed, value.ctypes.data, weight.ctypes.data, light.ctypes.data, threads,
                           ctr.ctypes.data)
    assert rc == 0
    return value, weight, light, dict(zip(ORACLE_COUNTERS, [int(x) for x in ctr]))


def oracle_render_tiles(scene, sample_begin, sample_end, seed, tile_stride, tile_offset=0, threads=0):
    """Renders only the 24x24 blocks with index % tile_stride == tile_offset (a bounded sample of a full-size workload).
    Returns (value, weight, light, counters, n_samples, tile_mask[H,W])."""
    lib = load_oracle()
    H, W, Cn = scene.height, scene.width, scene.channels
    value = np.zeros((H, W, Cn), np.float64)
    weight = np.zeros((H, W), np.float64)
    light = np.zeros((H, W, Cn), np.float64)
    ctr = np.zeros(lib.oracle_counters_count(), np.uint64)
    n = C.c_uint64(0)
    rc = lib.oracle_render_tiles(scene.host_desc(), sample_begin, sample_end, seed, value.ctypes.data, weight.ctypes.data, light.ctypes.data, threads,
                                 ctr.ctypes.data, tile_stride, tile_offset, C.byref(n))
    assert rc == 0
    B = 24
    bx = (W + B - 1) // B
    ys, xs = np.mgrid[0:H, 0:W]
    blk = (ys // B) * bx + xs // B
    mask = (blk % tile_stride) == tile_offset if tile_stride > 1 else np.ones((H, W), bool)
    return value, weight, light, dict(zip(ORACLE_COUNTERS, [int(x) for x in ctr])), int(n.value), mask


def paired_bias_stats(G, C, rng=None, n_boot=2000):
    """G, C: [chunks, H, W] film sums of the same samples.  Returns a dict of the statistics the test asserts."""
    rng = rng or np.random.default_rng(1)
    d = G - C
    div = np.abs(d) > 0.5 * np.maximum(G, C)            # cells that hold a sample on a different discrete path
    n_div, n_pos = int(div.sum()), int((d[div] > 0).sum())
    tot = C.sum()
    bias_all = d.sum() / tot
    bias_trim = d[~div].sum() / C[~div].sum()
    # bootstrap over cells (the paired differences are i.i.d. across (chunk, pixel) cells to a good approximation)
    flat_d, flat_c = d.ravel(), C.ravel()
    idx = rng.integers(0, flat_d.size, size=(n_boot, flat_d.size // 8))   # (1/8 subsamples, rescaled: keeps memory bounded)
    boot = flat_d[idx].sum(axis=1) / flat_c[idx].sum(axis=1)
    se_all = float(boot.std() / np.sqrt(8.0))
    keep = ~div.ravel()
    kd, kc = flat_d[keep], flat_c[keep]
    idx = rng.integers(0, kd.size, size=(n_boot, kd.size // 8))
    boot_t = kd[idx].sum(axis=1) / kc[idx].sum(axis=1)
    se_trim = float(boot_t.std() / np.sqrt(8.0))
    # two-sided binomial p-value of the sign split of the divergent cells
    from math import comb
    k = min(n_pos, n_div - n_pos)
    p_sign = min(1.0, 2.0 * sum(comb(n_div, i) for i in range(k + 1)) / 2.0 ** n_div) if n_div else 1.0
    return dict(bias_all=float(bias_all), se_all=se_all, bias_trim=float(bias_trim), se_trim=se_trim, n_div=n_div, n_pos=n_pos, frac_div=float(div.mean()),
                p_sign=float(p_sign), rel_l1_trim=float(np.abs(d[~div]).sum() / C[~div].sum()), div_share_of_flux=float(np.maximum(G, C)[div].sum() / tot))
