"""N>1 path on CPU: world_size-2 (and 3) gloo jobs through wave_tracer_amd.render.render_distributed.  Samples are sharded
by sample index, films are additive, so the reduced film must equal the single-process render of the whole range."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle_util import oracle_render

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_samples_partition():
    from wave_tracer_amd.render import shard_samples
    for spp in (1, 2, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            r = [shard_samples(spp, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == spp
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))           # contiguous, disjoint, complete
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1                                     # balanced


@pytest.mark.parametrize("world,spp,name", [(2, 6, "furnace"), (3, 5, "furnace"), (2, 8, "etoile")])
def test_gloo_sharded_render_equals_single_process(built, tmp_path, world, spp, name):
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "_dist_worker.py"), out, str(spp), "17", name]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    shards = d["shards"]
    assert shards[0][0] == 0 and shards[-1][1] == spp
    from wave_tracer_amd import Scene
    sc = Scene(name, res=16, lut=(32, 32), mesh_detail=0)   # (etoile: plt_path forward — sample sharding holds for every integrator)
    v, w, l, _ = oracle_render(sc, 0, spp, 17, threads=1)
    assert v.sum() + l.sum() > 0
    assert np.allclose(d["value"], v, rtol=1e-12) and np.allclose(d["weight"], w, rtol=1e-12) and np.allclose(d["light"], l, rtol=1e-12)


def test_gloo_progressive_render_reduces_partial_films(built, tmp_path):
    """render_distributed_progressive (the preview's multi-GPU path, §8f N4): three ranks render 7 samples in chunks of 2; after every chunk
    rank 0 holds the SUM of the ranks' partial films — a film of exactly the samples rendered so far (their number is reported, the weights
    of a furnace film grow in proportion) — and the last one is the finished film."""
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", WT_DIST_PROGRESSIVE="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "_dist_worker.py"), out, "7", "17", "furnace"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    from wave_tracer_amd import Scene
    sc = Scene("furnace", res=16, lut=(32, 32), mesh_detail=0)
    v, w, l, _ = oracle_render(sc, 0, 7, 17, threads=1)
    assert np.allclose(d["value"], v, rtol=1e-12) and np.allclose(d["weight"], w, rtol=1e-12) and np.allclose(d["light"], l, rtol=1e-12)
    part = d["partial"]
    # shards of 7 samples over 3 ranks: 2 + 2 + 3 -> chunks of 2: after chunk one 6 samples, after chunk two all 7
    assert [int(n) for n in part[:, 0]] == [6, 7]
    assert abs(part[0, 2] / part[1, 2] - 6 / 7) < 2e-2 and abs(part[1, 2] - w.sum()) < 1e-9 * w.sum()
    assert abs(part[1, 1] - (v.sum() + l.sum())) < 1e-9 * (v.sum() + l.sum())


def test_distributed_default_renderer_fails_loudly_without_gpu(built):
    """The product shard renderer is the HIP path; without an uploaded scene / GPU it must raise, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from wave_tracer_amd import Scene
    from wave_tracer_amd.api import WtgpuError
    from wave_tracer_amd.render import _render_shard_hip
    sc = Scene("furnace", res=8, lut=(32, 32))
    with pytest.raises(RuntimeError):
        _render_shard_hip(sc, 0, 1, 1)
    with pytest.raises(WtgpuError):
        sc.upload(0)
