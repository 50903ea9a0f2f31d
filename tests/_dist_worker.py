"""Worker of tests/test_distributed.py: one rank of a world_size-N gloo job exercising the sample-sharding + film
reduction host logic (wave_tracer_amd.render.render_distributed) with the CPU checker standing in for the GPU renderer."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out, spp, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    from wave_tracer_amd import Scene
    from wave_tracer_amd.render import render_distributed, shard_samples
    from oracle_util import oracle_render
    name = sys.argv[4] if len(sys.argv) > 4 else "furnace"
    sc = Scene(name, res=16, lut=(32, 32), mesh_detail=0)
    rank, world = dist.get_rank(), dist.get_world_size()

    def cpu_shard(scene, b, e, s):
        v, w, l, _ = oracle_render(scene, b, e, s, threads=1)
        return torch.from_numpy(v), torch.from_numpy(w), torch.from_numpy(l)

    partial = []
    if os.environ.get("WT_DIST_PROGRESSIVE"):   # chunked render with a partial film on rank 0 after every chunk (the preview's multi-GPU path)
        from wave_tracer_amd.render import render_distributed_progressive
        res = render_distributed_progressive(sc, spp, seed=seed, reduce_dst=0, chunk_spp=int(os.environ["WT_DIST_PROGRESSIVE"]), shard_renderer=cpu_shard,
                                             on_partial=lambda v, w, l, n: partial.append((n, float(v.sum() + l.sum()), float(w.sum()))))
    else:
        res = render_distributed(sc, spp, seed=seed, reduce_dst=0, shard_renderer=cpu_shard)
    shards = [None] * world
    dist.all_gather_object(shards, shard_samples(spp, rank, world))
    if rank == 0:
        np.savez(out, value=res[0], weight=res[1], light=res[2], shards=np.array(shards), partial=np.array(partial, dtype=np.float64).reshape(-1, 3))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
