"""Worker of tests/test_gpu_render.py::test_two_ranks_one_gpu_sharded_render: one rank of a world_size-2 gloo job whose ranks share the
box's single GPU and render their sample shards with the PRODUCT renderer (HIP, through the C-ABI); the films are summed over the host
(gloo).  With one GPU per rank the same code path takes backend nccl / the C-ABI communicator (wave_tracer_amd.render.make_film_comm)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))   # the runtime's kernel-argument ring per stream (default 1 MiB: a full ring blocks the enqueueing thread)


def main():
    out, spp, seed, name = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    from wave_tracer_amd import Scene
    from wave_tracer_amd.render import render_distributed
    sc = Scene(name, res=24, lut=(32, 32), mesh_detail=0)
    sc.upload(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    res = render_distributed(sc, spp, seed=seed, reduce_dst=0)
    if dist.get_rank() == 0:
        np.savez(out, value=res[0], weight=res[1], light=res[2], counters=np.array([sc.counters()["samples"]]))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
