"""Host-side render loop: the counterpart of scene_renderer_t::render / render_context_t::develop
(src/scene/render.cpp:381-579, 245-291).  Allocates the three linear film accumulators as torch tensors on the
GPU, calls the HIP path through the C-ABI and develops the film.  Multi-GPU: samples are sharded by sample index
(every rank renders all pixels for a disjoint sample range into its own film) and the films are summed with one
RCCL reduce — no collective on the data path."""
import ctypes as C

import numpy as np

from .api import Scene, load_library, _check


def alloc_films(scene, device):
    import torch
    H, W, Cn = scene.height, scene.width, scene.channels
    value = torch.zeros((H, W, Cn), dtype=torch.float64, device=device)
    weight = torch.zeros((H, W), dtype=torch.float64, device=device)
    light = torch.zeros((H, W, Cn), dtype=torch.float64, device=device)
    return value, weight, light


def develop(scene, value, weight, light, spe):
    """pixel = value/weight (0 where weight==0) + light/spe  — film_storage.hpp:256-287."""
    v = np.ascontiguousarray(value, dtype=np.float64)
    w = np.ascontiguousarray(weight, dtype=np.float64)
    l = np.ascontiguousarray(light, dtype=np.float64)
    out = np.zeros(v.shape, dtype=np.float32)
    _check(load_library().wtgpu_develop(scene.handle, v.ctypes.data, w.ctypes.data, l.ctypes.data, int(spe), out.ctypes.data))
    return out


def render(scene, spp, seed=1, device=0, sample_begin=0):
    """Renders `spp` samples per element on one GPU; returns (value, weight, light) as numpy f64 arrays."""
    import torch
    if scene.device is None:
        scene.upload(device)
    dev = torch.device("cuda", scene.device)
    with torch.cuda.device(dev):
        value, weight, light = alloc_films(scene, dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        scene.render_into(value, weight, light, sample_begin, sample_begin + spp, seed, stream)
        torch.cuda.synchronize(dev)
    return value.cpu().numpy(), weight.cpu().numpy(), light.cpu().numpy()


def shard_samples(spp, rank, world):
    """Sample-index sharding: rank r renders [r*spp/world, (r+1)*spp/world)."""
    b = (spp * rank) // world
    e = (spp * (rank + 1)) // world
    return b, e


def _render_shard_hip(scene, begin, end, seed):
    """Default shard renderer: the HIP path on the rank's GPU (fails loudly without one: there is no CPU fallback)."""
    import torch
    if scene.device is None:
        raise RuntimeError("render_distributed: scene.upload(local_rank) first")
    dev = torch.device("cuda", scene.device)
    value, weight, light = alloc_films(scene, dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    scene.render_into(value, weight, light, begin, end, seed, stream)
    return value, weight, light


def make_film_comm(device):
    """The C-ABI's RCCL communicator (wtgpu_comm_*: what a C++ host of the library uses) for the ranks of an initialised
    torch.distributed job: rank 0 creates the 128-byte id, torch.distributed carries it to the others (any transport would do)."""
    import torch.distributed as dist
    from .api import Comm
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return Comm(world, rank, device, box[0])


def render_distributed(scene, spp, seed=1, reduce_dst=0, shard_renderer=None, comm=None):
    """One process per GPU (torch.distributed already initialised; backend nccl = RCCL on GPUs).  Each rank renders its
    sample shard of every pixel into its own film; the films are summed onto `reduce_dst`: with `comm` (make_film_comm) by the
    C-ABI's wtgpu_film_reduce (one group of three ncclReduce), otherwise by torch.distributed (GPU films go through the host when the
    backend is gloo: ranks that share a GPU, CPU tests).
    `shard_renderer(scene, begin, end, seed) -> (value, weight, light)` torch tensors is a seam for the CPU (gloo) tests of
    the sharding / reduction logic; the product path is the default (HIP)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    b, e = shard_samples(spp, rank, world)
    value, weight, light = (shard_renderer or _render_shard_hip)(scene, b, e, seed)
    if comm is not None:
        comm.film_reduce(value, weight, light, root=reduce_dst, stream=torch.cuda.current_stream(value.device).cuda_stream)
    else:
        via_host = value.is_cuda and dist.get_backend() == "gloo"
        if via_host:
            torch.cuda.synchronize(value.device)
            value, weight, light = value.cpu(), weight.cpu(), light.cpu()
        for t in (value, weight, light):
            dist.reduce(t, dst=reduce_dst, op=dist.ReduceOp.SUM)
    if value.is_cuda:
        torch.cuda.synchronize(value.device)
    if rank == reduce_dst:
        return value.cpu().numpy(), weight.cpu().numpy(), light.cpu().numpy()
    return None


def render_distributed_progressive(scene, spp, seed=1, reduce_dst=0, chunk_spp=8, on_partial=None, shard_renderer=None, comm=None):
    """render_distributed in chunks, with a PARTIAL film on `reduce_dst` after every chunk — what the reference's preview shows while a render
    runs (src/scene/render.cpp:306-368: the intermediate result is the merge of every worker's image), across ranks.  Every rank renders its
    shard [b, e) of the samples in chunks of `chunk_spp`; after each chunk the ranks reduce COPIES of their accumulators (the accumulators
    themselves stay per-rank: samples keep adding up locally, only the last reduce is the final film) and `on_partial(value, weight, light,
    samples_per_element_so_far)` is called on `reduce_dst` with the summed numpy films — e.g. to develop and push them to a TevPreview.  The
    chunk boundaries are the same on every rank (shards differ by at most one sample: ranks that run out render empty chunks), so the reduces
    match up without any other coordination.  Returns the final films on `reduce_dst` (None elsewhere)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    b, e = shard_samples(spp, rank, world)
    longest = max(shard_samples(spp, r, world)[1] - shard_samples(spp, r, world)[0] for r in range(world))
    acc = None
    done_here = 0
    result = None
    for c0 in range(0, max(longest, 1), max(1, chunk_spp)):
        cb, ce = min(e, b + c0), min(e, b + c0 + max(1, chunk_spp))
        if ce > cb or acc is None:
            part = (shard_renderer or _render_shard_hip)(scene, cb, ce, seed)     # (an empty range renders nothing and returns zero films)
            acc = part if acc is None else tuple(a.add_(p_) for a, p_ in zip(acc, part))
            done_here += ce - cb
        copies = tuple(t.clone() for t in acc)
        if comm is not None:
            comm.film_reduce(*copies, root=reduce_dst, stream=torch.cuda.current_stream(copies[0].device).cuda_stream)
        else:
            if copies[0].is_cuda and dist.get_backend() == "gloo":
                torch.cuda.synchronize(copies[0].device)
                copies = tuple(t.cpu() for t in copies)
            for t in copies:
                dist.reduce(t, dst=reduce_dst, op=dist.ReduceOp.SUM)
        if copies[0].is_cuda:
            torch.cuda.synchronize(copies[0].device)
        counts = torch.tensor([done_here], dtype=torch.int64)
        if dist.get_backend() == "nccl":
            counts = counts.to(copies[0].device)
        dist.reduce(counts, dst=reduce_dst, op=dist.ReduceOp.SUM)
        if rank == reduce_dst:
            result = tuple(t.cpu().numpy() for t in copies)
            if on_partial is not None:
                on_partial(*result, int(counts.item()))
    return result if rank == reduce_dst else None
