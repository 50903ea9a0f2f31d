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
