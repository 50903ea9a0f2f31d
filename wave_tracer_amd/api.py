"""ctypes binding of the C-ABI in include/wtgpu.h (libwtgpu.so).  No compute happens in Python."""
import ctypes as C
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

# The renderer pipelines batches over several HIP streams.  The ROCm runtime maps streams onto GPU_MAX_HW_QUEUES (default 4)
# hardware queues and streams sharing a queue serialise (measured: 4.8 vs 3.5 Msamples/s).  The variable is read when
# libamdhip64 is loaded, i.e. it only takes effect if this package is imported BEFORE torch (bench.py sets it itself).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# ... and stages by-value kernel arguments in a 1 MiB ring per stream: a batch enqueues ~1000 launches of ~1 KB, a full ring blocks the
# enqueueing thread until the GPU catches up, and the internal streams serialise behind it (measured: enqueue 209 ms -> 16 ms per pass).
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))


def lib_path():
    # WTGPU_LIB: an alternative build of the library (A/B experiments with compile-time knobs)
    return os.environ.get("WTGPU_LIB") or os.path.join(_HERE, "libwtgpu.so")


class WtgpuError(RuntimeError):
    pass


class SceneParams(C.Structure):
    _fields_ = [("res", C.c_uint32), ("max_depth", C.c_int32), ("fsd", C.c_int32), ("mis", C.c_int32), ("rr", C.c_int32),
                ("force_ray_tracing", C.c_int32), ("mesh_detail", C.c_int32), ("lut_n_theta", C.c_uint32), ("lut_m", C.c_uint32),
                ("polarimetric", C.c_int32)]


class TestHooks(C.Structure):      # wave_tracer_amd/csrc/wtgpu_test_hooks.h (not part of the public C-ABI)
    _fields_ = [("only_s", C.c_uint32), ("only_t", C.c_uint32), ("crop_of", C.c_uint32)]


class SceneInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32), ("n_tris", C.c_uint32), ("n_edges", C.c_uint32),
                ("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("n_shapes", C.c_uint32), ("n_emitters", C.c_uint32),
                ("n_materials", C.c_uint32), ("max_depth", C.c_int32), ("sensor_type", C.c_uint32), ("fsd_lut_power", C.c_double * 2),
                ("bytes_per_sample_state", C.c_uint64), ("stokes", C.c_uint32), ("integrator", C.c_uint32)]


COUNTER_FIELDS = ["samples", "segments", "ray_queries", "cone_queries", "vertices", "connections", "shadow_rays", "cone_tri_overflow",
                  "edge_overflow", "fsd_edge_overflow", "fsd_pool_overflow", "fsd_interactions", "null_interactions",
                  "surface_interactions", "light_splats", "walk_iteration_cap_hits", "traversal_stack_dropped"]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}


# every symbol include/wtgpu.h declares
SYMBOLS = ["wtgpu_scene_create_named", "wtgpu_scene_create_from_desc", "wtgpu_scene_get_info", "wtgpu_scene_host_desc",
           "wtgpu_scene_upload", "wtgpu_render", "wtgpu_trace_rays", "wtgpu_traverse_cones", "wtgpu_get_counters",
           "wtgpu_reset_counters", "wtgpu_last_render_timings", "wtgpu_develop", "wtgpu_scene_destroy", "wtgpu_last_error",
           "wtgpu_scene_stats_json", "wtgpu_calibrate_copy", "wtgpu_render_async", "wtgpu_join", "wtgpu_query_regions", "wtgpu_render_progressive",
           "wtgpu_cancel", "wtgpu_pause", "wtgpu_resume", "wtgpu_capture_intermediate", "wtgpu_comm_unique_id", "wtgpu_comm_create", "wtgpu_film_reduce", "wtgpu_comm_destroy", "wtgpu_scene_create_from_xml"]
PROGRESS_CB = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_uint64, C.c_void_p)
CAPTURE_CB = C.CFUNCTYPE(None, C.c_uint64, C.c_void_p)

_lib = None


def load_library():
    """Loads libwtgpu.so; raises loudly if the HIP extension has not been built (there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise WtgpuError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         f"(hipcc --offload-arch=gfx950); wave_tracer_amd has no CPU fallback")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process leave the second one without
    # devices.  Importing torch first makes the dynamic loader bind libwtgpu.so to the runtime torch already loaded, so
    # that torch tensors / streams and our kernels share one HIP context.
    # The renderer pipelines batches over several HIP streams; the ROCm runtime maps streams onto 4 hardware queues by default
    # and streams sharing a queue serialise.  Must be set before the HIP runtime initialises (its first API call).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    lib.wtgpu_scene_create_named.argtypes = [C.c_char_p, C.POINTER(SceneParams), C.POINTER(vp)]
    lib.wtgpu_scene_create_named_hooks.argtypes = [C.c_char_p, C.POINTER(SceneParams), C.POINTER(TestHooks), C.POINTER(vp)]
    lib.wtgpu_scene_create_from_desc.argtypes = [vp, C.POINTER(vp)]
    lib.wtgpu_scene_create_from_xml.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), u32, C.POINTER(SceneParams), C.POINTER(vp)]
    lib.wtgpu_scene_compare.argtypes = [vp, vp, C.c_char_p, C.c_size_t]
    lib.wtgpu_scene_compare_part.argtypes = [vp, vp, C.c_char_p, C.c_char_p, C.c_size_t]
    lib.wtgpu_trace_ab_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    lib.wtgpu_render_progressive.argtypes = [vp, vp, vp, vp, vp, u64, u64, u64, u32, PROGRESS_CB, vp, C.POINTER(u64)]
    lib.wtgpu_cancel.argtypes = [vp]
    lib.wtgpu_pause.argtypes = [vp]
    lib.wtgpu_resume.argtypes = [vp]
    lib.wtgpu_capture_intermediate.argtypes = [vp, CAPTURE_CB, vp]
    lib.wtgpu_comm_unique_id.argtypes = [vp]
    lib.wtgpu_comm_create.argtypes = [i32, i32, i32, vp, C.POINTER(vp)]
    lib.wtgpu_film_reduce.argtypes = [vp, vp, vp, vp, vp, u64, u64, i32]
    lib.wtgpu_comm_destroy.argtypes = [vp]
    lib.wtgpu_comm_destroy.restype = None
    lib.wtgpu_scene_get_info.argtypes = [vp, C.POINTER(SceneInfo)]
    lib.wtgpu_scene_host_desc.argtypes = [vp]
    lib.wtgpu_scene_host_desc.restype = vp
    lib.wtgpu_scene_upload.argtypes = [vp, i32, u64]
    lib.wtgpu_render.argtypes = [vp, vp, vp, vp, vp, u64, u64, u64]
    lib.wtgpu_render_async.argtypes = [vp, vp, vp, vp, vp, u64, u64, u64]
    lib.wtgpu_join.argtypes = [vp, vp]
    lib.wtgpu_trace_rays.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp]
    lib.wtgpu_traverse_cones.argtypes = [vp, vp, vp, u32, u32, vp, vp, vp, vp]
    lib.wtgpu_query_regions.argtypes = [vp, vp, vp, u32, u32, vp, vp, vp, vp, vp, vp, vp]
    lib.wtgpu_calibrate_copy.argtypes = [u64, i32]
    lib.wtgpu_get_counters.argtypes = [vp, C.POINTER(Counters)]
    lib.wtgpu_reset_counters.argtypes = [vp]
    lib.wtgpu_last_render_timings.argtypes = [vp, C.POINTER(C.c_float * 12)]
    lib.wtgpu_develop.argtypes = [vp, vp, vp, vp, u64, vp]
    lib.wtgpu_scene_destroy.argtypes = [vp]
    lib.wtgpu_scene_destroy.restype = None
    lib.wtgpu_last_error.restype = C.c_char_p
    lib.wtgpu_scene_stats_json.argtypes = [vp]
    lib.wtgpu_scene_stats_json.restype = C.c_char_p
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise WtgpuError(f"wtgpu error {rc}: {load_library().wtgpu_last_error().decode(errors='replace')}")


class Scene:
    """Handle of a flattened scene (host-baked; optionally uploaded to one GPU)."""

    def __init__(self, name, res=256, max_depth=-1, fsd=-1, mis=-1, rr=-1, force_ray_tracing=0, mesh_detail=1, lut=(0, 0), only_s=None, only_t=None, crop_of=0,
                 polarimetric=0):
        lib = load_library()
        p = SceneParams(res, max_depth, fsd, mis, rr, force_ray_tracing, mesh_detail, lut[0], lut[1], polarimetric)
        h = C.c_void_p()
        if only_s is None and only_t is None and not crop_of:
            _check(lib.wtgpu_scene_create_named(name.encode(), C.byref(p), C.byref(h)))
        else:       # test hooks (single-strategy renders, crops of a larger film)
            hooks = TestHooks(0 if only_s is None else only_s + 1, 0 if only_t is None else only_t + 1, crop_of)
            _check(lib.wtgpu_scene_create_named_hooks(name.encode(), C.byref(p), C.byref(hooks), C.byref(h)))
        self._h = h
        self.name = name
        info = SceneInfo()
        _check(lib.wtgpu_scene_get_info(h, C.byref(info)))
        self.info = info
        # `channels` = film planes per pixel: spectral channels x Stokes components (1, or 4 for polarimetric sensors)
        self.spectral_channels, self.stokes = info.channels, info.stokes
        self.width, self.height, self.channels = info.width, info.height, info.channels * info.stokes
        self.device = None

    @classmethod
    def from_desc(cls, desc_ptr, keepalive=None, name="<desc>"):
        """Wraps an already flattened scene (pointer to a host `wt::scene_t`, wtgpu_scene_create_from_desc): the entry point a port of
        the reference's own loader would use.  `keepalive`: the object owning the host arrays (they must outlive the handle)."""
        lib = load_library()
        self = cls.__new__(cls)
        h = C.c_void_p()
        _check(lib.wtgpu_scene_create_from_desc(C.c_void_p(desc_ptr), C.byref(h)))
        self._h = h
        self._keepalive = keepalive
        self.name = name
        info = SceneInfo()
        _check(lib.wtgpu_scene_get_info(h, C.byref(info)))
        self.info = info
        self.spectral_channels, self.stokes = info.channels, info.stokes
        self.width, self.height, self.channels = info.width, info.height, info.channels * info.stokes
        self.device = None
        return self

    @classmethod
    def from_xml(cls, path, defines=None, res=0, max_depth=-1, fsd=-1, mis=-1, rr=-1, force_ray_tracing=0, lut=(0, 0), polarimetric=0, mesh_detail=1):
        """Loads a scene file of the reference's XML format with the minimal reader (wtgpu_scene_create_from_xml).  `defines`: dict of
        the reference's -D command-line defines.  mesh_detail: tessellation of the procedural stand-ins that replace Git-LFS pointer
        files (0: low-poly, for the CPU checker)."""
        lib = load_library()
        self = cls.__new__(cls)
        p = SceneParams(res, max_depth, fsd, mis, rr, force_ray_tracing, mesh_detail, lut[0], lut[1], polarimetric)
        d = [f"{k}={v}".encode() for k, v in (defines or {}).items()]
        arr = (C.c_char_p * max(1, len(d)))(*d)
        h = C.c_void_p()
        _check(lib.wtgpu_scene_create_from_xml(str(path).encode(), arr, len(d), C.byref(p), C.byref(h)))
        self._h = h
        self.name = os.path.basename(str(path))
        info = SceneInfo()
        _check(lib.wtgpu_scene_get_info(h, C.byref(info)))
        self.info = info
        self.spectral_channels, self.stokes = info.channels, info.stokes
        self.width, self.height, self.channels = info.width, info.height, info.channels * info.stokes
        self.device = None
        return self

    def first_difference(self, other, part=None):
        """Test hook (wtgpu_scene_compare[_part]): '' when the two flattened scenes are identical byte for byte, else the first differing
        array.  part: "sensor" | "opts" | "emitters" compares only that record."""
        buf = C.create_string_buffer(256)
        if part:
            rc = load_library().wtgpu_scene_compare_part(self._h, other._h, part.encode(), buf, 256)
        else:
            rc = load_library().wtgpu_scene_compare(self._h, other._h, buf, 256)
        if rc not in (0, 1):
            _check(rc)
        return buf.value.decode()

    @property
    def handle(self):
        return self._h

    def host_desc(self):
        return load_library().wtgpu_scene_host_desc(self._h)

    def stats(self):
        return json.loads(load_library().wtgpu_scene_stats_json(self._h).decode())

    def emitter_summary(self):
        """The scene's emitters in selection order: [{type, cutoff_deg, shape, select_pmf}]."""
        return self.stats()["emitter_list"]

    def upload(self, device=0, max_batch_samples=0):
        _check(load_library().wtgpu_scene_upload(self._h, int(device), int(max_batch_samples)))
        self.device = int(device)
        return self

    def render_into(self, value, weight, light, sample_begin, sample_end, seed, stream=None):
        """value/weight/light: CUDA(HIP) float64 torch tensors [H,W,C], [H,W], [H,W,C] (accumulated into)."""
        sp = C.c_void_p(stream) if stream else None
        _check(load_library().wtgpu_render(self._h, sp, value.data_ptr(), weight.data_ptr(), light.data_ptr(), int(sample_begin),
                                           int(sample_end), int(seed)))

    def render_async_into(self, value, weight, light, sample_begin, sample_end, seed, stream=None):
        """Like render_into, but `stream` does not wait for the work: consecutive calls pipeline on the GPU.  Call join(stream)
        before anything reads or overwrites the films."""
        sp = C.c_void_p(stream) if stream else None
        _check(load_library().wtgpu_render_async(self._h, sp, value.data_ptr(), weight.data_ptr(), light.data_ptr(), int(sample_begin),
                                                 int(sample_end), int(seed)))

    def render_progressive(self, value, weight, light, sample_begin, sample_end, seed, chunk_spp=1, progress=None, stream=None):
        """Blocking render with the reference's control surface: progress(samples_done, samples_total) -> truthy to stop; cancel() from
        any thread.  Returns (cancelled, samples_per_element_done)."""
        sp = C.c_void_p(stream) if stream else None
        cb = PROGRESS_CB((lambda d, t, u: 1 if progress(d, t) else 0) if progress else (lambda d, t, u: 0))
        done = C.c_uint64(0)
        lib = load_library()
        rc = lib.wtgpu_render_progressive(self._h, sp, value.data_ptr(), weight.data_ptr(), light.data_ptr(), int(sample_begin), int(sample_end),
                                          int(seed), int(chunk_spp), cb, None, C.byref(done))
        if rc == 6:      # WTGPU_CANCELLED
            return True, int(done.value)
        _check(rc)
        return False, int(done.value)

    def cancel(self):
        _check(load_library().wtgpu_cancel(self._h))

    def pause(self):
        """The running render_progressive stops launching at its next chunk boundary until resume() (scene_renderer_t's pause interrupt)."""
        _check(load_library().wtgpu_pause(self._h))

    def resume(self):
        _check(load_library().wtgpu_resume(self._h))

    def capture_intermediate(self, fn):
        """`capture intermediate`: fn(samples_per_element_done) is called once by the render thread at its next chunk boundary, with the films
        consistent (exactly the completed chunks).  Thread-safe."""
        # every trampoline stays alive until it has been CALLED: a request that replaces a pending one may arrive after the render thread has
        # already copied the old pointer
        keep = self.__dict__.setdefault("_capture_keepalive", [])
        slot = []

        def tramp(done, user):
            try:
                fn(int(done))
            finally:
                if slot and slot[0] in keep:
                    keep.remove(slot[0])
        cb = CAPTURE_CB(tramp)
        slot.append(cb)
        keep.append(cb)
        if len(keep) > 64:   # (requests that were replaced before they ran are never called: bound the list, oldest first)
            del keep[0]
        _check(load_library().wtgpu_capture_intermediate(self._h, cb, None))

    def join(self, stream=None):
        _check(load_library().wtgpu_join(self._h, C.c_void_p(stream) if stream else None))

    def trace_rays(self, rays):
        """rays: [n,8] f32 {o, d, tmin, tmax} (numpy) -> (dist, tuid, bary, front) numpy; device arrays are torch tensors."""
        import numpy as np
        import torch
        dev = torch.device("cuda", self.device)
        n = len(rays)
        d_rays = torch.from_numpy(np.ascontiguousarray(rays, dtype=np.float32)).to(dev)
        dist = torch.zeros(n, dtype=torch.float32, device=dev)
        tuid = torch.zeros(n, dtype=torch.int32, device=dev)
        bary = torch.zeros((n, 2), dtype=torch.float32, device=dev)
        front = torch.zeros(n, dtype=torch.int32, device=dev)
        _check(load_library().wtgpu_trace_rays(self._h, None, d_rays.data_ptr(), n, dist.data_ptr(), tuid.data_ptr(), bary.data_ptr(), front.data_ptr()))
        torch.cuda.synchronize(dev)
        return dist.cpu().numpy(), tuid.cpu().numpy().view(np.uint32), bary.cpu().numpy(), front.cpu().numpy().view(np.uint32)

    def traverse_cones(self, cones, cap=64):
        """cones: [n,10] f32 {o, d, tan_alpha, x0, ecc, lambda_m} -> (dist, flags, ntris, tris[n,cap] sorted)."""
        import numpy as np
        import torch
        dev = torch.device("cuda", self.device)
        n = len(cones)
        d_cones = torch.from_numpy(np.ascontiguousarray(cones, dtype=np.float32)).to(dev)
        dist = torch.zeros(n, dtype=torch.float32, device=dev)
        flags = torch.zeros(n, dtype=torch.int32, device=dev)
        ntris = torch.zeros(n, dtype=torch.int32, device=dev)
        tris = torch.zeros((n, cap), dtype=torch.int32, device=dev)
        _check(load_library().wtgpu_traverse_cones(self._h, None, d_cones.data_ptr(), n, cap, dist.data_ptr(), flags.data_ptr(), ntris.data_ptr(),
                                                   tris.data_ptr()))
        torch.cuda.synchronize(dev)
        return (dist.cpu().numpy(), flags.cpu().numpy().view(np.uint32), ntris.cpu().numpy().view(np.uint32),
                tris.cpu().numpy().view(np.uint32))

    def query_regions(self, cones, edge_cap=96):
        """cones: [n,10] f32 -> dict of numpy arrays: dist, flags, primary, ntris, nedges, edges[n,edge_cap] (sorted, 0xFFFFFFFF padded), flux."""
        import numpy as np
        import torch
        dev = torch.device("cuda", self.device)
        n = len(cones)
        d_cones = torch.from_numpy(np.ascontiguousarray(cones, dtype=np.float32)).to(dev)
        dist = torch.zeros(n, dtype=torch.float32, device=dev)
        flux = torch.zeros(n, dtype=torch.float32, device=dev)
        flags, primary, ntris, nedges = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4))
        edges = torch.full((n, edge_cap), -1, dtype=torch.int32, device=dev)
        _check(load_library().wtgpu_query_regions(self._h, None, d_cones.data_ptr(), n, edge_cap, dist.data_ptr(), flags.data_ptr(), primary.data_ptr(),
                                                  ntris.data_ptr(), nedges.data_ptr(), edges.data_ptr(), flux.data_ptr()))
        torch.cuda.synchronize(dev)
        u = lambda t: t.cpu().numpy().view(np.uint32)
        return {"dist": dist.cpu().numpy(), "flags": u(flags), "primary": u(primary), "ntris": u(ntris), "nedges": u(nedges), "edges": u(edges),
                "flux": flux.cpu().numpy()}

    def counters(self):
        c = Counters()
        _check(load_library().wtgpu_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        _check(load_library().wtgpu_reset_counters(self._h))

    def trace_ab_stats(self):
        """Test hook (WTGPU_TRACE_AB=n at upload: the first n rounds of every batch replay their trace queue through both per-lane trace kernels)."""
        a, b, d, w, r = C.c_double(0), C.c_double(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(load_library().wtgpu_trace_ab_stats(self._h, C.byref(a), C.byref(b), C.byref(d), C.byref(w), C.byref(r)))
        return {"ms_refill": a.value, "ms_sm": b.value, "differing_words": int(d.value), "walks": int(w.value), "rounds": int(r.value)}

    def timings(self):
        t = (C.c_float * 12)()
        _check(load_library().wtgpu_last_render_timings(self._h, C.byref(t)))
        return {"generate_ms": t[0], "trace_ms": t[1], "interact_ms": t[2], "connect_ms": t[3], "rounds": int(t[4]),
                "trace_launches": int(t[5]), "batches": int(t[6]), "trace_heavy_ms": t[7], "interact_b_ms": t[8], "flux_ms": t[9],
                "interact_c_ms": t[10], "rounds_per_batch": float(t[11])}

    def close(self):
        if self._h:
            load_library().wtgpu_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator of the C-ABI (wtgpu_comm_*): one process per GPU, films summed onto a root rank."""

    def __init__(self, world, rank, device, unique_id):
        h = C.c_void_p()
        self._id = C.create_string_buffer(bytes(unique_id), 128)
        _check(load_library().wtgpu_comm_create(int(world), int(rank), int(device), self._id, C.byref(h)))
        self._h = h

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(load_library().wtgpu_comm_unique_id(buf))
        return buf.raw

    def film_reduce(self, value, weight, light, root=0, stream=None):
        sp = C.c_void_p(stream) if stream else None
        _check(load_library().wtgpu_film_reduce(self._h, sp, value.data_ptr(), weight.data_ptr(), light.data_ptr(), value.numel(), weight.numel(), int(root)))

    def close(self):
        if self._h:
            load_library().wtgpu_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
