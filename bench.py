#!/usr/bin/env python3
"""bench.py — throughput of the MI355X-native plt_bdpt hot path on BASELINE.json's headline workload.

A *step* = one pass of the hot path over one batch of synthetic input = 1 sample per pixel of the cornell-box
(stand-in) scene at 1440x1440, visible-spectrum wave mode (BDPT, max_depth 16, MIS, RR, Fraunhofer FSD): 2,073,600
samples.  Inputs (flattened scene, BVH, LUTs) are resident in HBM before the timed region; film buffers are torch
tensors on the GPU.  Multi-GPU: samples are sharded by sample index across ranks (weak scaling by default: every rank renders
`steps` passes; --scaling strong: the ranks split the --spp-per-step samples of every step), no collective on the data path; one RCCL
reduce of the film afterwards (outside the timed region it would be <2 ms; it is included in the timed region for honesty).
`python bench.py --gpus N` without a launcher starts its N ranks itself (torch.distributed.run, 127.0.0.1).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

# The renderer pipelines batches over 3 HIP streams (plus the caller's); the ROCm runtime reads this when libamdhip64 is loaded (import torch), its
# default of 4 hardware queues makes streams share queues and serialise (see wave_tracer_amd/api.py).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))   # the runtime's kernel-argument ring per stream (default 1 MiB: a full ring blocks the enqueueing thread)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cpu_baseline(scene_name, res, seconds_target=15.0, mesh_detail=1, polarimetric=0):
    """CPU checker (oracle/, kind 'port') timed on the host cores on a bounded sample of the SAME workload: every n-th
    24x24 block of the same full-size film (same pixel pitch => same beam footprints and per-sample work), for about
    `seconds_target` seconds.  Only rank 0 at N=1 runs this."""
    from wave_tracer_amd.api import Scene
    from oracle_util import oracle_render_tiles
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    sc = Scene(scene_name, res=res, mesh_detail=mesh_detail, polarimetric=polarimetric)
    n_tiles = ((sc.width + 23) // 24) * ((sc.height + 23) // 24)
    # work unit of the checker = one 24x24 block x one sample index: >= 8 blocks per core, spp work items per block
    stride = max(1, n_tiles // (8 * cores))
    from oracle_util import load_oracle
    load_oracle().oracle_set_fine_items(1)     # work items of one block x one sample index: no thread waits for a block of 576 x spp samples at the end
    t = time.time()
    _, _, _, _, n1, _ = oracle_render_tiles(sc, 0, 1, 123, stride, 0, threads=cores)      # calibration pass
    dt1 = max(1e-3, time.time() - t)
    spp = max(1, min(256, int(seconds_target / dt1)))
    t = time.time()
    _, _, _, _, n, _ = oracle_render_tiles(sc, 1, 1 + spp, 123, stride, 0, threads=cores)
    dt = time.time() - t
    import ctypes
    lib = load_oracle()
    lib.oracle_set_fine_items(0)
    lib.oracle_last_utilisation.restype = ctypes.c_double
    util = float(lib.oracle_last_utilisation())
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port", "thread_utilisation": util,
            "sample": f"{scene_name} res={res}: every {stride}th 24x24 block of the SAME {sc.width}x{sc.height} film, {spp} spp "
                      f"({n} samples, {dt:.1f}s), {cores} threads, work items of one block x one sample index (mean busy / longest busy thread "
                      f"{util:.2f}); scalar fp32 restatement (oracle/), baseline only"}


def measure_traffic(kernel, scene_args, timeout_s=240):
    """HBM-side bytes per launch of `kernel` measured NOW, on this box: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, with
    --kernel-trace only) over one step of the same workload in a child process, summed over the kernel's dispatches of that step and scaled to bytes with the
    newest calibration committed under profiles/ (a streaming copy of known size with this code's access width, tools/profile_round.sh: the counters
    read KiB; FETCH_SIZE under-reports by 2 on gfx950, MI355X_MICROARCH.md).  Returns (bytes_per_step, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not found"
    # counter -> bytes factors: the newest committed calibration (profiles/rNN_pmc_traffic.json["calibration"], written by
    # tools/make_traffic_json.py from the k_calib_copy passes of tools/profile_round.sh); without one, the guide's value for FETCH_SIZE (x2) and 1
    kf, kw, cal_src = 2.0, 1.0, "MI355X_MICROARCH.md (FETCH_SIZE x 2), uncalibrated"
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        try:
            with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")) as f:
                cal = json.load(f)["calibration"]
            kf, kw, cal_src = float(cal["fetch_factor"]), float(cal["write_factor"]), f"profiles/{tag}_pmc_traffic.json (k_calib_copy: a streaming copy of known size)"
            break
        except (OSError, KeyError, ValueError):
            continue
    tot = {}
    n_disp = 0
    for counter, k in (("FETCH_SIZE", kf), ("WRITE_SIZE", kw)):
        d = tempfile.mkdtemp(prefix="wtgpu_pmc_")
        try:
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-traffic"] + scene_args
            r = subprocess.run(cmd, cwd=d, env=dict(os.environ, TMPDIR=d), capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            # (`kernel` may be a tuple: a HIP-event bracket that spans several kernels — their bytes are summed, per launch of the bracket =
            # per dispatch of its most frequent kernel)
            names = kernel if isinstance(kernel, (tuple, list)) else (kernel,)
            s, ids = 0.0, {n: set() for n in names}
            for fn in files:
                with open(fn) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] != counter:
                            continue
                        for n in names:
                            if n in row["Kernel_Name"] and (n + "_") not in row["Kernel_Name"]:
                                s += float(row["Counter_Value"])
                                ids[n].add((fn, row["Dispatch_Id"]))
            tot[counter] = s * 1024.0 * k
            n_disp = max(n_disp, max(len(v) for v in ids.values()))
        except (subprocess.TimeoutExpired, OSError) as e:
            return None, f"rocprofv3 --pmc {counter}: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if not n_disp:
        return None, "kernel not found in the counter collection"
    # bytes of ONE step (the caller divides by the launches per step of its own timed region: the child's first batch launches its rounds without
    # the history the library's expectation of rounds-with-work builds on, i.e. more empty ones — same bytes, other dispatch count)
    return tot["FETCH_SIZE"] + tot["WRITE_SIZE"], {"fetch_bytes": tot["FETCH_SIZE"], "write_bytes": tot["WRITE_SIZE"], "dispatches": n_disp, "fetch_factor": kf, "write_factor": kw,
                                                   "factors_from": cal_src}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # passes pipeline on the GPU (a pass is in flight for ~0.25 s): with few steps the ramp-up and drain of that pipeline weigh in
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="cornell_box")
    ap.add_argument("--res", type=int, default=1440)
    ap.add_argument("--batch", type=int, default=0, help="samples per kernel batch (0: one full step)")
    ap.add_argument("--batch-target", type=int, default=0, help="default step size: as many samples per element as bring one step (= one batch) to about "
                    "this many samples (every batch runs its rounds down to a thin tail, so the tails cost per BATCH: DESIGN.md §0).  0: 4.2 M for plt_bdpt "
                    "scenes (three slices of that are 186 GB), 9 M for plt_path scenes (no vertex stores: 2 KB of state per sample)")
    ap.add_argument("--mesh-detail", type=int, default=-1, help="stand-in geometry level (default: 1 for the cornell box = SURVEY 8(d) C1/C3's 283 K triangles; 2 for "
                    "etoile / bidir_room = C4's ~560 seeded buildings, C5's ~50 objects and 34 materials)")
    ap.add_argument("--polarimetric", type=int, default=-1, help="Stokes film (default: on for bidir_room = BASELINE.json configs[4])")
    ap.add_argument("--ray-tracing", action="store_true", help="diagnostic: --ray-tracing of the reference CLI (wt_context.hpp:43), no cones / diffraction")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the live PMC measurement of roofline.traffic (two rocprofv3 passes over one step, ~20 s)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="film reduce: nccl = RCCL over xGMI (one GPU per rank); gloo: host reduce, ranks may share a GPU (launcher tests on 1-GPU boxes)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--spp-per-step", type=int, default=0, help="samples per element and step (default 1; strong scaling: the number of ranks)")
    ap.add_argument("--film-sums", action="store_true", help="add the sums of the (reduced) film planes to the JSON line: with --warmup 0 the ranks of a strong-scaling run "
                    "render exactly the samples a single rank renders with the same --spp-per-step, so the sums must agree (tests/test_gpu_render.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: become the launcher (one rank per GPU over RCCL, rendezvous on 127.0.0.1)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    import torch
    import torch.distributed as dist
    from wave_tracer_amd import Scene
    from wave_tracer_amd.render import alloc_films

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":
            local_rank %= max(1, torch.cuda.device_count())
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, f"--gpus {args.gpus} but the launcher started {dist.get_world_size()} ranks"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    comm = None
    if distributed and args.backend == "nccl":
        # the film reduce of the timed region goes through the C-ABI (wtgpu_comm_* / wtgpu_film_reduce: RCCL inside the library, what a
        # C++ host would call); torch.distributed only carries the communicator id and the barriers
        from wave_tracer_amd.render import make_film_comm
        comm = make_film_comm(local_rank)
    md = args.mesh_detail if args.mesh_detail >= 0 else (2 if args.scene in ("etoile", "bidir_room") else 1)
    pol = args.polarimetric if args.polarimetric >= 0 else (1 if args.scene == "bidir_room" else 0)
    sc = Scene(args.scene, res=args.res, mesh_detail=md, polarimetric=pol, force_ray_tracing=1 if args.ray_tracing else 0)
    npix = sc.width * sc.height
    value, weight, light = None, None, None
    K, Wm = args.steps, args.warmup
    # one step = S samples per element over the whole job.  weak: every rank renders its own S per step (work per GPU fixed); strong:
    # the ranks split the S samples of a step (total work fixed).  Sample indices are disjoint across ranks and steps.
    # default: whole passes that fill one batch (round 4: cornell 1440^2 -> 2 spp per step, etoile 720x540 -> 23); strong scaling: a multiple of the ranks
    batch_target = args.batch_target or (9000000 if int(sc.info.integrator) != 0 else 4200000)   # (measured, run r4z: etoile 720x540 at 4.3 / 6.2 / 9.3 / 12.4 M
    # samples per batch 59.1 / 62.8 / 66.9 / 62.9 Msamples/s, 1440x1080 at 4.7 / 7.8 / 12.4 M 60.7 / 64.6 / 62.6; cornell at 6.2 M 24.7 vs 25.7 at 4.15 M)
    S_auto = max(1, min(32, round(batch_target / npix)))
    S = args.spp_per_step or (world * max(1, S_auto // world) if args.scaling == "strong" else S_auto)
    if args.scaling == "strong":
        assert S % world == 0, "--spp-per-step must be a multiple of the number of ranks for strong scaling"
    s_rank = S // world if args.scaling == "strong" else S          # samples per element this rank renders per step
    base = rank * (K + Wm) * s_rank
    sc.upload(local_rank, args.batch or npix * s_rank)              # one step = one batch (the library shrinks it to what the free HBM holds)
    value, weight, light = alloc_films(sc, dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def sync():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for s in range(Wm):
        sc.render_into(value, weight, light, base + s * s_rank, base + (s + 1) * s_rank, 1, stream)
    sc.reset_counters()
    for t in (value, weight, light):
        t.zero_()
    sync()
    t0 = time.time()
    # wtgpu_render_async only enqueues and does not make `stream` wait: consecutive steps (more samples into the same film
    # accumulators) pipeline on the GPU; the join + closing sync below complete all K steps inside the timed region
    for s in range(K):
        sc.render_async_into(value, weight, light, base + (Wm + s) * s_rank, base + (Wm + s + 1) * s_rank, 1, stream)
    sc.join(stream)
    if distributed:
        if comm is not None:
            comm.film_reduce(value, weight, light, root=0, stream=stream)
        else:
            for t in (value, weight, light):
                torch.cuda.synchronize(dev)
                h = t.cpu()
                dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
                t.copy_(h)
    sync()
    dt = time.time() - t0
    dt_rank = dt
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # self-diagnosing multi-GPU line: every rank's own wall time of the K steps (before the max), gathered on rank 0
    per_rank_s = [dt_rank]
    if distributed:
        gl = [torch.zeros(1, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu") for _ in range(world)]
        dist.all_gather(gl, torch.tensor([dt_rank], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu"))
        per_rank_s = [float(x.item()) for x in gl]
    film_sums = [float(t.sum().item()) for t in (value, weight, light)] if args.film_sums else None   # (rank 0 holds the reduced film)
    counters = sc.counters()
    tsum = sc.timings()      # HIP-event kernel times accumulated over the timed region (reset after the warm-up)
    if rank == 0:
        samples_total = npix * K * s_rank * world
        msps = samples_total / dt / 1e6
        # ---- roofline of the dominant kernel.  Algorithmic bytes per sample from SURVEY.md §8(d):
        #   B = N_seg*2*S_path + N_vtx*S_vtx + N_conn*2*S_vtx + N_q*S_hit + B_film   with the measured per-sample counts.
        ns = max(1, counters["samples"])
        n_seg = counters["segments"] / ns
        n_vtx = counters["vertices"] / ns
        n_conn = counters["connections"] / ns
        n_q = (counters["ray_queries"] + counters["cone_queries"] + counters["shadow_rays"]) / ns
        n_light = counters["light_splats"] / ns
        C = sc.channels   # film planes per pixel (spectral channels x Stokes components)
        S_path = 200.0    # mean of backward (224 B) and forward (176 B) walk records
        S_vtx, S_hit = 320.0, 32.0
        b_film = 2 * (9 * C * 16) + n_light * 2 * (9 * C * 8)
        bytes_per_sample = n_seg * 2 * S_path + n_vtx * S_vtx + n_conn * 2 * S_vtx + n_q * S_hit + b_film
        kernels = {"k_trace": tsum["trace_ms"], "k_trace_heavy": tsum["trace_heavy_ms"], "k_interact": tsum["interact_ms"],
                   "k_edges+k_interact_b": tsum["interact_b_ms"], "k_flux_split+k_flux_tasks": tsum["flux_ms"], "k_interact_c": tsum["interact_c_ms"],
                   "k_connect": tsum["connect_ms"], "k_generate": tsum["generate_ms"]}
        path_mode = int(sc.info.integrator) != 0
        PATH_BRACKET = "k_path_fsd+interact+edges+interact_b+nee"
        if path_mode:      # plt_path scenes: ONE bracket spans the five kernels between k_trace_heavy and the next round (the later brackets are empty)
            kernels = {(PATH_BRACKET if k == "k_interact" else k): v for k, v in kernels.items()}
        dom = max(kernels, key=kernels.get)
        # bytes attributed to the dominant kernel per step (one step = npix samples); the two trace kernels split the segments
        # (every segment is traced by exactly one of them), the two interaction passes split the vertices the same way: each is
        # credited with the WHOLE term (an upper bound of its algorithmic bytes, hence of `achieved`)
        share = {"k_trace": n_seg * S_path + n_q * S_hit, "k_trace_heavy": n_seg * S_path + n_q * S_hit, "k_interact": n_seg * S_path + n_vtx * S_vtx,
                 PATH_BRACKET: n_seg * 2 * S_path,
                 "k_edges+k_interact_b": n_seg * S_path + n_vtx * S_vtx, "k_flux_split+k_flux_tasks": n_seg * S_path + n_vtx * S_vtx,
                 "k_interact_c": n_seg * S_path + n_vtx * S_vtx, "k_connect": n_conn * 2 * S_vtx + b_film, "k_generate": 2 * S_path + 2 * S_vtx}[dom]
        # A batch launches its round kernels for the rounds its walks are expected to need (+ a margin; wtgpu.hip: batch_launcher_t) — until round 4
        # all kMaxWalkIters = 96 of them: `launches` is the count rocprofv3 --kernel-trace --stats averages over (profiles/rNN_kernel_stats*.csv),
        # `launches_with_work` the rounds that had walks queued.  achieved = algorithmic bytes / the kernel's HIP-event time: the same for either count.
        rounds = int(round(tsum["rounds_per_batch"] * tsum["batches"]))
        launches = {"k_connect": tsum["batches"], "k_generate": tsum["batches"]}.get(dom, rounds)
        with_work = tsum["trace_launches"] if launches == rounds else launches
        avg_ms = kernels[dom] / max(1, launches)
        steps_rank = K * s_rank                                   # passes over the film this rank rendered
        alg_bytes_per_launch = share * npix * steps_rank / max(1, launches)
        achieved = alg_bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # whole path: all algorithmic bytes of a step over the wall time of a step
        whole = bytes_per_sample * npix * s_rank / (dt / K) / 1e9
        # HBM traffic of that kernel, per launch like `achieved`: measured live (measure_traffic: two rocprofv3 --pmc passes over one step of this
        # workload, on this box, in child processes after the timed region); if rocprofv3 is unavailable, the summary committed under profiles/
        traffic, traffic_src = None, None
        integ, max_depth, n_tris_scene, emitters = int(sc.info.integrator), int(sc.info.max_depth), int(sc.info.n_tris), sc.emitter_summary()
        if world == 1 and not args.no_traffic:
            # the child processes size their batches from the FREE device memory (wtgpu_scene_upload): give this process's state and films back
            # first, or the child splits the step into smaller batches than the timed region ran and `traffic` is per a different launch
            del value, weight, light
            sc.close()
            torch.cuda.empty_cache()
            # (the brackets of the material-sorted pass A and of the staged connections span several kernels: their bytes are summed)
            staged = os.environ.get("WTGPU_STAGED_CONNECT", "0") != "0"
            sorted_a = os.environ.get("WTGPU_SORTED_INTERACT", "0") != "0"
            kname = {"k_trace": ("k_tr_axis", "k_tr_cone", "k_tr_policy", "k_tr_tail", "k_trace_refill") if os.environ.get("WTGPU_TRACE_STAGED", "1") != "0" else "k_trace_refill", "k_connect": ("k_connect_eval", "k_connect_shadow", "k_connect_mis") if staged else "k_connect_strat",
                     "k_interact": ("k_classify", "k_interact_diffuse", "k_interact_dielectric", "k_interact_spm", "k_interact_any") if sorted_a else "k_interact",
                     "k_edges+k_interact_b": "k_interact_b",
                     "k_flux_split+k_flux_tasks": "k_flux_tasks",
                     PATH_BRACKET: ("k_path_fsd", "k_path_interact", "k_path_edges", "k_path_interact_b", "k_path_nee")}.get(dom, dom)
            scene_args = ["--scene", args.scene, "--res", str(args.res), "--mesh-detail", str(md), "--polarimetric", str(pol)] + (["--ray-tracing"] if args.ray_tracing else [])
            traffic, detail = measure_traffic(kname, scene_args)
            if traffic is not None:
                traffic /= max(1.0, launches / K)   # per launch like `achieved`: the step's bytes over the launches per step of the timed region
            traffic_src = {"measured": "live, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over one step", "kernel": kname, **detail} if traffic is not None else {"measured": None, "reason": detail}
        if traffic is None:
            try:
                # (the live passes failed or were switched off: the newest committed measurement of this workload)
                names = {"etoile": ["r03_pmc_traffic_etoile.json"], "bidir_room": ["r03_pmc_traffic_bidir_room.json"]}.get(
                    args.scene, [f"r{r:02d}_pmc_traffic.json" for r in range(9, 3, -1)])
                name = next(n for n in names if os.path.exists(os.path.join(ROOT, "profiles", n)))
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    pt = json.load(f)
                kk = {"k_trace": "k_trace_refill"}.get(dom, dom)
                if kk in pt["kernels"] and pt["workload"]["res"] == args.res and pt["workload"]["scene"] == args.scene:
                    traffic = pt["kernels"][kk]["hbm_bytes_per_launch"]
                    traffic_src = dict(traffic_src or {}, fallback="profiles/" + os.path.basename(f.name))
            except (OSError, KeyError, ValueError, StopIteration):
                pass
        # the same fraction against the kernels' EXCLUSIVE time: the newest committed one-stream rocprofv3 statistics of this workload (profiles/
        # rNN_kernel_stats_streams1.csv: three 2-spp steps = 6 passes of the film), when the dominant bracket's kernels are in it — the HIP-event bracket
        # above includes the time a kernel shares the GPU with the other two streams' kernels, this one does not
        excl = None
        if (args.scene, args.res) == ("cornell_box", 1440):
            import csv
            import glob
            import re
            names = {"k_trace": ("k_tr_axis", "k_tr_cone", "k_tr_policy", "k_tr_tail", "k_trace_refill"), "k_interact": ("k_interact",), "k_trace_heavy": ("k_trace_heavy",),
                     "k_connect": ("k_connect_enum", "k_connect_scan", "k_connect_strat", "k_connect_splat_tiled")}.get(dom)
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_stats_streams1.csv")))
            if names and files:
                try:
                    ms = 0.0
                    with open(files[-1]) as fh:
                        for row in csv.DictReader(fh):
                            m = re.search(r"_ZN3wtk\d+(k_\w+?)E", row["kernel"])
                            if m and m.group(1) in names:
                                ms += float(row["total_ms"])
                    if ms > 0:
                        ms_pass = ms / 6.0
                        ach = share * npix / (ms_pass * 1e-3) / 1e9
                        excl = {"ms_per_pass": ms_pass, "achieved": ach, "frac": ach / 8000.0, "source": "profiles/" + os.path.basename(files[-1]), "kernels": list(names)}
                except (OSError, KeyError, ValueError):
                    excl = None
        out = {
            "metric": ("Msamples/sec (whole node), cornell-box 1440^2 wave-mode" if (args.scene, args.res) == ("cornell_box", 1440)
                       else f"Msamples/sec (whole node), {args.scene} res={args.res}"),
            "value": msps, "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} stand-in (the scene file's geometry, LFS meshes replaced by procedural stand-ins) res={args.res} "
                                   f"{['plt_bdpt', 'plt_path forward', 'plt_path backward'][integ]} max_depth={max_depth} "
                                   f"{'MIS RR Fraunhofer-FSD' if integ == 0 else 'UTD-FSD'}, {S} spp per step; interaction regions exact "
                                   f"(unbounded: regions beyond the 64-triangle fast path are walked in full, DESIGN.md §5)",
                       "samples_per_step": npix * S, "tris": n_tris_scene,
                       "emitter_selection": ", ".join(f"{e['type']} {e['select_pmf']:.4g}" for e in emitters),
                       "parallelism": f"sample-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": dom, "avg_launch_ms": avg_ms, "launches": launches, "launches_with_work": with_work,
                         "avg_launch_ms_with_work": kernels[dom] / max(1, with_work),
                         "alg_bytes_per_launch": alg_bytes_per_launch,
                         "alg_bytes_per_sample_all_kernels": bytes_per_sample,
                         "whole_path": {"alg_bytes_per_step": bytes_per_sample * npix * s_rank, "achieved": whole, "frac": whole / 8000.0},
                         "exclusive": excl,
                         # HIP-event brackets on the concurrent slice streams: each includes the time the kernel shares the GPU with the
                         # other streams' kernels, so the sum exceeds ms_per_step (exclusive times: profiles/r03_kernel_stats_streams1.csv)
                         "kernel_ms_per_step_stream_summed": {k: v / K for k, v in kernels.items()}},
            "counters_per_sample": {"segments": n_seg, "vertices": n_vtx, "connections": n_conn, "bvh_queries": n_q, "light_splats": n_light,
                                    # (not an overflow of anything: triangles of interaction regions beyond the 64-entry fast-path list, which the
                                    # whole-region walks of k_edges / k_flux_* cover — DESIGN.md §5; the C-ABI counter is still called cone_tri_overflow)
                                    "region_tris_beyond_fast_path_list": counters["cone_tri_overflow"] / ns, "fsd_interactions": counters["fsd_interactions"] / ns,
                                    "iteration_cap_hits": counters["walk_iteration_cap_hits"] / ns,
                                    "traversal_stack_dropped": counters["traversal_stack_dropped"] / ns},
        }
        if distributed:
            rccl = None
            try:
                v = torch.cuda.nccl.version()
                rccl = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
            except Exception:
                pass
            film_bytes = sum(int(t.numel()) * t.element_size() for t in (value, weight, light)) if value is not None else None
            out["film_reduce"] = {"through": "wtgpu_film_reduce (RCCL inside the C-ABI library: one group of three ncclReduce)" if comm is not None else "torch.distributed gloo (host)",
                                  "ranks": world, "rccl_version": rccl, "bytes_per_rank": film_bytes, "backend": args.backend}
            # per rank: wall seconds of the timed region and the rate of the samples that rank rendered (weak: K * S * npix each)
            out["per_rank"] = [{"rank": r, "seconds": per_rank_s[r], "msamples_per_s": npix * K * s_rank / per_rank_s[r] / 1e6} for r in range(world)]
        if film_sums is not None:
            out["film_sums"] = {"value": film_sums[0], "weight": film_sums[1], "light": film_sums[2]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.scene, args.res, args.cpu_seconds, md, pol)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
