// wave_tracer_amd — HIP (gfx950 / CDNA4) wavefront implementation of the plt_bdpt hot path + the C-ABI (include/wtgpu.h).
//
// Kernel pipeline of one batch of samples (DESIGN.md "Kernels"):
//   k_generate   one thread per sample : spectral/emitter/sensor samples, vertex 0 of both subpaths
//   repeat until the walk queue is empty (<= kMaxWalkIters rounds):
//     k_trace    one thread per queued walk : integrator::traverse (ballistic ray segments + cone queries) over the
//                8-wide BVH; traversal stack in LDS (lane-interleaved), spill to scratch
//     k_interact one thread per queued walk : surface / Fraunhofer-FSD / null interaction, vertex append, RR,
//                re-enqueue
//   k_connect    one thread per sample : all (s,t) connections, shadow rays, MIS, film splats (f64 atomics)
// Every round kernel is *persistent*: a fixed grid whose wavefronts grab 64 queue items at a time through a device-side head
// counter and read the queue length from a device control block, so the host never reads anything back: a whole batch
// (generate, kMaxWalkIters rounds, connect) is enqueued blindly, rounds after the queue ran empty cost a few us each.
// Batches are round-robined over several state slices, each with its own HIP stream, so that the long tails of one batch
// (a handful of slow walks) overlap with the bulk of the others; wtgpu_render is asynchronous w.r.t. the host.
// All per-walk / per-sample state lives in HBM as one contiguous record per walk (wt::soa_load/soa_store, record-major since round 3:
// after the first queue compaction the walks of a wavefront are scattered over the batch, and a lane that reads whole lines of its own
// record wastes nothing, whereas the word-interleaved layout of rounds 1-2 fetched a 64-byte line per word and lane).
//
// There is no CPU fallback in this file: every entry point that computes requires a HIP device.
#include <hip/hip_runtime.h>
#include <chrono>
#include <rccl/rccl.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "wtgpu_kernels.h"
#include "wtgpu_test_hooks.h"
#include "scene_abi_check.h"
#include "host/scene_builder.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_CHECK(x)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (x);                                                                                 \
        if (e_ != hipSuccess) return fail(WTGPU_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));    \
    } while (0)


}   // namespace

struct chunk_rec_t {
    std::vector<hipEvent_t> ev;   // [0] start, [1] after generate, then 6 per LAUNCHED round, last used: after connect
    uint32_t* h_ctl = nullptr;    // pinned snapshot of the slice's control block after the batch
    uint32_t* h_mid = nullptr;    // ... and after the rounds launched up front (render_first_part): is the queue empty?
    hipEvent_t ev_mid = nullptr;
    hipEvent_t ev_stagger = nullptr;   // recorded after the batch's round `stagger_round`: the next batch (on the next stream) starts there
    uint32_t rounds_launched = 0;
    uint32_t rounds_timed = 0;   // ... of which bracketed by timing events (6 per round, in launch order)
    size_t ev_used = 0;           // timing events recorded so far (the next one closes the batch)
    size_t ev_final = 0;          // index of the event recorded after the batch's last kernel
    bool busy = false;
};

struct wtgpu_scene {
    std::unique_ptr<wth::scene_builder_t> builder;   // owns the host arrays (named scenes)
    scene_t host{};                                  // host-pointer scene
    scene_t dev{};                                   // device-pointer scene
    std::vector<void*> dev_allocs;
    int device = -1;
    bool uploaded = false;
    std::vector<device_state_t> slices;              // per-batch path state, one slice per internal stream
    std::vector<const path_state_t*> d_path_slices;  // ... and its plt_path part (device copies)
    const unsigned char* d_tri_class = nullptr;       // walk class of every triangle (bdpt_ext_t::tri_class)
    uint32_t pend_cap = 0, n_chunks = 0;              // staged connections: items per chunk, chunks per batch (bdpt_ext_t)
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> ev_done;
    hipEvent_t ev_begin = nullptr;
    hipEvent_t ev_stagger_last = nullptr;   // the previous batch's stagger event (owned by its record)
    std::vector<chunk_rec_t> recs;                   // in-flight batch records (events + control block snapshot)
    size_t rec_next = 0;
    size_t slice_next = 0;   // batches go round-robin over the slices ACROSS render calls (a call with one batch does not always land on stream 0)
    // A batch is enqueued in two parts (render_first_part / render_finish_part): generation + the rounds its walks are EXPECTED to need, and —
    // once the host has seen that the round queue is empty (or has launched the remaining rounds) — the connections.  Between the two it is
    // `pending` on its slice; the next batch of that slice, wtgpu_join and everything that reads results finish it first.
    struct pending_t {
        bool active = false;
        unsigned char args[1024];   // the batch's launch block (launch_args_t, defined below)
        chunk_rec_t* rec = nullptr;
        uint32_t rounds_first = 0;
        uint32_t launched = 0, rounds_step = 8;   // rounds enqueued so far; rounds to add at the next look (finish_look)
    };
    std::vector<pending_t> pending;   // per slice
    uint32_t rounds_hist[8] = {0};    // rounds with work of the last batches seen (the expectation is their maximum + a margin)
    uint32_t rounds_hist_n = 0;
    uint64_t round_fallbacks = 0;     // batches whose queue was not empty after the first part (they got all kMaxWalkIters rounds)
    uint64_t rounds_launched_total = 0;
    bool timing = true;
    std::string stats;
    double lut_power[2] = {0, 0};
    double acc[12] = {0};                             // accumulated timings since the last reset (see wtgpu_last_render_timings)
    uint64_t samples_rendered = 0;
    uint64_t cap_hits = 0;
    std::atomic<int> cancel{0};
    std::atomic<int> paused{0};
    std::mutex capture_mutex;
    wtgpu_capture_cb capture_cb = nullptr;   // pending `capture intermediate` (under capture_mutex)
    void* capture_user = nullptr;
    uint32_t* query_scratch = nullptr;   // wtgpu_traverse_cones
    size_t query_scratch_bytes = 0;
    // WTGPU_TRACE_AB (diagnostic, tests/test_gpu_traversal.py): accumulated over the replayed rounds — milliseconds of k_trace_refill / k_trace_sm on the
    // same queue, words of their outputs that differ (traversal records + triangle lists + heavy-queue checksums), walks replayed
    double ab_ms[2] = {0, 0};
    uint64_t ab_mismatch = 0, ab_walks = 0, ab_rounds = 0;
    uint64_t light_rounds_run = 0;   // rounds k_light_rounds ran (diagnostic)
    // tuning knobs (environment, read ONCE at upload: wtgpu_scene_upload)
    struct knobs_t {
        uint32_t cone_budget = 0, count_stats = 1, profile = 0, no_lists = 0, stagger_round = 0, lane_cache = 1, heavy_cache = 1, split_queues = 1;
        uint32_t shrink_r1 = 8, shrink_f1 = 4, shrink_r2 = 16, shrink_f2 = 32, shrink_h1 = 4, decay_q = 0, decay_c = 4;   // persistent-grid sizes of the later rounds (see wtgpu_render_async)
        uint32_t heavy_waves_per_cu = 8, round_blocks_per_cu = 8, grid_div_b = 4, grid_div_c = 1, grid_div_hard = 4, grid_mul_flux = 2, coop_aperture_min = 8, heavy_probe = 1, flux_task_tris = kFluxTaskTris;
        uint32_t coop_io = 0, primary_axis = 0, sorted_interact = 0, staged_connect = 0, conn_pool = 16, grid_div_cls[4] = {1, 4, 2, 4};   // WTGPU_SORTED_INTERACT / WTGPU_STAGED_CONNECT = 0: the one-kernel forms (A/B); WTGPU_GRID_CLS=a,b,c,d: persistent grids of the class kernels relative to the round's
        uint32_t trace_staged = 1, trace_stages = 3, trace_staged_rounds = 4;   // ... for the first WTGPU_TRACE_STAGED_ROUNDS rounds of a batch (the long ones: a stage is a launch, and a short round is bound by its launches)   // WTGPU_TRACE_STAGED=1: the traversal in stages (k_tr_axis / k_tr_cone / k_tr_policy / k_tr_tail), WTGPU_TRACE_STAGES cone stages before the tail
        uint32_t trace_sm = 0, trace_ab = 0;   // WTGPU_TRACE_SM=1: the phase-machine trace kernel (k_trace_sm); WTGPU_TRACE_AB=n: the first n rounds replay their trace queue through both kernels (timed, outputs compared)
        uint32_t light_rounds = 1;   // WTGPU_LIGHT_ROUNDS=0: the rounds beyond the expected ones as ordinary rounds only (k_light_rounds off)
        uint32_t max_rounds = kWalkIterLimit;   // WTGPU_MAX_ROUNDS: rounds a batch may get before its surviving walks are dropped and counted (default: wt/bdpt.h kWalkIterLimit — nothing is dropped in any workload seen; 96 = rounds 1-5)
        uint32_t first_rounds = 0, rounds_margin = 2, tiled_splat = 1;   // WTGPU_TILED_SPLAT=0: the plain per-sample splat kernel   // WTGPU_FIRST_ROUNDS (0: adaptive), WTGPU_ROUNDS_MARGIN
        int dbg_stage = 1 << 30;
    } knobs;
};

struct wtgpu_comm {
    ncclComm_t comm = nullptr;
    int device = -1, world = 0, rank = 0;
};

// restores the calling thread's current device when an entry point returns
struct device_guard_t {
    int prev = -1;
    explicit device_guard_t(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~device_guard_t() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

namespace {

template <class T>
int upload(wtgpu_scene* s, const T* src, size_t n, const T** dst) {
    *dst = nullptr;
    if (n == 0 || !src) return WTGPU_OK;
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, n * sizeof(T)));
    s->dev_allocs.push_back(p);
    HIP_CHECK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return WTGPU_OK;
}
template <class T>
int dmalloc(wtgpu_scene* s, T** p, size_t n) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess) return fail(WTGPU_ERR_OOM, std::string("hipMalloc of ") + std::to_string(n * sizeof(T)) + " bytes: " + hipGetErrorString(e));
    s->dev_allocs.push_back(q);
    *p = static_cast<T*>(q);
    return WTGPU_OK;
}

}   // namespace

// ================================================ C-ABI ==============================================================
extern "C" {

const char* wtgpu_last_error(void) { return g_err.c_str(); }

// The HIP runtime stages by-value kernel arguments in a 1 MiB ring per stream; a batch enqueues ~1000 launches of ~1 KB, and a full ring blocks
// the enqueueing thread until the GPU has caught up — which serialises the internal streams.  Ask for 16 MiB before the runtime reads its
// settings (it does so at its first use; a process that initialised HIP earlier sets HSA_KERNARG_POOL_SIZE itself).
// GPU_MAX_HW_QUEUES likewise (default 4 hardware queues: the three internal streams and the caller's share queues and serialise).  Both are only
// defaults (a host's own setting wins) and both take effect only if the runtime has not initialised yet; g_env_by_host records what the host had
// set itself, wtgpu_scene_upload refuses settings that are KNOWN to serialise the streams (runtime_settings_ok).
static int g_env_by_host = 0;   // bit 0: HSA_KERNARG_POOL_SIZE, bit 1: GPU_MAX_HW_QUEUES were in the environment when the library was loaded
__attribute__((constructor)) static void wtgpu_runtime_settings() {
    if (getenv("HSA_KERNARG_POOL_SIZE")) g_env_by_host |= 1;
    if (getenv("GPU_MAX_HW_QUEUES")) g_env_by_host |= 2;
    setenv("HSA_KERNARG_POOL_SIZE", "16777216", 0);
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
}
// The two runtime settings the stream pipeline needs.  An explicit setting that is too small is an ERROR (it would silently cost
// 30-50 % — WTGPU_ALLOW_SLOW_RUNTIME=1 overrides); settings this library had to default itself are reported once: they are in effect only if HIP
// was not initialised before libwtgpu.so was loaded, which cannot be queried.
static int runtime_settings_ok(std::string& why) {
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    const char* k = getenv("HSA_KERNARG_POOL_SIZE");
    const long nq = q ? atol(q) : 4, ring = k ? atol(k) : (1l << 20);
    if (getenv("WTGPU_ALLOW_SLOW_RUNTIME")) return 1;
    if (nq < 4) {
        why = "GPU_MAX_HW_QUEUES=" + std::string(q ? q : "(unset)") + ": the renderer's three internal streams and the caller's need >= 4 hardware queues (8 recommended); "
              "export GPU_MAX_HW_QUEUES=8 before the HIP runtime initialises, or WTGPU_ALLOW_SLOW_RUNTIME=1 to run serialised";
        return 0;
    }
    if (ring < (4l << 20)) {
        why = "HSA_KERNARG_POOL_SIZE=" + std::string(k ? k : "(unset)") + ": a batch enqueues ~900 launches of ~1 KB of kernel arguments; with a ring below 4 MiB the enqueueing "
              "thread blocks and the streams serialise; export HSA_KERNARG_POOL_SIZE=16777216 before the HIP runtime initialises, or WTGPU_ALLOW_SLOW_RUNTIME=1";
        return 0;
    }
    if ((g_env_by_host & 3) != 3 && !getenv("WTGPU_QUIET")) {
        static bool told = false;
        if (!told) fprintf(stderr, "[wtgpu] note: %s%s defaulted by libwtgpu.so; effective only if the HIP runtime had not initialised before the library was loaded "
                           "(a host that uses HIP earlier exports them itself, INTEGRATION.md)\n", (g_env_by_host & 1) ? "" : "HSA_KERNARG_POOL_SIZE=16777216 ",
                           (g_env_by_host & 2) ? "" : "GPU_MAX_HW_QUEUES=8");
        told = true;
    }
    return 1;
}

int wtgpu_scene_create_named_hooks(const char* name, const wtgpu_scene_params* params, const wtgpu_test_hooks* hooks, wtgpu_scene** out) {
    if (!name || !params || !out) return fail(WTGPU_ERR_INVALID, "null argument");
    try {
        auto s = std::make_unique<wtgpu_scene>();
        s->builder = std::make_unique<wth::scene_builder_t>();
        wth::scene_params_t p{};
        p.res = params->res ? params->res : 256;
        p.max_depth = params->max_depth;
        p.fsd = params->fsd;
        p.mis = params->mis;
        p.rr = params->rr;
        p.force_ray_tracing = params->force_ray_tracing;
        p.mesh_detail = params->mesh_detail;
        p.lut_n_theta = params->lut_n_theta;
        p.lut_m = params->lut_m;
        p.debug_only_s = hooks ? hooks->only_s : 0u;
        p.debug_only_t = hooks ? hooks->only_t : 0u;
        p.crop_of = hooks ? hooks->crop_of : 0u;
        p.polarimetric = params->polarimetric;
        if (!wth::build_named_scene(name, p, *s->builder)) return fail(WTGPU_ERR_INVALID, std::string("unknown scene ") + name);
        s->host = s->builder->scene();
        s->stats = s->builder->stats();
        s->lut_power[0] = s->builder->fsd_lut_power(0);
        s->lut_power[1] = s->builder->fsd_lut_power(1);
        *out = s.release();
        return WTGPU_OK;
    } catch (const std::exception& e) {
        return fail(WTGPU_ERR_INVALID, e.what());
    }
}
int wtgpu_scene_create_named(const char* name, const wtgpu_scene_params* params, wtgpu_scene** out) {
    return wtgpu_scene_create_named_hooks(name, params, nullptr, out);
}

static void finish_built_scene(wtgpu_scene* s) {
    s->host = s->builder->scene();
    s->stats = s->builder->stats();
    s->lut_power[0] = s->builder->fsd_lut_power(0);
    s->lut_power[1] = s->builder->fsd_lut_power(1);
}
int wtgpu_scene_create_from_xml(const char* path, const char* const* defines, uint32_t n_defines, const wtgpu_scene_params* params, wtgpu_scene** out) {
    if (!path || !out || (n_defines && !defines)) return fail(WTGPU_ERR_INVALID, "null argument");
    try {
        auto s = std::make_unique<wtgpu_scene>();
        s->builder = std::make_unique<wth::scene_builder_t>();
        wth::scene_params_t p{};
        p.max_depth = p.fsd = p.mis = p.rr = -1;
        p.mesh_detail = 1;
        if (params) {
            p.res = params->res;
            p.max_depth = params->max_depth;
            p.fsd = params->fsd;
            p.mis = params->mis;
            p.rr = params->rr;
            p.force_ray_tracing = params->force_ray_tracing;
            p.lut_n_theta = params->lut_n_theta;
            p.lut_m = params->lut_m;
            p.mesh_detail = params->mesh_detail;
            p.polarimetric = params->polarimetric;
        }
        std::vector<std::string> defs;
        for (uint32_t i = 0; i < n_defines; ++i) {
            if (!defines[i]) return fail(WTGPU_ERR_INVALID, "null define");
            defs.emplace_back(defines[i]);
        }
        wth::build_scene_from_xml(path, defs, p, *s->builder);
        finish_built_scene(s.get());
        *out = s.release();
        return WTGPU_OK;
    } catch (const std::exception& e) {
        return fail(WTGPU_ERR_INVALID, e.what());
    }
}

// Test hook: field-by-field comparison of two flattened scenes (every array the description names, byte for byte).  0: identical;
// 1: different — `what` names the first difference.
// `only`: nullptr = everything, else one of "sensor", "opts", "emitters" (records that do not depend on the geometry: a scene file whose
// meshes are absent can still be checked for what else it describes)
static int scene_compare(const wtgpu_scene* a, const wtgpu_scene* b, const char* only, char* what, size_t n_what) {
    if (!a || !b) return fail(WTGPU_ERR_INVALID, "null scene");
    const scene_t &x = a->host, &y = b->host;
    std::string diff;
    const std::string part = only ? only : "";
    auto wanted = [&](const char* name) {
        if (part.empty()) return true;
        const std::string n(name);
        if (part == "sensor") return n == "sensor";
        if (part == "opts") return n == "opts";
        if (part == "emitters") return n == "n_emitters" || n == "emitters" || n == "emitter_cdf";
        return false;
    };
    auto cnt = [&](const char* name, uint64_t u, uint64_t v) {
        if (!wanted(name)) return;
        if (diff.empty() && u != v) diff = std::string(name) + ": " + std::to_string(u) + " vs " + std::to_string(v);
    };
    auto arr = [&](const char* name, const void* u, const void* v, size_t bytes, size_t elem) {
        if (!wanted(name)) return;
        if (!diff.empty() || bytes == 0) return;
        if (!u || !v) {
            if (u != v) diff = std::string(name) + ": missing array";
            return;
        }
        if (std::memcmp(u, v, bytes) != 0) {
            size_t k = 0;
            while (k < bytes && ((const unsigned char*)u)[k] == ((const unsigned char*)v)[k]) ++k;
            diff = std::string(name) + ": element " + std::to_string(k / elem) + ", byte " + std::to_string(k % elem);
        }
    };
    cnt("n_tris", x.n_tris, y.n_tris);
    cnt("n_edges", x.n_edges, y.n_edges);
    cnt("n_nodes", x.n_nodes, y.n_nodes);
    cnt("n_leaves", x.n_leaves, y.n_leaves);
    cnt("n_shapes", x.n_shapes, y.n_shapes);
    cnt("n_materials", x.n_materials, y.n_materials);
    cnt("n_spectra", x.n_spectra, y.n_spectra);
    cnt("n_emitters", x.n_emitters, y.n_emitters);
    cnt("n_textures", x.n_textures, y.n_textures);
    cnt("lut.n_theta", x.lut.n_theta, y.lut.n_theta);
    cnt("lut.m", x.lut.m, y.lut.m);
    arr("sensor", &x.sensor, &y.sensor, sizeof(sensor_t), sizeof(sensor_t));
    arr("opts", &x.opts, &y.opts, sizeof(integrator_opts_t), sizeof(integrator_opts_t));
    arr("world_min", &x.world_min, &y.world_min, sizeof(vec3), sizeof(vec3));
    arr("world_max", &x.world_max, &y.world_max, sizeof(vec3), sizeof(vec3));
    arr("tri_geo", x.tri_geo, y.tri_geo, sizeof(tri_geo_t) * x.n_tris, sizeof(tri_geo_t));
    arr("tri_meta", x.tri_meta, y.tri_meta, sizeof(tri_meta_t) * x.n_tris, sizeof(tri_meta_t));
    arr("tri_shade", x.tri_shade, y.tri_shade, sizeof(tri_shade_t) * x.n_tris, sizeof(tri_shade_t));
    arr("edges", x.edges, y.edges, sizeof(edge_t) * x.n_edges, sizeof(edge_t));
    arr("nodes", x.nodes, y.nodes, sizeof(bvh8_node_t) * x.n_nodes, sizeof(bvh8_node_t));
    arr("leaves", x.leaves, y.leaves, sizeof(bvh8_leaf_t) * x.n_leaves, sizeof(bvh8_leaf_t));
    arr("shapes", x.shapes, y.shapes, sizeof(shape_t) * x.n_shapes, sizeof(shape_t));
    arr("materials", x.materials, y.materials, sizeof(material_t) * x.n_materials, sizeof(material_t));
    arr("spectra", x.spectra, y.spectra, sizeof(spectrum_t) * x.n_spectra, sizeof(spectrum_t));
    arr("textures", x.textures, y.textures, sizeof(texture_t) * x.n_textures, sizeof(texture_t));
    arr("emitters", x.emitters, y.emitters, sizeof(emitter_t) * x.n_emitters, sizeof(emitter_t));
    arr("emitter_cdf", x.emitter_cdf, y.emitter_cdf, sizeof(float) * (x.n_emitters + 1), sizeof(float));
    if (diff.empty() && part.empty()) {
        size_t nsd = 0, nkd = 0, nk = 0;
        for (uint32_t i = 0; i < x.n_spectra; ++i)
            if (x.spectra[i].type != SPEC_CONST && x.spectra[i].type != SPEC_DISCRETE) nsd = std::max<size_t>(nsd, x.spectra[i].offset + (size_t)x.spectra[i].count * (x.spectra[i].is_complex ? 2 : 1));
        arr("spectra_data", x.spectra_data, y.spectra_data, sizeof(float) * nsd, sizeof(float));
        for (uint32_t i = 0; i < x.n_emitters; ++i) nk = std::max<size_t>(nk, (size_t)x.emitters[i].k_dist + 1);
        arr("kdists", x.kdists, y.kdists, sizeof(kdist_t) * nk, sizeof(kdist_t));
        for (size_t i = 0; i < nk && diff.empty(); ++i) nkd = std::max<size_t>(nkd, x.kdists[i].offset + 2 * (size_t)x.kdists[i].count);
        arr("kdist_data", x.kdist_data, y.kdist_data, sizeof(float) * nkd, sizeof(float));
        arr("lut.icdf_theta1", x.lut.icdf_theta1, y.lut.icdf_theta1, sizeof(float) * x.lut.n_theta, sizeof(float));
        arr("lut.icdf_theta2", x.lut.icdf_theta2, y.lut.icdf_theta2, sizeof(float) * x.lut.n_theta, sizeof(float));
        arr("lut.icdf1", x.lut.icdf1, y.lut.icdf1, sizeof(float) * (size_t)x.lut.m * x.lut.m, sizeof(float));
        arr("lut.icdf2", x.lut.icdf2, y.lut.icdf2, sizeof(float) * (size_t)x.lut.m * x.lut.m, sizeof(float));
    }
    if (what && n_what) {
        std::strncpy(what, diff.c_str(), n_what - 1);
        what[n_what - 1] = 0;
    }
    return diff.empty() ? 0 : 1;
}
int wtgpu_scene_compare(const wtgpu_scene* a, const wtgpu_scene* b, char* what, size_t n_what) { return scene_compare(a, b, nullptr, what, n_what); }
int wtgpu_trace_ab_stats(wtgpu_scene* s, double* ms_refill, double* ms_sm, uint64_t* differing_words, uint64_t* walks, uint64_t* rounds) {
    if (!s) return fail(WTGPU_ERR_INVALID, "null scene");
    if (ms_refill) *ms_refill = s->ab_ms[0];
    if (ms_sm) *ms_sm = s->ab_ms[1];
    if (differing_words) *differing_words = s->ab_mismatch;
    if (walks) *walks = s->ab_walks;
    if (rounds) *rounds = s->ab_rounds;
    return WTGPU_OK;
}
int wtgpu_scene_compare_part(const wtgpu_scene* a, const wtgpu_scene* b, const char* part, char* what, size_t n_what) {
    if (!part) return fail(WTGPU_ERR_INVALID, "null part");
    return scene_compare(a, b, part, what, n_what);
}

int wtgpu_scene_create_from_desc(const wtgpu_scene_desc* desc, wtgpu_scene** out) {
    if (!desc || !out) return fail(WTGPU_ERR_INVALID, "null argument");
    auto s = std::make_unique<wtgpu_scene>();
    std::memcpy(&s->host, desc, sizeof(scene_t));   // identical layouts: scene_abi_check.h
    if (s->host.n_tris > 0 && (!s->host.tri_geo || !s->host.tri_meta || !s->host.tri_shade || !s->host.nodes)) return fail(WTGPU_ERR_INVALID, "scene description lacks geometry arrays");
    s->stats = "{}";
    *out = s.release();
    return WTGPU_OK;
}

int wtgpu_scene_get_info(const wtgpu_scene* s, wtgpu_scene_info* info) {
    if (!s || !info) return fail(WTGPU_ERR_INVALID, "null argument");
    const scene_t& h = s->host;
    info->width = h.sensor.width;
    info->height = h.sensor.height;
    info->channels = h.sensor.channels;
    info->stokes = film_stokes(h.sensor);
    info->integrator = h.opts.integrator;
    info->n_tris = h.n_tris;
    info->n_edges = h.n_edges;
    info->n_nodes = h.n_nodes;
    info->n_leaves = h.n_leaves;
    info->n_shapes = h.n_shapes;
    info->n_emitters = h.n_emitters;
    info->n_materials = h.n_materials;
    info->max_depth = h.opts.max_depth;
    info->sensor_type = (uint32_t)h.sensor.type;
    info->fsd_lut_power[0] = s->lut_power[0];
    info->fsd_lut_power[1] = s->lut_power[1];
    const uint64_t mv = (uint64_t)h.opts.max_depth + 2;
    info->bytes_per_sample_state = 4ull * (2 * (kWalkWords + mv * kVertexWords + kTravWords + kMaxConeTris) + kCtxWords);
    return WTGPU_OK;
}

const wtgpu_scene_desc* wtgpu_scene_host_desc(const wtgpu_scene* s) { return s ? reinterpret_cast<const wtgpu_scene_desc*>(&s->host) : nullptr; }
const char* wtgpu_scene_stats_json(const wtgpu_scene* s) { return s ? s->stats.c_str() : "{}"; }

static int upload_impl(wtgpu_scene* s, int device, uint64_t max_batch);
static void release_device(wtgpu_scene* s);

int wtgpu_scene_upload(wtgpu_scene* s, int device, uint64_t max_batch) {
    if (!s) return fail(WTGPU_ERR_INVALID, "null scene");
    if (s->uploaded) return fail(WTGPU_ERR_INVALID, "scene already uploaded");
    {
        std::string why;
        if (!runtime_settings_ok(why)) return fail(WTGPU_ERR_INVALID, why);
    }
    int ndev = 0;
    const hipError_t dc = hipGetDeviceCount(&ndev);
    if (dc != hipSuccess || ndev == 0)
        return fail(WTGPU_ERR_NO_DEVICE, std::string("no HIP device present (there is no CPU fallback): hipGetDeviceCount -> ") + hipGetErrorString(dc) +
                                             ", count " + std::to_string(ndev));
    if (device < 0 || device >= ndev) return fail(WTGPU_ERR_NO_DEVICE, "invalid device index");
    device_guard_t guard(device);
    s->device = device;
    const int rc_up = upload_impl(s, device, max_batch);
    if (rc_up != WTGPU_OK) release_device(s);   // nothing half-uploaded stays behind: a retry starts from scratch
    return rc_up;
}

static void read_knobs(wtgpu_scene* s) {
    auto u = [](const char* name, uint32_t dflt) {
        const char* e = getenv(name);
        return e ? (uint32_t)std::max(0, atoi(e)) : dflt;
    };
    wtgpu_scene::knobs_t& k = s->knobs;
    k.cone_budget = u("WTGPU_CONE_BUDGET", kConeBudget);
    k.count_stats = u("WTGPU_COUNT_STATS", 1);
    k.split_queues = u("WTGPU_SPLIT_QUEUES", 1);
    k.shrink_r1 = u("WTGPU_SHRINK_R1", k.shrink_r1);
    k.first_rounds = std::min<uint32_t>(u("WTGPU_FIRST_ROUNDS", k.first_rounds), kMaxWalkIters);
    k.rounds_margin = u("WTGPU_ROUNDS_MARGIN", k.rounds_margin);
    k.light_rounds = u("WTGPU_LIGHT_ROUNDS", k.light_rounds);
    k.max_rounds = std::min<uint32_t>(kWalkIterLimit, std::max<uint32_t>(8u, u("WTGPU_MAX_ROUNDS", k.max_rounds)));
    k.tiled_splat = u("WTGPU_TILED_SPLAT", k.tiled_splat);
    k.shrink_f1 = std::max(1u, u("WTGPU_SHRINK_F1", k.shrink_f1));
    k.shrink_r2 = u("WTGPU_SHRINK_R2", k.shrink_r2);
    k.shrink_f2 = std::max(1u, u("WTGPU_SHRINK_F2", k.shrink_f2));
    k.shrink_h1 = std::max(1u, u("WTGPU_SHRINK_H1", k.shrink_h1));
    k.decay_q = u("WTGPU_DECAY_Q", k.decay_q);   // per cent; 0: the step schedule above
    k.decay_c = std::max(1u, u("WTGPU_DECAY_C", k.decay_c));
    k.lane_cache = u("WTGPU_LANE_CACHE", 1);
    k.heavy_cache = u("WTGPU_HEAVY_CACHE", 1);
    k.stagger_round = std::min<uint32_t>(u("WTGPU_STAGGER_ROUND", 0), kMaxWalkIters - 1);   // 0: all streams start at once
    k.profile = u("WTGPU_PROFILE", 0);
    k.no_lists = getenv("WTGPU_NO_LISTS") ? 1u : 0u;
    k.heavy_waves_per_cu = std::max(1u, u("WTGPU_HEAVY_WAVES", 8));   // swept 6 / 8 / 10 / 12 / 16 / 24 / 32: 169.6 / 168.0 / 171.6 / 174.3 / 176 / 181 / 183 ms per pass
    k.round_blocks_per_cu = std::max(1u, u("WTGPU_ROUND_BLOCKS", 8));
    k.grid_div_b = std::max(1u, u("WTGPU_GRID_B", 4));
    k.grid_div_c = std::max(1u, u("WTGPU_GRID_C", 1));   // (2 until round 4; 1: bidir_room 33.6 -> 33.9, cornell 25.15 -> 25.35 Msamples/s, pass C's bracket 69 -> 56 / 93 -> 65 ms)
    k.grid_div_hard = std::max(1u, u("WTGPU_GRID_HARD", 4));
    k.grid_mul_flux = std::max(1u, u("WTGPU_GRID_FLUX", 2));
    k.heavy_probe = u("WTGPU_HEAVY_PROBE", 1);
    k.trace_sm = u("WTGPU_TRACE_SM", k.trace_sm);
    k.trace_ab = u("WTGPU_TRACE_AB", 0);
    k.trace_staged = u("WTGPU_TRACE_STAGED", k.trace_staged);
    k.trace_staged_rounds = u("WTGPU_TRACE_STAGED_ROUNDS", k.trace_staged_rounds);
    k.trace_stages = std::min(16u, std::max(1u, u("WTGPU_TRACE_STAGES", k.trace_stages)));
    // Pass A and the connections each exist in two forms (DESIGN.md §4 has the measurements: the one-kernel forms are 1-6 % faster on the headline
    // workload and are the default; the sorted / staged forms move a third of the bytes):
    //   WTGPU_SORTED_INTERACT  0: k_interact (one kernel, every walk); 1: k_classify + one kernel per material class; 2: k_classify + k_interact_sorted
    //   WTGPU_STAGED_CONNECT   0: k_connect_strat (one kernel per strategy item); 1: k_connect_eval -> k_connect_shadow -> k_connect_mis, in chunks
    k.sorted_interact = u("WTGPU_SORTED_INTERACT", 0);
    k.primary_axis = u("WTGPU_PRIMARY_AXIS", 0);
    k.coop_io = u("WTGPU_COOP_IO", 0);   // pass A with wave-cooperative record transfers (k_interact_coop)
    k.staged_connect = u("WTGPU_STAGED_CONNECT", 0);
    k.conn_pool = std::max(1u, u("WTGPU_CONN_POOL", 16));
    if (const char* e = getenv("WTGPU_GRID_CLS")) {
        unsigned v[4] = {1, 4, 2, 4};
        sscanf(e, "%u,%u,%u,%u", &v[0], &v[1], &v[2], &v[3]);
        for (int c = 0; c < 4; ++c) k.grid_div_cls[c] = std::max(1u, v[c]);
    }
    k.flux_task_tris = std::max(64u, u("WTGPU_FLUX_TASK_TRIS", kFluxTaskTris));
    k.coop_aperture_min = u("WTGPU_COOP_APERTURE_MIN", 8);   // 0xFFFFFFFF: every aperture by a single lane of pass B
    if (const char* e = getenv("WTGPU_DEBUG_STAGE")) k.dbg_stage = atoi(e);   // bring-up aid: stops launching the round kernels after stage n (invalid results)
    if (const char* e = getenv("WTGPU_TIMING")) s->timing = atoi(e) != 0;
}

static int upload_impl(wtgpu_scene* s, int device, uint64_t max_batch) {
    (void)device;
    read_knobs(s);
    const scene_t& h = s->host;
    scene_t d = h;
    int rc;
#define UP(field, n) \
    if ((rc = upload(s, h.field, (size_t)(n), &d.field)) != WTGPU_OK) return rc;
    {   // the triangles, and behind them — same allocation — their bounding spheres (coop_tri_spheres, wt/coop.h: the first filter of the
        // wave-cooperative queries)
        d.tri_geo = nullptr;
        if (h.n_tris > 0 && h.tri_geo) {
            const size_t nt = h.n_tris;
            std::vector<float> sph(4 * nt);
            for (size_t i = 0; i < nt; ++i) tri_bounding_sphere(h.tri_geo[i].a, h.tri_geo[i].b, h.tri_geo[i].c, &sph[4 * i]);
            void* p = nullptr;
            HIP_CHECK(hipMalloc(&p, nt * (sizeof(tri_geo_t) + 16)));
            s->dev_allocs.push_back(p);
            HIP_CHECK(hipMemcpy(p, h.tri_geo, nt * sizeof(tri_geo_t), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(static_cast<char*>(p) + nt * sizeof(tri_geo_t), sph.data(), nt * 16, hipMemcpyHostToDevice));
            d.tri_geo = static_cast<const tri_geo_t*>(p);
        }
    }
    UP(tri_meta, h.n_tris)
    UP(tri_shade, h.n_tris)
    UP(edges, h.n_edges)
    {   // the nodes, and behind them — same allocation — the 128-byte nodes of the per-lane traversals and their grid (wt/bvh.h: lane_nodes)
        d.nodes = nullptr;
        if (h.n_nodes > 0 && h.nodes) {
            const size_t nn = h.n_nodes;
            vec3 mn{WT_INF, WT_INF, WT_INF}, mx{-WT_INF, -WT_INF, -WT_INF};
            for (size_t i = 0; i < nn; ++i)
                for (int c = 0; c < 8; ++c)
                    if (h.nodes[i].child[c] != 0) {
                        mn = vmin(mn, vec3{h.nodes[i].minx[c], h.nodes[i].miny[c], h.nodes[i].minz[c]});
                        mx = vmax(mx, vec3{h.nodes[i].maxx[c], h.nodes[i].maxy[c], h.nodes[i].maxz[c]});
                    }
            const qgrid_t g = qgrid_make(mn, mx);
            std::vector<bvh8_qnode_t> qn(nn);
            bool ok = finitef(mn.x) && finitef(mn.y) && finitef(mn.z) && finitef(mx.x) && finitef(mx.y) && finitef(mx.z);
            for (size_t i = 0; i < nn && ok; ++i) ok = qnode_make(h.nodes[i], g, qn[i]);
            if (!ok) return fail(WTGPU_ERR_INVALID, "the scene's BVH boxes cannot be enclosed by the 16-bit node grid (non-finite or out-of-range box)");
            const float gw[8] = {g.origin.x, g.origin.y, g.origin.z, g.cell.x, g.cell.y, g.cell.z, 0.f, 0.f};
            void* p = nullptr;
            HIP_CHECK(hipMalloc(&p, nn * (sizeof(bvh8_node_t) + sizeof(bvh8_qnode_t)) + sizeof(gw)));
            s->dev_allocs.push_back(p);
            HIP_CHECK(hipMemcpy(p, h.nodes, nn * sizeof(bvh8_node_t), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(static_cast<char*>(p) + nn * sizeof(bvh8_node_t), qn.data(), nn * sizeof(bvh8_qnode_t), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(static_cast<char*>(p) + nn * (sizeof(bvh8_node_t) + sizeof(bvh8_qnode_t)), gw, sizeof(gw), hipMemcpyHostToDevice));
            d.nodes = static_cast<const bvh8_node_t*>(p);
        }
    }
    UP(leaves, h.n_leaves)
    UP(shapes, h.n_shapes)
    size_t total_shape_tris = 0;
    for (uint32_t i = 0; i < h.n_shapes; ++i) total_shape_tris += h.shapes[i].tri_count;
    UP(shape_tri_tuid, total_shape_tris)
    UP(shape_tri_cdf, total_shape_tris + h.n_shapes)
    UP(materials, h.n_materials)
    UP(spectra, h.n_spectra)
    size_t spec_words = 0;
    for (uint32_t i = 0; i < h.n_spectra; ++i)
        if (h.spectra[i].type == SPEC_TABLE) spec_words = std::max(spec_words, (size_t)h.spectra[i].offset + (size_t)h.spectra[i].count * (h.spectra[i].is_complex ? 2 : 1));
    UP(spectra_data, spec_words)
    UP(textures, h.n_textures)
    size_t tex_words = 0;
    for (uint32_t i = 0; i < h.n_textures; ++i)
        if (h.textures[i].type == TEX_BITMAP)
            tex_words = std::max(tex_words, (size_t)h.textures[i].offset + (size_t)h.textures[i].width * h.textures[i].height * h.textures[i].channels);
        else if (h.textures[i].type == TEX_FUNCTION)
            tex_words = std::max(tex_words, (size_t)h.textures[i].offset + (size_t)h.textures[i].width);
    for (uint32_t i = 0; i < h.n_emitters; ++i)   // the texel tables of textured area emitters live in texture_data as well
        if (h.emitters[i].type == EMIT_AREA && h.emitters[i].radiance_tex > 0) {
            const emitter_t& e = h.emitters[i];
            if ((uint32_t)e.radiance_tex > h.n_textures || h.textures[e.radiance_tex - 1].type != TEX_BITMAP || e.shape < 0 || (uint32_t)e.shape >= h.n_shapes ||
                e.tab_words < h.shapes[e.shape].tri_count * 5ull + 1)
                return fail(WTGPU_ERR_INVALID, "area emitter " + std::to_string(i) + ": radiance_tex must name a bitmap texture and tab / tab_words the emitter's sampling tables (wt/sources.h area_table_*)");
            tex_words = std::max(tex_words, (size_t)e.tab + (size_t)e.tab_words);
        }
    UP(texture_data, tex_words)
    UP(emitters, h.n_emitters)
    UP(emitter_cdf, h.n_emitters + 1)
    UP(kdists, h.n_emitters)
    size_t kd_words = 0;
    for (uint32_t i = 0; i < h.n_emitters; ++i)
        if (!h.kdists[i].discrete) kd_words = std::max(kd_words, (size_t)h.kdists[i].offset + 2 * (size_t)h.kdists[i].count);
    UP(kdist_data, kd_words)
    if ((rc = upload(s, h.lut.icdf_theta1, h.lut.m ? h.lut.n_theta : 0, &d.lut.icdf_theta1)) != WTGPU_OK) return rc;
    if ((rc = upload(s, h.lut.icdf_theta2, h.lut.m ? h.lut.n_theta : 0, &d.lut.icdf_theta2)) != WTGPU_OK) return rc;
    if ((rc = upload(s, h.lut.icdf1, (size_t)h.lut.m * h.lut.m, &d.lut.icdf1)) != WTGPU_OK) return rc;
    if ((rc = upload(s, h.lut.icdf2, (size_t)h.lut.m * h.lut.m, &d.lut.icdf2)) != WTGPU_OK) return rc;
#undef UP
    s->dev = d;
    s->d_tri_class = nullptr;
    if (h.n_tris > 0 && h.opts.integrator == INTEGRATOR_BDPT) {   // the material-sorted pass A: class of every triangle (wt/bdpt.h: walk_class_of_triangle)
        std::vector<unsigned char> cls(h.n_tris);
        for (uint32_t t = 0; t < h.n_tris; ++t) cls[t] = (unsigned char)walk_class_of_triangle(h, t);
        if ((rc = upload(s, cls.data(), cls.size(), &s->d_tri_class)) != WTGPU_OK) return rc;
    }

    // per-batch path state: `n_slices` slices (one internal stream each), EACH holding a batch of up to `max_batch` samples.
    // Three internal streams: the tails of one batch overlap the bulk of the others.  A single batch already fills the GPU in its first rounds, so
    // more streams only add contention — measured with an unthrottled enqueue (16 MiB kernel-argument ring), ms per pass at 1 / 2 / 3 / 4 / 6 / 8
    // streams: 158 / 142 / 130 / 142 / 157 / 206 (headline); etoile 66 vs 78, bidir_room 69 vs 82 at 3 vs 4.
    // Batches as LARGE as the memory allows: every batch runs its ~30 rounds down to a thin tail, so the cost of the tails is per batch, not
    // per sample — measured on the headline workload (2.07 M samples per pass, three streams), samples per batch 0.23 / 0.35 / 0.69 / 1.38 /
    // 2.07 M -> 218 / 175 / 130 / 115 / 101 ms per pass.  288 GB of HBM are there to be used: three slices of a whole 1440^2 pass are 93 GB.
    const uint64_t npix = (uint64_t)h.sensor.width * h.sensor.height;
    uint32_t n_slices = 3;
    if (const char* e = getenv("WTGPU_STREAMS")) n_slices = (uint32_t)std::max(1, atoi(e));
    uint64_t batch_cap = max_batch ? std::min<uint64_t>(max_batch, 1u << 24) : std::min<uint64_t>(npix, 1u << 22);
    n_slices = (uint32_t)std::min<uint64_t>(n_slices, std::max<uint64_t>(1, batch_cap / 64));
    {   // the vertex stores grow with max_depth (2 x (max_depth + 2) vertices of 356 B per sample): keep the state of all slices within a budget
        // (WTGPU_STATE_GB, default 224 of the 288 GB, and never more than 85 % of what is free) by shrinking the batches of deep scenes — more, smaller batches, same results
        const bool pm = h.opts.integrator != INTEGRATOR_BDPT;
        const uint64_t mv = (uint64_t)h.opts.max_depth + 2;
        uint64_t per_sample = 4ull * (2 * ((pm ? kPathWalkWords : kWalkWords) + (pm ? 0 : mv * kVertexWords) + kTravWords + kStageWords + 2 + kTriListWords) + kCtxWords) + 64ull * 28ull + 2048ull;
        // plt_path: two wedge pools of 48 records per walk, the deferred-NEE records, the queues of the wave-per-walk kernels
        if (pm) per_sample += 2ull * 48ull * sizeof(utd_edge_rec_t) + sizeof(path_nee_rec_t) + 3ull * 4ull + 4ull + sizeof(uint2);
        else per_sample += (s->knobs.staged_connect ? (uint64_t)s->knobs.conn_pool * (sizeof(conn_pending_t) + 4ull) : 0ull) + (s->knobs.sorted_interact ? 4ull * 2ull * kNumWalkClasses : 0ull);   // pending connections, class queues
        uint64_t budget = 224ull << 30;   // of the MI355X's 288 GB (three slices of a two-pass 1440^2 batch are 186 GB); WTGPU_STATE_GB overrides
        if (const char* e = getenv("WTGPU_STATE_GB")) budget = (uint64_t)std::max(1, atoi(e)) << 30;
        // ... and within what the device has free right now (another scene, torch's caching allocator, a smaller GPU): 85 % of it, the rest is
        // for the per-slice pools (edge ids, region-sum tasks, Fraunhofer segments: ~0.3 GB per slice) and the caller's films
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) budget = std::min<uint64_t>(budget, (uint64_t)((double)free_b * 0.85));
        const uint64_t fixed = (uint64_t)n_slices * ((1ull << 23) * 4ull + (1ull << 22) * 8ull + (64ull << 20));   // pools that do not scale with the batch
        budget = budget > 2 * fixed ? budget - fixed : budget / 2;
        const uint64_t fit = std::max<uint64_t>(4096, budget / per_sample / n_slices);
        if (batch_cap > fit) batch_cap = fit;
    }
    unsigned long long* counters = nullptr;
    if ((rc = dmalloc(s, &counters, kNumCounters + kProfSlots + 1))) return rc;
    HIP_CHECK(hipMemset(counters, 0, (kNumCounters + kProfSlots + 1) * sizeof(unsigned long long)));
    s->slices.resize(n_slices);
    s->streams.resize(n_slices);
    s->ev_done.resize(n_slices);
    for (uint32_t k = 0; k < n_slices; ++k) {
        device_state_t& st = s->slices[k];
        st.cap = batch_cap;
        st.max_verts = (uint32_t)h.opts.max_depth + 2;
        st.walk_words = (uint32_t)(h.opts.integrator != INTEGRATOR_BDPT ? kPathWalkWords : kWalkWords);
        st.vert_words = (size_t)st.max_verts * kVertexWords;
        st.counters = counters;
        const size_t W2 = 2 * (size_t)st.cap;
        const bool path_mode = h.opts.integrator != INTEGRATOR_BDPT;   // plt_path: no vertex store / strategy buckets / Fraunhofer pool
        if ((rc = dmalloc(s, &st.walks, (path_mode ? kPathWalkWords : kWalkWords) * W2))) return rc;
        {
            path_state_t P;
            P.utd_cap = path_mode ? (uint32_t)std::min<uint64_t>(48ull * st.cap + 65536, 1ull << 28) : 1u;   // measured mean on the 576-building etoile: 11 wedges per aperture
            for (int q = 0; q < 2; ++q) {
                if ((rc = dmalloc(s, &P.utd[q], (size_t)P.utd_cap))) return rc;
                if ((rc = dmalloc(s, &P.fsdq[q], path_mode ? (size_t)st.cap : 1))) return rc;
            }
            if ((rc = dmalloc(s, &P.neeq, path_mode ? (size_t)st.cap : 1))) return rc;
            if ((rc = dmalloc(s, &P.fsd_f, path_mode ? (size_t)st.cap : 1))) return rc;
            if ((rc = dmalloc(s, &P.nee_recs, path_mode ? (size_t)st.cap : 1))) return rc;
            if ((rc = dmalloc(s, &P.gather_info, path_mode ? (size_t)st.cap : 1))) return rc;
            path_state_t* dP = nullptr;
            if ((rc = dmalloc(s, &dP, 1))) return rc;
            HIP_CHECK(hipMemcpy(dP, &P, sizeof(P), hipMemcpyHostToDevice));
            s->d_path_slices.push_back(dP);
        }
        if (!path_mode) {
            bdpt_ext_t X;
            X.tri_class = s->d_tri_class;
            if (s->knobs.sorted_interact && (rc = dmalloc(s, &X.cls_queue, (size_t)kNumWalkClasses * W2))) return rc;
            // staged connections: the strategy items of a batch are connected in chunks of pend_cap = `conn_pool` (16; WTGPU_CONN_POOL) x batch size
            // items — a chunk's pending connections cannot outnumber its items; as many chunks as a batch of this scene's depth can hold items for
            // (every (s,t) with s + t - 2 <= max_depth for every sample: 186 per sample at max_depth 16 => 12 chunks, all but the first one or two empty)
            X.pend_cap = (uint32_t)std::min<uint64_t>((uint64_t)s->knobs.conn_pool * st.cap + 4096, 0xFFFFFF00ull);
            if (s->knobs.staged_connect) {
                uint64_t pairs = 0;
                const int md = h.opts.max_depth;
                for (int t = 0; t <= md + 2; ++t)
                    for (int q = 0; q <= md + 2; ++q)
                        if (t + q - 2 >= 0 && t + q - 2 <= md && !(t == 1 && q == 1)) ++pairs;
                X.n_chunks = (uint32_t)((pairs * st.cap + X.pend_cap - 1) / X.pend_cap);
                if ((rc = dmalloc(s, &X.pend, (size_t)X.pend_cap))) return rc;
                if ((rc = dmalloc(s, &X.surv, (size_t)X.pend_cap))) return rc;
                if ((rc = dmalloc(s, &X.chunk_ctl, (size_t)X.n_chunks * kChunkCtlWords))) return rc;
            } else
                X.pend_cap = 0;
            s->pend_cap = X.pend_cap;
            s->n_chunks = X.n_chunks;
            bdpt_ext_t* dX = nullptr;
            if ((rc = dmalloc(s, &dX, 1))) return rc;
            HIP_CHECK(hipMemcpy(dX, &X, sizeof(X), hipMemcpyHostToDevice));
            st.ext = dX;
        }
        if ((rc = dmalloc(s, &st.verts, path_mode ? 1 : (size_t)st.max_verts * kVertexWords * W2))) return rc;
        if ((rc = dmalloc(s, &st.ctx, kCtxWords * (size_t)st.cap))) return rc;
        if ((rc = dmalloc(s, &st.trav, (kTravWords + kStageWords) * W2))) return rc;   // (+ the staged trace kernels' records: trace_stage_words)
        if ((rc = dmalloc(s, &st.tris, (size_t)kTriListWords * W2))) return rc;
        if ((rc = dmalloc(s, &st.queue[0], W2))) return rc;
        if ((rc = dmalloc(s, &st.queue[1], W2))) return rc;
        if ((rc = dmalloc(s, &st.heavy_queue, 3 * W2))) return rc;   // (+ the staged trace kernels' two queues: trace_pol_queue / trace_cone_queue)
        if ((rc = dmalloc(s, &st.intb_queue, W2))) return rc;
        if ((rc = dmalloc(s, &st.gather_queue, W2))) return rc;
        if ((rc = dmalloc(s, &st.intc_queue, W2))) return rc;
        if ((rc = dmalloc(s, &st.intd_queue, W2))) return rc;
        st.ftask_cap = path_mode ? 1u : (1u << 22);
        if ((rc = dmalloc(s, &st.ftasks, (size_t)st.ftask_cap))) return rc;
        if ((rc = dmalloc(s, &st.facc, path_mode ? 1 : W2))) return rc;
        st.epool_cap = 1u << 23;
        if ((rc = dmalloc(s, &st.epool, (size_t)st.epool_cap))) return rc;
        if ((rc = dmalloc(s, &st.ctl, (size_t)CTL_WORDS))) return rc;
        HIP_CHECK(hipMemset(st.ctl, 0, CTL_WORDS * sizeof(uint32_t)));
        st.fsd_cap = (h.opts.FSD && !h.opts.force_ray_tracing && !path_mode) ? (uint32_t)std::min<uint64_t>(W2, 1u << 22) : 1u;
        if ((rc = dmalloc(s, &st.fsd_hdr, st.fsd_cap))) return rc;
        // apertures own variable-size ranges of one segment pool: 64 records per sample in flight (measured mean of the headline
        // workload: 1.6 per sample; an aperture holds up to kFsdMaxEdges = 4096)
        st.fsd_ecap = st.fsd_cap > 1 ? (uint32_t)std::min<uint64_t>(64ull * st.cap + kFsdMaxEdges, 1ull << 28) : 1u;
        if ((rc = dmalloc(s, &st.fsd_edges, (size_t)st.fsd_ecap))) return rc;
        if ((rc = dmalloc(s, &st.strat_items, path_mode ? 1 : (size_t)kNumKeys * st.cap))) return rc;
        if ((rc = dmalloc(s, &st.strat_count, (size_t)kNumKeys))) return rc;
        if ((rc = dmalloc(s, &st.strat_prefix, (size_t)kNumKeys + 1))) return rc;
        if ((rc = dmalloc(s, &st.lacc, 4 * (size_t)st.cap))) return rc;
        HIP_CHECK(hipMemset(st.strat_count, 0, kNumKeys * sizeof(uint32_t)));
        HIP_CHECK(hipStreamCreateWithFlags(&s->streams[k], hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&s->ev_done[k], hipEventDisableTiming));
    }
    HIP_CHECK(hipEventCreateWithFlags(&s->ev_begin, hipEventDisableTiming));
    // in-flight batch records: events for per-kernel timings + pinned snapshot of the control block
    s->recs.resize(4 * (size_t)n_slices);
    s->pending.assign(n_slices, wtgpu_scene::pending_t{});
    for (auto& r : s->recs) {
        r.ev.resize(s->timing ? 3 + 6 * (size_t)kMaxWalkIters : 1);
        for (auto& e : r.ev) HIP_CHECK(hipEventCreate(&e));
        HIP_CHECK(hipHostMalloc((void**)&r.h_ctl, CTL_WORDS * sizeof(uint32_t), hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void**)&r.h_mid, CTL_WORDS * sizeof(uint32_t), hipHostMallocDefault));
        HIP_CHECK(hipEventCreateWithFlags(&r.ev_mid, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&r.ev_stagger, hipEventDisableTiming));
    }
    s->uploaded = true;
    return WTGPU_OK;
}

static void note_rounds(wtgpu_scene* s, uint32_t n) { s->rounds_hist[s->rounds_hist_n++ % 8u] = n; }
// Waits for one in-flight batch record and folds its event timings / control-block snapshot into the accumulators.
static int drain_rec(wtgpu_scene* s, chunk_rec_t& r) {
    if (!r.busy) return WTGPU_OK;
    HIP_CHECK(hipEventSynchronize(r.ev[r.ev_final]));
    const uint32_t rounds = r.h_ctl[CTL_ROUNDS];
    s->cap_hits += r.h_ctl[CTL_COUNT0 + (r.rounds_launched & 1u)] + r.h_ctl[CTL_BACK0 + (r.rounds_launched & 1u)];   // walks still active after the last round
    s->acc[4] += rounds;
    s->acc[5] += rounds;
    s->acc[6] += 1;
    s->rounds_launched_total += r.rounds_launched;
    if (s->timing) {
        // (an event pair that cannot be resolved contributes 0 ms: timings are diagnostics, the render itself has completed)
        auto elapsed = [](hipEvent_t a, hipEvent_t b) {
            float ms = 0.f;
            return hipEventElapsedTime(&ms, a, b) == hipSuccess ? ms : 0.f;
        };
        s->acc[0] += elapsed(r.ev[0], r.ev[1]);
        size_t e = 1;
        for (uint32_t k = 0; k < r.rounds_timed; ++k, e += 6) {
            static const int slot[6] = {1, 7, 2, 8, 9, 10};   // trace, heavy trace, pass A, edges + pass B, region flux, pass C
            for (int q = 0; q < 6; ++q) s->acc[slot[q]] += elapsed(r.ev[e + q], r.ev[e + q + 1]);
        }
        s->acc[3] += elapsed(r.ev[e], r.ev[r.ev_final]);   // (the connections' bracket: from the last timed round's end to the batch's end)
    }
    r.busy = false;
    return WTGPU_OK;
}

// ---- WTGPU_TRACE_AB: in-situ replay of a round's trace queue through both per-lane trace kernels (the "replay harness": each variant sees exactly the
// queue, walk records and scene the pipeline produced, so what is timed is the real mix of beam widths and what is compared is every word they write)
__global__ void __launch_bounds__(256) k_ab_compare(const uint32_t* x, const uint32_t* y, size_t n, unsigned long long* out) {
    unsigned long long bad = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) bad += x[i] != y[i] ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(out, bad);
}
__global__ void __launch_bounds__(256) k_ab_queue_sums(const uint32_t* q, const uint32_t* count, unsigned long long* out) {   // order-independent checksums of a queue
    unsigned long long s1 = 0, s2 = 0;
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        s1 += q[i];
        s2 += (unsigned long long)q[i] * 2654435761ull + ((unsigned long long)q[i] << 7 ^ q[i]);
    }
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out, s1);
        atomicAdd(out + 1, s2);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(out + 2, (unsigned long long)n);
}
// the per-lane traversal of a round in its alternative forms (the default, k_trace_refill, is launched by batch_launcher_t::rounds itself)
static void launch_trace_alt(const wtgpu_scene* s, const launch_args_t& a, hipStream_t st_, int in, int first, uint32_t round, uint32_t g0) {
    const wtgpu_scene::knobs_t& K = s->knobs;
    if (K.trace_staged) {
        hipLaunchKernelGGL(k_tr_axis, dim3(g0), dim3(kBlock), 0, st_, a, in, first, round);
        for (uint32_t it = 0; it < K.trace_stages; ++it) {
            const uint32_t g = std::max<uint32_t>(1u, g0 >> std::min(it, 4u));   // (the queues shrink from stage to stage; a grid that is too small only loops longer)
            if (it > 0) hipLaunchKernelGGL(k_tr_policy, dim3(g), dim3(kBlock), 0, st_, a, it);
            hipLaunchKernelGGL(k_tr_cone, dim3(g), dim3(kBlock), 0, st_, a, it);
        }
        hipLaunchKernelGGL(k_tr_tail, dim3(std::max<uint32_t>(1u, g0 >> 3)), dim3(kBlock), 0, st_, a, K.trace_stages);
    } else
        hipLaunchKernelGGL(k_trace_sm, dim3(g0), dim3(kBlock), 0, st_, a, in, first, round);
}
static int trace_ab_round(wtgpu_scene* s, const launch_args_t& a, hipStream_t st_, int in, int first, uint32_t round, uint32_t g0) {
    const size_t n_trav = 2 * (size_t)a.st.cap * kTravWords, n_tris = 2 * (size_t)a.st.cap * kTriListWords;
    uint32_t *b_trav = nullptr, *b_tris = nullptr;
    unsigned long long* d_out = nullptr;
    uint32_t h_n[2] = {0, 0};
    HIP_CHECK(hipMalloc(&b_trav, n_trav * 4));
    HIP_CHECK(hipMalloc(&b_tris, n_tris * 4));
    HIP_CHECK(hipMalloc(&d_out, 8 * sizeof(unsigned long long)));
    HIP_CHECK(hipMemsetAsync(d_out, 0, 8 * sizeof(unsigned long long), st_));
    hipEvent_t ev[4];
    for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
    HIP_CHECK(hipMemcpyAsync(h_n, a.st.ctl + CTL_COUNT0 + in, 4, hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipMemcpyAsync(h_n + 1, a.st.ctl + CTL_BACK0 + in, 4, hipMemcpyDeviceToHost, st_));
    // A: k_trace_refill
    HIP_CHECK(hipEventRecord(ev[0], st_));
    hipLaunchKernelGGL(k_trace_refill, dim3(g0), dim3(kBlock), 0, st_, a, in, first, round);
    HIP_CHECK(hipEventRecord(ev[1], st_));
    HIP_CHECK(hipMemcpyAsync(b_trav, a.st.trav, n_trav * 4, hipMemcpyDeviceToDevice, st_));
    HIP_CHECK(hipMemcpyAsync(b_tris, a.st.tris, n_tris * 4, hipMemcpyDeviceToDevice, st_));
    hipLaunchKernelGGL(k_ab_queue_sums, dim3(64), dim3(256), 0, st_, a.st.heavy_queue, a.st.ctl + CTL_HEAVY_COUNT, d_out + 1);
    // the queue again from its start, an empty heavy queue
    HIP_CHECK(hipMemsetAsync(a.st.ctl + CTL_HEAD_TRACE, 0, 4, st_));
    HIP_CHECK(hipMemsetAsync(a.st.ctl + CTL_HEAVY_COUNT, 0, 4, st_));
    // B: the alternative form the knobs select (k_trace_sm, or the staged kernels)
    HIP_CHECK(hipEventRecord(ev[2], st_));
    launch_trace_alt(s, a, st_, in, first, round, g0);
    HIP_CHECK(hipEventRecord(ev[3], st_));
    hipLaunchKernelGGL(k_ab_compare, dim3(2048), dim3(256), 0, st_, b_trav, a.st.trav, n_trav, d_out);
    hipLaunchKernelGGL(k_ab_compare, dim3(2048), dim3(256), 0, st_, b_tris, a.st.tris, n_tris, d_out);
    hipLaunchKernelGGL(k_ab_queue_sums, dim3(64), dim3(256), 0, st_, a.st.heavy_queue, a.st.ctl + CTL_HEAVY_COUNT, d_out + 4);
    unsigned long long h_out[8];
    HIP_CHECK(hipMemcpyAsync(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    float ms_a = 0.f, ms_b = 0.f;
    (void)hipEventElapsedTime(&ms_a, ev[0], ev[1]);
    (void)hipEventElapsedTime(&ms_b, ev[2], ev[3]);
    const uint64_t bad = h_out[0] + (h_out[1] != h_out[4]) + (h_out[2] != h_out[5]) + (h_out[3] != h_out[6]);
    s->ab_ms[0] += ms_a;
    s->ab_ms[1] += ms_b;
    s->ab_mismatch += bad;
    s->ab_walks += (uint64_t)h_n[0] + h_n[1];
    s->ab_rounds++;
    if (getenv("WTGPU_TRACE_AB_VERBOSE"))
        fprintf(stderr, "[trace ab] round %2u: %8u walks  refill %8.3f ms  alt %8.3f ms  heavy %llu / %llu  differing words %llu\n", round, h_n[0] + h_n[1], ms_a, ms_b,
                h_out[3], h_out[6], (unsigned long long)bad);
    for (auto& e : ev) (void)hipEventDestroy(e);
    (void)hipFree(b_trav);
    (void)hipFree(b_tris);
    (void)hipFree(d_out);
    return WTGPU_OK;
}

// ---- enqueueing a batch -------------------------------------------------------------------------------------------------------------
// The launches of one batch, in two parts.  FIRST: generation and the rounds its walks are expected to need — the rounds with work of the
// last batches on average + a margin (`rounds_hist`; a guess of 32 until a batch has been seen) — then a copy of the control block to pinned
// memory and an event.  FINISH (when the slice is needed again, at wtgpu_join, or before results are read): the host waits for that event;
// if the round queue is NOT empty — a batch whose walks outlasted the expectation — 8 more rounds are launched and the host looks again, up
// to kMaxWalkIters; then the connections.  Nothing is ever dropped that a blind launch of every round (as until round 4) would have kept.
// Why: ~25 of the 96 rounds have work; the other ~70 x 9 launches only find an empty queue, and cost 2.6 % of a pass on the headline
// workload and 3 % on the 720 x 540 film (run r4r: the same build launching 96 / 48 / 32 rounds).
struct batch_launcher_t {
    wtgpu_scene* s;
    const wtgpu_scene::knobs_t& K;
    uint32_t grid_round = 0, grid_heavy = 0;
    bool path_mode = false;
    bool ev_fail = false;
    bool hp_on = false, trace_on = false;
    double hp_t[32] = {0};
    unsigned long hp_n[32] = {0};
    explicit batch_launcher_t(wtgpu_scene* s_) : s(s_), K(s_->knobs) {
        int n_cu = 256;   // persistent grids: enough blocks to fill the 256 CUs; wavefronts pull work until the queue is empty
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, s->device);
        grid_round = (uint32_t)n_cu * K.round_blocks_per_cu;
        grid_heavy = (uint32_t)n_cu * K.heavy_waves_per_cu;
        path_mode = s->host.opts.integrator != INTEGRATOR_BDPT;
        static const bool hp = getenv("WTGPU_HOST_PROF") != nullptr;   // WTGPU_HOST_PROF=1: host time spent inside each kind of launch call (diagnostic)
        hp_on = hp;
        static const bool tr = getenv("WTGPU_TRACE_LAUNCH") != nullptr;
        trace_on = tr;
    }
#define HP_LAUNCH(slot, ...)                                                                                              \
    do {                                                                                                                  \
        if (trace_on) {   /* WTGPU_TRACE_LAUNCH=1 (bring-up aid): every launch is named and waited for — the last line names a kernel that hangs */ \
            fprintf(stderr, "[wtgpu launch] slot %d ...", slot);                                                            \
            hipLaunchKernelGGL(__VA_ARGS__);                                                                              \
            const hipError_t e_ = hipDeviceSynchronize();                                                                 \
            fprintf(stderr, " done (%s)\n", hipGetErrorString(e_));                                                       \
        } else if (hp_on) {                                                                                                      \
            const auto t0_ = std::chrono::steady_clock::now();                                                            \
            hipLaunchKernelGGL(__VA_ARGS__);                                                                              \
            hp_t[slot] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0_).count();      \
            hp_n[slot]++;                                                                                                 \
        } else                                                                                                            \
            hipLaunchKernelGGL(__VA_ARGS__);                                                                              \
    } while (0)
    void rec(chunk_rec_t& r, hipStream_t st_) {
        const auto hp0_ = std::chrono::steady_clock::now();
        if (s->timing && r.ev_used + 1 >= r.ev.size()) return;   // (rounds beyond what the event array holds — a batch with a very long walk — are not timed: the last event is the batch's)
        if (s->timing && hipEventRecord(r.ev[r.ev_used++], st_) != hipSuccess) ev_fail = true;
        if (hp_on) { hp_t[31] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - hp0_).count(); hp_n[31]++; }
    }
    uint32_t g_full(uint32_t nb) const { return std::min<uint32_t>(grid_round, ((path_mode ? 1u : 2u) * nb + kBlock - 1) / kBlock); }
    void generate(const launch_args_t& a, chunk_rec_t& r, hipStream_t st_) {
        r.ev_used = 0;
        r.rounds_launched = 0;
        r.rounds_timed = 0;
        rec(r, st_);
        if (path_mode)
            HP_LAUNCH(0, k_path_generate, dim3((a.nb + kBlock - 1) / kBlock), dim3(kBlock), 0, st_, a);
        else
            HP_LAUNCH(1, k_generate, dim3((a.nb + kBlock - 1) / kBlock), dim3(kBlock), 0, st_, a);
        rec(r, st_);
    }
    // stage_from (plt_bdpt): the first round of the range begins at this stage — 0 trace, 1 wave-cooperative trace, 3 k_edges, 5 region sums (a round k_light_rounds stopped in)
    int rounds(const launch_args_t& a, const path_state_t* ps, chunk_rec_t& r, hipStream_t st_, uint32_t r_begin, uint32_t r_end, int stage_from = 0) {
        const uint32_t nb = a.nb, walks_per_sample = path_mode ? 1u : 2u, gf = g_full(nb);
        const uint32_t grid_div_b = K.grid_div_b, grid_div_c = K.grid_div_c, grid_mul_flux = K.grid_mul_flux;   // persistent grids of the expensive-interaction passes relative to the round's
        const int dbg_stage = K.dbg_stage;
        for (uint32_t round = r_begin; round < r_end; ++round) {
            const int in = (int)(round & 1u), first = round == 0 ? 1 : 0;
            if (s->timing && r.ev_used + 7 < r.ev.size()) r.rounds_timed++;   // (rec() below records this round's six events only while they fit)
            if (round == K.stagger_round && K.stagger_round > 0) {
                HIP_CHECK(hipEventRecord(r.ev_stagger, st_));
                s->ev_stagger_last = r.ev_stagger;
            }
            // the queue roughly halves every round and is normally empty after ~25: later rounds get smaller persistent grids
            // (an empty launch costs its grid size; a grid that turns out too small only takes longer, the wavefronts loop)
            uint32_t g0, gh;
            if (K.decay_q > 0) {
                // geometric schedule: the queue of round r holds ~ N q^r walks (q ~ 0.55 in the headline workload); grids follow with a safety
                // factor — an undersized persistent grid only takes longer, an oversized one on a short queue holds up the other streams
                const double f = std::min(1.0, (double)K.decay_c * std::pow(K.decay_q * .01, (double)round));
                g0 = std::max<uint32_t>(2u, (uint32_t)(gf * f));
                gh = std::max<uint32_t>(2u, (uint32_t)(std::min<uint32_t>(grid_heavy, walks_per_sample * nb) * f));
            } else {
                const uint32_t shrink = round < K.shrink_r1 ? 1u : (round < K.shrink_r2 ? K.shrink_f1 : K.shrink_f2);
                g0 = std::max<uint32_t>(1u, gf / shrink);
                gh = std::max<uint32_t>(1u, std::min<uint32_t>(grid_heavy, walks_per_sample * nb) / (round < K.shrink_r1 ? 1u : (round < K.shrink_r2 ? K.shrink_h1 : 32u)));
            }
            const int sf = round == r_begin ? stage_from : 0;
            if (sf <= 0 && dbg_stage >= 2 + 3 * (int)round) {
                if (round < K.trace_ab) {
                    const int rc = trace_ab_round(s, a, st_, in, first, round, g0);
                    if (rc) return rc;
                } else if (K.trace_sm || (K.trace_staged && round < K.trace_staged_rounds))
                    launch_trace_alt(s, a, st_, in, first, round, g0);
                else
                    HP_LAUNCH(3, k_trace_refill, dim3(g0), dim3(kBlock), 0, st_, a, in, first, round);
            }
            rec(r, st_);
            if (sf <= 1 && dbg_stage >= 3 + 3 * (int)round) HP_LAUNCH(5, k_trace_heavy, dim3(gh), dim3(64), 0, st_, a);
            rec(r, st_);
            if (path_mode) {
                if (round > 0) HP_LAUNCH(6, k_path_fsd, dim3(gh), dim3(64), 0, st_, a, ps, round);
                if (dbg_stage >= 4 + 3 * (int)round) HP_LAUNCH(7, k_path_interact, dim3(g0), dim3(kBlock), 0, st_, a, ps, in, first, round);
                HP_LAUNCH(8, k_path_edges, dim3(gh), dim3(64), 0, st_, a, ps);
                HP_LAUNCH(9, k_path_interact_b, dim3(std::max<uint32_t>(1u, g0 / 2u)), dim3(kBlock), 0, st_, a, ps, in, round);
                HP_LAUNCH(10, k_path_nee, dim3(gh), dim3(64), 0, st_, a, ps, round);
                rec(r, st_);
                rec(r, st_);
                rec(r, st_);
                rec(r, st_);
                continue;
            }
            if (sf > 2) {
            } else if (K.sorted_interact) {
                HP_LAUNCH(11, k_classify, dim3(g0), dim3(kBlock), 0, st_, a, in, first);
                if (K.sorted_interact >= 2)
                    HP_LAUNCH(24, k_interact_sorted, dim3(g0), dim3(kBlock), 0, st_, a, in);
                else {
                HP_LAUNCH(24, k_interact_diffuse, dim3(std::max<uint32_t>(1u, g0 / K.grid_div_cls[0])), dim3(kBlock), 0, st_, a, in);
                HP_LAUNCH(25, k_interact_dielectric, dim3(std::max<uint32_t>(1u, g0 / K.grid_div_cls[1])), dim3(kBlock), 0, st_, a, in);
                HP_LAUNCH(26, k_interact_spm, dim3(std::max<uint32_t>(1u, g0 / K.grid_div_cls[2])), dim3(kBlock), 0, st_, a, in);
                HP_LAUNCH(27, k_interact_any, dim3(std::max<uint32_t>(1u, g0 / K.grid_div_cls[3])), dim3(kBlock), 0, st_, a, in);
                }
            } else if (K.coop_io)
                HP_LAUNCH(11, k_interact_coop, dim3(g0), dim3(kBlock), 0, st_, a, in, first);
            else
                HP_LAUNCH(11, k_interact, dim3(g0), dim3(kBlock), 0, st_, a, in, first);
            rec(r, st_);
            if (sf <= 3) HP_LAUNCH(12, k_edges, dim3(gh), dim3(64), 0, st_, a);
            if (sf <= 4) HP_LAUNCH(13, k_interact_b, dim3(std::max<uint32_t>(1u, g0 / grid_div_b)), dim3(kBlock), 0, st_, a, in);
            rec(r, st_);
            HP_LAUNCH(14, k_flux_split, dim3(std::max<uint32_t>(1u, gh / 4u)), dim3(64), 0, st_, a);
            HP_LAUNCH(15, k_flux_tasks, dim3(std::max<uint32_t>(1u, gh * grid_mul_flux)), dim3(64), 0, st_, a);
            rec(r, st_);
            HP_LAUNCH(16, k_interact_c, dim3(std::max<uint32_t>(1u, gh / grid_div_c)), dim3(64), 0, st_, a, in);
            HP_LAUNCH(17, k_interact_c_hard, dim3(std::max<uint32_t>(1u, gh / K.grid_div_hard)), dim3(WTGPU_HARD_BLOCK), 0, st_, a, in);
            rec(r, st_);
        }
        r.rounds_launched = r_end;
        return WTGPU_OK;
    }
    // k_light_rounds (kernels_walk.hip) behind the rounds launched so far: whatever the batch's last walks still need, in one launch of one block
    // (nothing, when the queue is empty; plt_bdpt only)
    void light(const launch_args_t& a, hipStream_t st_, uint32_t launched) {
        if (path_mode || !K.light_rounds || launched >= K.max_rounds) return;
        HP_LAUNCH(2, k_light_rounds, dim3(1), dim3(kBlock), 0, st_, a, (int)(launched & 1u), launched, K.max_rounds - launched);
    }
    // after the batch's last round: connections (plt_bdpt) / what is left of the walks (plt_path), the control block's snapshot, the closing event
    int tail(const launch_args_t& a, chunk_rec_t& r, hipStream_t st_) {
        const uint32_t nb = a.nb, gf = g_full(nb);
        if (path_mode) {
            HP_LAUNCH(18, k_path_flush, dim3(kFlushGrid), dim3(kBlock), 0, st_, a, (int)(r.rounds_launched & 1u));
        } else {
            HP_LAUNCH(19, k_connect_enum, dim3((nb + kEnumBlock - 1) / kEnumBlock), dim3(kEnumBlock), 0, st_, a);
            HP_LAUNCH(20, k_connect_scan, dim3(1), dim3(64), 0, st_, a);
            const bool open = (uint32_t)s->host.opts.max_depth + 2 >= kKeyDim - 1;
            // staged connections (chunked, see upload_impl); subpaths beyond 17 vertices (open-ended strategy buckets: an item there holds several
            // strategies) keep the one-kernel form
            if (K.staged_connect && !open) {
                for (uint32_t c = 0; c < s->n_chunks; ++c) {
                    const uint32_t g = c == 0 ? gf : std::max<uint32_t>(1u, gf / 8u);   // (later chunks are normally empty: small grids, they only loop longer when not)
                    HP_LAUNCH(28, k_connect_eval, dim3(g), dim3(kBlock), 0, st_, a, c);
                    HP_LAUNCH(29, k_connect_shadow, dim3(g), dim3(kBlock), 0, st_, a, c);
                    HP_LAUNCH(30, k_connect_mis, dim3(g), dim3(kBlock), 0, st_, a, c);
                }
            } else {
                HP_LAUNCH(21, k_connect_strat, dim3(gf), dim3(kBlock), 0, st_, a);
                if (open) HP_LAUNCH(22, k_connect_strat_open, dim3(std::max<uint32_t>(1u, gf / 8u)), dim3(kBlock), 0, st_, a);
            }
            // (the tiled splat pays off when the batch holds most of the film's elements: it visits every row segment of the film)
            const uint32_t fw = a.film.width, fh = a.film.height, planes = film_planes(s->host.sensor);
            if (K.tiled_splat && s->host.sensor.rf_radius <= 1 && planes <= 16 && (uint64_t)nb * 2u >= (uint64_t)a.npix)
                HP_LAUNCH(23, k_connect_splat_tiled, dim3(fh * ((fw + kBlock - 1) / kBlock)), dim3(kBlock), 3 * kSplatCols * (planes + 1) * sizeof(double), st_, a);
            else
                HP_LAUNCH(23, k_connect_splat, dim3((nb + kBlock - 1) / kBlock), dim3(kBlock), 0, st_, a);
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(r.h_ctl, a.st.ctl, CTL_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
        r.ev_final = s->timing ? r.ev_used : 0;
        HIP_CHECK(hipEventRecord(r.ev[r.ev_final], st_));
        if (ev_fail) return fail(WTGPU_ERR_HIP, "hipEventRecord failed");
        r.busy = true;
        return WTGPU_OK;
    }
#undef HP_LAUNCH
    void report() const {
        if (!hp_on) return;
        static const char* hp_names[] = {"k_path_generate","k_generate","(unused)","k_trace_refill","(unused)","k_trace_heavy","k_path_fsd","k_path_interact","k_path_edges","k_path_interact_b","k_path_nee","k_interact","k_edges","k_interact_b","k_flux_split","k_flux_tasks","k_interact_c","k_interact_c_hard","k_path_flush","k_connect_enum","k_connect_scan","k_connect_strat","k_connect_strat_open","k_connect_splat","k_interact_diffuse","k_interact_dielectric","k_interact_spm","k_interact_any","k_connect_eval","k_connect_shadow","k_connect_mis"};
        for (int i = 0; i < 31; ++i)
            if (hp_n[i]) fprintf(stderr, "[host prof] %-22s %6lu calls %9.1f us total %7.2f us each\n", hp_names[i], hp_n[i], hp_t[i], hp_t[i] / hp_n[i]);
        if (hp_n[31]) fprintf(stderr, "[host prof] %-22s %6lu calls %9.1f us total %7.2f us each\n", "hipEventRecord", hp_n[31], hp_t[31], hp_t[31] / hp_n[31]);
    }
};
static_assert(sizeof(launch_args_t) <= sizeof(wtgpu_scene::pending_t::args), "pending_t::args holds a launch block");

// rounds to launch up front: what the recent batches needed ON AVERAGE + a small margin.  (Not their maximum: the number of rounds a batch needs
// is set by its single longest walk — 36 on average on the headline workload, now and then 60 — and a batch that needs more than expected only
// costs another look, 8 rounds at a time.)  WTGPU_FIRST_ROUNDS forces a number: tests use 2, 96 = as before round 4.
static uint32_t expected_rounds(const wtgpu_scene* s) {
    if (s->knobs.first_rounds) return s->knobs.first_rounds;
    if (s->rounds_hist_n == 0) return std::min<uint32_t>(kMaxWalkIters, 32u);   // nothing seen yet: a guess
    const uint32_t n = std::min<uint32_t>(s->rounds_hist_n, 8u);
    uint32_t sum = 0, mx = 0;
    for (uint32_t i = 0; i < n; ++i) {
        sum += s->rounds_hist[i];
        mx = std::max(mx, s->rounds_hist[i]);
    }
    const uint32_t mean = (sum + n - 1) / n;
    (void)mx;
    // (Batches that need THOUSANDS of rounds — bidir_room: a handful of walks restart behind empty apertures 1800-3800 times per 4.2 M-sample batch — are
    // not given them up front: 4096 rounds are 37,000 launches, 330 ms of host time per batch, measured 11.9 Msamples/s against 15.8 with the rounds
    // added as the host sees the queue still filled.  Both are far from the 36.1 of WTGPU_MAX_ROUNDS=96, which drops those walks: DESIGN.md §0.)
    return std::min<uint32_t>(kMaxWalkIters, mean + s->knobs.rounds_margin);
}
// One LOOK at the batch pending on slice k, whose ev_mid has completed (its control block is in r.h_mid): is the round queue empty?  Then the
// connections (the batch's second part) are enqueued and the batch is no longer pending.  If not — a batch whose walks outlasted the expectation —
// the next rounds are enqueued with another copy of the control block behind them, and the batch waits for its next look.
static int finish_look(wtgpu_scene* s, size_t k, batch_launcher_t& L) {
    wtgpu_scene::pending_t& p = s->pending[k];
    chunk_rec_t& r = *p.rec;
    launch_args_t a;
    std::memcpy(&a, p.args, sizeof(a));
    hipStream_t st_ = s->streams[k];
    uint32_t& launched = p.launched;
    const bool light = !L.path_mode && s->knobs.light_rounds != 0;   // (a k_light_rounds launch stands behind the rounds enqueued so far)
    uint32_t stop = 0;
    if (light && launched < s->knobs.max_rounds) {
        launched += r.h_mid[CTL_LIGHT_DONE];
        stop = r.h_mid[CTL_LIGHT_STOP];
        s->light_rounds_run += r.h_mid[CTL_LIGHT_DONE];
    }
    bool done = false;
    static const bool diag = getenv("WTGPU_TAIL_DIAG") != nullptr;   // diagnostic: why the host was called back, printed at exit
    if (diag) {
        static unsigned long long looks = 0, by_stop[8] = {0}, light_done = 0, left_sum = 0;
        static bool reg = false;
        if (!reg) { reg = true; atexit([] { fprintf(stderr, "[tail diag] looks %llu (stop 0/1/2/3/4: %llu %llu %llu %llu %llu), light rounds %llu, walks left at looks (sum) %llu\n", looks, by_stop[0], by_stop[1], by_stop[2], by_stop[3], by_stop[4], light_done, left_sum); }); }
        looks++; by_stop[stop < 8 ? stop : 7]++; light_done += light ? r.h_mid[CTL_LIGHT_DONE] : 0;
        left_sum += r.h_mid[CTL_COUNT0 + (launched & 1u)] + r.h_mid[CTL_BACK0 + (launched & 1u)];
    }
    if (stop >= 1 && stop <= 3) {
        // a walk needs a stage the light kernel does not hold: the rest of THAT round by the ordinary kernels, then light again
        static const int from[4] = {0, 1, 3, 5};
        const int rc = L.rounds(a, s->d_path_slices[k], r, st_, launched, launched + 1, from[stop]);
        if (rc) return rc;
        launched += 1;
    } else {
        const uint32_t q = launched & 1u;
        const uint32_t left = r.h_mid[CTL_COUNT0 + q] + r.h_mid[CTL_BACK0 + q];
        if (left == 0) {
            note_rounds(s, std::min<uint32_t>(r.h_mid[CTL_ROUNDS], kMaxWalkIters));
            done = true;
        } else if (launched >= s->knobs.max_rounds)
            done = true;   // (WTGPU_MAX_ROUNDS: what is left is dropped and counted, drain_rec)
        else {
            s->round_fallbacks++;
            const uint32_t next = std::min<uint32_t>(s->knobs.max_rounds, launched + p.rounds_step);
            const int rc = L.rounds(a, s->d_path_slices[k], r, st_, launched, next);
            if (rc) return rc;
            launched = next;
            p.rounds_step = std::min<uint32_t>(64u, p.rounds_step * 2u);   // (a batch far beyond its expectation is not looked at every 8 rounds)
        }
    }
    if (done) {
        p.active = false;
        r.rounds_launched = launched;
        return L.tail(a, r, st_);
    }
    L.light(a, st_, launched);
    HIP_CHECK(hipMemcpyAsync(r.h_mid, a.st.ctl, CTL_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipEventRecord(r.ev_mid, st_));
    return WTGPU_OK;
}
// Serves the pending batches — whichever has its control block back gets its look (finish_look) — until the one on slice k (k = npos: every one)
// is finished.  The calling thread waits here for the GPU; it does not wait for ONE batch while another's stream stands idle behind a finished
// first part (round 6: a batch of bidir_room needs fifteen looks, a hundred light rounds apart, for the walks that restart thousands of times).
static int serve_pending(wtgpu_scene* s, size_t k, batch_launcher_t& L) {
    const size_t n = s->pending.size();
    for (;;) {
        bool any = false, progressed = false;
        for (size_t j = 0; j < n; ++j) {
            wtgpu_scene::pending_t& p = s->pending[j];
            if (!p.active) continue;
            if (k != (size_t)-1 && !s->pending[k].active) break;
            any = true;
            const hipError_t q = hipEventQuery(p.rec->ev_mid);
            if (q == hipSuccess) {
                const int rc = finish_look(s, j, L);
                if (rc) return rc;
                progressed = true;
            } else if (q != hipErrorNotReady)
                HIP_CHECK(q);
            else
                (void)hipGetLastError();   // (not ready is not an error: nothing of it may reach the next hipGetLastError check)
        }
        if (k != (size_t)-1 ? !s->pending[k].active : !any) return WTGPU_OK;
        if (!progressed) {
            std::this_thread::sleep_for(std::chrono::microseconds(20));   // (0 / 5 / 20 / 100 us measured alike: the looks, not the polling, are the tail)
        }
    }
}
// the second part of the batch pending on slice k (see batch_launcher_t); blocks the calling thread until its first part has run
static int render_finish_part(wtgpu_scene* s, size_t k, batch_launcher_t& L) {
    if (!s->pending[k].active) return WTGPU_OK;
    return serve_pending(s, k, L);
}
static int finish_all_pending(wtgpu_scene* s) {
    bool any = false;
    for (const auto& p : s->pending) any = any || p.active;
    if (!any) return WTGPU_OK;
    batch_launcher_t L(s);
    return serve_pending(s, (size_t)-1, L);
}
static int drain_all(wtgpu_scene* s) {
    {
        const int rc = finish_all_pending(s);
        if (rc) return rc;
    }
    for (auto& r : s->recs) {
        const int rc = drain_rec(s, r);
        if (rc) return rc;
    }
    return WTGPU_OK;
}

int wtgpu_join(wtgpu_scene* s, void* stream_) {
    if (!s || !s->uploaded) return fail(WTGPU_ERR_INVALID, "scene not uploaded");
    hipStream_t caller = static_cast<hipStream_t>(stream_);
    device_guard_t guard(s->device);
    {   // (blocks until the first parts of the pending batches have run: their second parts are enqueued here)
        const int rc = finish_all_pending(s);
        if (rc) return rc;
    }
    for (size_t k = 0; k < s->slices.size(); ++k) {
        HIP_CHECK(hipEventRecord(s->ev_done[k], s->streams[k]));
        HIP_CHECK(hipStreamWaitEvent(caller, s->ev_done[k], 0));
    }
    return WTGPU_OK;
}

int wtgpu_render(wtgpu_scene* s, void* stream_, double* d_value, double* d_weight, double* d_light, uint64_t sb, uint64_t se, uint64_t seed) {
    const int rc = wtgpu_render_async(s, stream_, d_value, d_weight, d_light, sb, se, seed);
    return rc ? rc : wtgpu_join(s, stream_);
}

int wtgpu_render_async(wtgpu_scene* s, void* stream_, double* d_value, double* d_weight, double* d_light, uint64_t sb, uint64_t se, uint64_t seed) {
    if (!s || !s->uploaded) return fail(WTGPU_ERR_INVALID, "scene not uploaded");
    if (!d_value || !d_weight || !d_light || se < sb) return fail(WTGPU_ERR_INVALID, "bad film pointers / sample range");
    hipStream_t caller = static_cast<hipStream_t>(stream_);
    device_guard_t guard(s->device);
    const scene_t& h = s->host;
    const uint64_t npix = (uint64_t)h.sensor.width * h.sensor.height;
    const uint64_t total = npix * (se - sb);
    if (total == 0) return WTGPU_OK;
    launch_args_t a;
    static_assert(sizeof(launch_args_t) <= 984, "by-value kernel arguments beyond 1 KB serialise the streams (see path_state_t)");
    a.sc = s->dev;
    a.film = film_t{d_value, d_weight, d_light, h.sensor.width, h.sensor.height, h.sensor.channels};
    a.seed = seed;
    a.npix = (uint32_t)npix;
    a.sample_begin = sb;
    const wtgpu_scene::knobs_t& K = s->knobs;   // environment knobs, read once at upload
    a.count_stats = K.count_stats;
    a.cone_budget = K.cone_budget;
    a.profile = K.profile;
    a.coop_aperture_min = K.coop_aperture_min;
    a.heavy_probe = K.heavy_probe;
    a.split_queues = K.split_queues;
    a.lane_cache = K.lane_cache;
    a.heavy_cache = K.heavy_cache;
    a.flux_task_tris = K.flux_task_tris;
    // Bounded triangle lists (64) are the fast path of an interaction region; a region that overflows its list is handled exactly by
    // walks of the WHOLE region: primary triangle (resolve_primary), classified edges (k_edges), intercepted power (k_flux_*).
    // WTGPU_NO_LISTS=1 (plt_bdpt, diagnostic): no lists at all, every region is gathered.
    // bit 0: the cone queries keep the region's bounded triangle list; bit 1 (plt_bdpt, WTGPU_PRIMARY_AXIS=1): the triangle under the beam axis of EVERY
    // diffusive hit comes from the trace kernels' axis query (as it does for regions beyond the list), not from a scan of the list in pass A
    a.collect_list = ((h.opts.integrator != INTEGRATOR_BDPT || !K.no_lists) ? 1u : 0u) | ((h.opts.integrator == INTEGRATOR_BDPT && K.primary_axis) ? 2u : 0u);
    batch_launcher_t L(s);

    // the internal streams start after everything already enqueued on the caller's stream ...
    HIP_CHECK(hipEventRecord(s->ev_begin, caller));
    const size_t n_slices = s->slices.size();
    std::vector<char> used(n_slices, 0);
    const uint64_t cap = s->slices[0].cap;
    for (uint64_t j0 = 0; j0 < total; j0 += cap) {
        const size_t k = s->slice_next++ % n_slices;
        hipStream_t st_ = s->streams[k];
        {   // the batch that still holds this slice gets its second part first (the host waits for its first part here: by now the other
            // slices' batches have been enqueued behind it, so the GPU is not idle meanwhile)
            const int rc = render_finish_part(s, k, L);
            if (rc) return rc;
        }
        if (!used[k]) {
            HIP_CHECK(hipStreamWaitEvent(st_, s->ev_begin, 0));
            used[k] = 1;
        }
        chunk_rec_t& r = s->recs[s->rec_next];
        s->rec_next = (s->rec_next + 1) % s->recs.size();
        int rc = drain_rec(s, r);   // recycles the oldest record (blocks only when > recs.size() batches are in flight)
        if (rc) return rc;
        const uint32_t nb = (uint32_t)std::min<uint64_t>(cap, total - j0);
        // WTGPU_STAGGER_ROUND=r (diagnostic, default off): a batch starts when the previous one (on the previous stream) has finished its round r.
        // Measured on the headline workload with 4 streams: r = 0 / 3 / 6 / 10 / 16 -> 151 / 150 / 161 / 200 / 263 ms per pass: the first
        // rounds ARE most of a batch, holding the next batch back only idles the GPU.
        if (K.stagger_round > 0 && s->ev_stagger_last && n_slices > 1) HIP_CHECK(hipStreamWaitEvent(st_, s->ev_stagger_last, 0));
        a.st = s->slices[k];
        a.j0 = j0;
        a.nb = nb;
        const uint32_t r1 = expected_rounds(s);
        L.generate(a, r, st_);
        rc = L.rounds(a, s->d_path_slices[k], r, st_, 0, r1);
        if (rc) return rc;
        HIP_CHECK(hipGetLastError());
        L.light(a, st_, r1);   // (the batch's last walks — a handful that restart thousands of times in some scenes — without another host round trip)
        HIP_CHECK(hipMemcpyAsync(r.h_mid, a.st.ctl, CTL_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipEventRecord(r.ev_mid, st_));
        wtgpu_scene::pending_t& p = s->pending[k];
        std::memcpy(p.args, &a, sizeof(a));
        p.rec = &r;
        p.rounds_first = r1;
        p.launched = r1;
        p.rounds_step = 8;
        p.active = true;

    }
    L.report();
    // (wtgpu_join enqueues what is pending and makes the caller's stream continue after all of it)
    s->samples_rendered += total;
    return WTGPU_OK;
}

int wtgpu_last_render_timings(wtgpu_scene* s, float out[12]) {
    if (!s || !out || !s->uploaded) return fail(WTGPU_ERR_INVALID, "scene not uploaded");
    const int rc = drain_all(s);
    if (rc) return rc;
    for (int i = 0; i < 12; ++i) out[i] = (float)s->acc[i];
    out[11] = s->acc[6] > 0 ? (float)((double)s->rounds_launched_total / s->acc[6]) : (float)kMaxWalkIters;   // rounds launched per batch (mean)
    return WTGPU_OK;
}

int wtgpu_get_counters(wtgpu_scene* s, wtgpu_counters* out) {
    if (!s || !out || !s->uploaded) return fail(WTGPU_ERR_INVALID, "scene not uploaded");
    bdpt_counters_t c;
    {
        const int rc = drain_all(s);
        if (rc) return rc;
    }
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(&c, s->slices[0].counters, sizeof(c), hipMemcpyDeviceToHost));
    out->samples = s->samples_rendered;
    out->segments = c.segments;
    out->ray_queries = c.ray_queries;
    out->cone_queries = c.cone_queries;
    out->vertices = c.vertices;
    out->connections = c.connections;
    out->shadow_rays = c.shadow_rays;
    out->cone_tri_overflow = c.cone_tri_overflow;
    out->edge_overflow = c.edge_overflow;
    out->fsd_edge_overflow = c.fsd_edge_overflow;
    out->fsd_pool_overflow = c.fsd_pool_overflow;
    out->fsd_interactions = c.fsd_interactions;
    out->null_interactions = c.null_interactions;
    out->surface_interactions = c.surface_interactions;
    out->light_splats = c.light_splats;
    out->walk_iteration_cap_hits = s->cap_hits;
    {
        unsigned long long dropped = 0;   // (per scene since round 4: the slot behind the profile counters)
        HIP_CHECK(hipMemcpy(&dropped, s->slices[0].counters + kDroppedSlot, sizeof(dropped), hipMemcpyDeviceToHost));
        out->traversal_stack_dropped = dropped;
    }
#ifdef WTGPU_STEP_PROF
    {
        unsigned long long p[8];
        HIP_CHECK(hipMemcpy(p, s->slices[0].counters + kNumCounters, sizeof(p), hipMemcpyDeviceToHost));
        fprintf(stderr, "[wtgpu step prof] pass-B walks %llu (with aperture %llu); total Mticks: scan %.1f pre %.1f edges %.1f integrals+aperture %.1f sample+append %.1f continue %.1f\n", p[7], p[6],
                double(p[0]) * 1e-6, double(p[1]) * 1e-6, double(p[2]) * 1e-6, double(p[3]) * 1e-6, double(p[4]) * 1e-6, double(p[5]) * 1e-6);
    }
#endif
#ifdef WTGPU_COOP_PROF
    if (getenv("WTGPU_PROFILE")) {
        unsigned long long p[12];
        HIP_CHECK(hipMemcpy(p, s->slices[0].counters + kNumCounters, sizeof(p), hipMemcpyDeviceToHost));
        const double n = p[4] ? double(p[4]) : 1.;
        fprintf(stderr, "[coop prof] items %llu; per item ticks: A.pop %.0f A.node+test %.0f A.push %.0f B1.filter %.0f flush+phaseB %.0f; per item: phase-B entries %.1f, candidates %.0f, filter batches %.1f, exact batches %.1f\n", p[4], p[0] / n, p[1] / n,
                p[2] / n, p[7] / n, p[6] / n, p[10] / n, p[11] / n, p[8] / n, p[9] / n);
    }
#endif
    if (s->knobs.profile == 1) {
        unsigned long long p[8];
        HIP_CHECK(hipMemcpy(p, s->slices[0].counters + kNumCounters, sizeof(p), hipMemcpyDeviceToHost));
        fprintf(stderr, "[wtgpu profile] flux tasks: %llu, candidates %llu (max %llu per task), exact-tested %llu; k_edges: %llu walks, %llu edges\n", p[0], p[1], p[4], p[2], p[5], p[6]);
    }
    if (s->knobs.profile == 3) {
        unsigned long long p[kProfSlots];
        HIP_CHECK(hipMemcpy(p, s->slices[0].counters + kNumCounters, sizeof(p), hipMemcpyDeviceToHost));
        fprintf(stderr, "[wtgpu profile] pass C by aperture size (slice 0): bin=log2(segments) items tries/item kticks/item total-Mticks\n");
        for (int b = 0; b < 16; ++b)
            if (p[8 + b]) fprintf(stderr, "[wtgpu profile]   C %2d %8llu %10.1f %10.1f %10.1f   fetch+load %.1f commit %.1f kticks/item\n", b, p[8 + b], double(p[24 + b]) / p[8 + b], double(p[40 + b]) / p[8 + b] * 1e-3, double(p[40 + b]) * 1e-6, double(p[112 + b]) / p[8 + b] * 1e-3, double(p[96 + b]) / p[8 + b] * 1e-3);
        fprintf(stderr, "[wtgpu profile] pass B by gathered scene edges: bin items kticks/item total-Mticks\n");
        for (int b = 0; b < 16; ++b)
            if (p[56 + b]) fprintf(stderr, "[wtgpu profile]   B %2d %8llu %10.1f %10.1f\n", b, p[56 + b], double(p[72 + b]) / p[56 + b] * 1e-3, double(p[72 + b]) * 1e-6);
    }
#ifdef WTGPU_REFILL_PROF
    {
        unsigned long long p[16];
        HIP_CHECK(hipMemcpy(p, s->slices[0].counters + kNumCounters, sizeof(p), hipMemcpyDeviceToHost));
        const char* nm[6] = {"serve", "fetch", "store", "nodes", "leaf", "(ray in fetch)"};
        for (int i = 0; i < 6; ++i) fprintf(stderr, "[refill prof] %-16s %10.1f Mticks  lanes %.1f\n", nm[i], p[i] * 1e-6, p[i] ? double(p[8 + i]) / p[i] : 0.);
    }
#endif
#ifdef WTGPU_SM_PROF
    {
        unsigned long long p[64];
        HIP_CHECK(hipMemcpy(p, s->slices[0].counters + kNumCounters, sizeof(p), hipMemcpyDeviceToHost));
        const char* nm[7] = {"serve", "fetch", "aw_next", "store", "NODE", "TRI", "EXACT"};
        for (int i = 0; i < 7; ++i)
            fprintf(stderr, "[sm prof] %-8s %10.1f Mticks  %10.2f Msteps  %7.0f ticks/step  lanes %.1f\n", nm[i], p[32 + i] * 1e-6, p[48 + i] * 1e-6, p[48 + i] ? double(p[32 + i]) / p[48 + i] : 0.,
                    p[32 + i] ? double(p[40 + i]) / p[32 + i] : 0.);
    }
#endif
    if (s->knobs.profile == 2) {
        unsigned long long p[8];
        HIP_CHECK(hipMemcpy(p, s->slices[0].counters + kNumCounters, sizeof(p), hipMemcpyDeviceToHost));
        fprintf(stderr, "[wtgpu profile] heavy items %llu: clock ticks ray %llu probe %llu cone %llu total %llu (per item: ray %.0f probe %.0f cone %.0f total %.0f; cone+probe phase A %.0f phase B %.0f; phase-A steps %.1f entries %.1f)\n", p[4], p[0],
                p[1], p[2], p[3], p[4] ? double(p[0]) / p[4] : 0., p[4] ? double(p[1]) / p[4] : 0., p[4] ? double(p[2]) / p[4] : 0., p[4] ? double(p[3]) / p[4] : 0., p[4] ? double(p[5]) / p[4] : 0., p[4] ? double(p[6]) / p[4] : 0., p[4] ? double(p[7] & 0xffffffffull) / p[4] : 0., p[4] ? double(p[7] >> 32) / p[4] : 0.);
    }
    return WTGPU_OK;
}
int wtgpu_reset_counters(wtgpu_scene* s) {
    if (!s || !s->uploaded) return fail(WTGPU_ERR_INVALID, "scene not uploaded");
    {
        const int rc = drain_all(s);
        if (rc) return rc;
    }
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemset(s->slices[0].counters, 0, (kNumCounters + kProfSlots + 1) * sizeof(unsigned long long)));
    s->samples_rendered = 0;
    s->cap_hits = 0;
    for (double& v : s->acc) v = 0;
    s->rounds_launched_total = 0;
    return WTGPU_OK;
}

int wtgpu_trace_rays(wtgpu_scene* s, void* stream_, const float* d_rays, uint32_t n, float* d_dist, uint32_t* d_tuid, float* d_bary, uint32_t* d_front) {
    if (!s || !s->uploaded) return fail(WTGPU_ERR_INVALID, "scene not uploaded");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_trace_rays, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, s->dev, d_rays, n, d_dist, d_tuid, d_bary, d_front);
    HIP_CHECK(hipGetLastError());
    return WTGPU_OK;
}
int wtgpu_traverse_cones(wtgpu_scene* s, void* stream_, const float* d_cones, uint32_t n, uint32_t cap, float* d_dist, uint32_t* d_flags,
                         uint32_t* d_ntris, uint32_t* d_tris) {
    if (!s || !s->uploaded) return fail(WTGPU_ERR_INVALID, "scene not uploaded");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // scratch for the bounded lists (ids + distances), kept with the scene between calls
    const size_t need = (size_t)n * kMaxConeTris * 4 * 2;
    if (need > s->query_scratch_bytes) {
        device_guard_t guard(s->device);
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, need));
        s->dev_allocs.push_back(p);   // (the old block, if any, is released with the scene)
        s->query_scratch = static_cast<uint32_t*>(p);
        s->query_scratch_bytes = need;
    }
    hipLaunchKernelGGL(k_traverse_cones, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, s->dev, d_cones, n, cap, d_dist, d_flags, d_ntris,
                       d_tris, s->query_scratch);
    HIP_CHECK(hipGetLastError());
    return WTGPU_OK;
}

int wtgpu_query_regions(wtgpu_scene* s, void* stream_, const float* d_cones, uint32_t n, uint32_t edge_cap, float* d_dist, uint32_t* d_flags,
                        uint32_t* d_primary, uint32_t* d_ntris, uint32_t* d_nedges, uint32_t* d_edges, float* d_flux) {
    if (!s || !s->uploaded) return fail(WTGPU_ERR_INVALID, "scene not uploaded");
    if (n == 0) return WTGPU_OK;
    hipLaunchKernelGGL(k_query_regions, dim3(n), dim3(64), 0, static_cast<hipStream_t>(stream_), s->dev, d_cones, n, edge_cap, d_dist, d_flags, d_primary,
                       d_ntris, d_nedges, d_edges, d_flux, s->slices[0].counters + kDroppedSlot);
    HIP_CHECK(hipGetLastError());
    return WTGPU_OK;
}

int wtgpu_calibrate_copy(uint64_t n_dwords, int repeats) {
    uint32_t *in = nullptr, *out = nullptr;
    HIP_CHECK(hipMalloc((void**)&in, n_dwords * 4));
    HIP_CHECK(hipMalloc((void**)&out, n_dwords * 4));
    HIP_CHECK(hipMemset(in, 1, n_dwords * 4));
    for (int r = 0; r < repeats; ++r) hipLaunchKernelGGL(k_calib_copy, dim3(256 * 32), dim3(256), 0, 0, in, out, (size_t)n_dwords);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipFree(in));
    HIP_CHECK(hipFree(out));
    return WTGPU_OK;
}

int wtgpu_develop(const wtgpu_scene* s, const double* value, const double* weight, const double* light, uint64_t spe, float* out) {
    if (!s || !value || !weight || !light || !out) return fail(WTGPU_ERR_INVALID, "null argument");
    const sensor_t& sn = s->host.sensor;
    const double sl = spe > 0 ? 1.0 / double(spe) : 0.0;
    for (size_t p = 0; p < (size_t)sn.width * sn.height; ++p)
        for (uint32_t c = 0, P = film_planes(sn); c < P; ++c) {
            const double w = weight[p];
            const double v = w != 0 ? value[p * P + c] / w : 0.0;
            out[p * P + c] = (float)(v + light[p * P + c] * sl);
        }
    return WTGPU_OK;
}

static void release_device(wtgpu_scene* s) {
    if (s->device < 0) return;
    device_guard_t guard(s->device);
    (void)hipDeviceSynchronize();
    for (void* p : s->dev_allocs) (void)hipFree(p);
    s->dev_allocs.clear();
    for (auto& r : s->recs) {
        for (auto& e : r.ev)
            if (e) (void)hipEventDestroy(e);
        if (r.h_ctl) (void)hipHostFree(r.h_ctl);
        if (r.h_mid) (void)hipHostFree(r.h_mid);
        if (r.ev_mid) (void)hipEventDestroy(r.ev_mid);
        if (r.ev_stagger) (void)hipEventDestroy(r.ev_stagger);
    }
    s->recs.clear();
    s->pending.clear();
    for (auto& e : s->ev_done)
        if (e) (void)hipEventDestroy(e);
    s->ev_done.clear();
    s->ev_stagger_last = nullptr;
    if (s->ev_begin) (void)hipEventDestroy(s->ev_begin);
    s->ev_begin = nullptr;
    for (auto& st_ : s->streams)
        if (st_) (void)hipStreamDestroy(st_);
    s->streams.clear();
    s->slices.clear();
    s->d_path_slices.clear();
    s->d_tri_class = nullptr;
    s->pend_cap = s->n_chunks = 0;
    s->uploaded = false;
}

void wtgpu_scene_destroy(wtgpu_scene* s) {
    if (!s) return;
    release_device(s);
    delete s;
}

// ---- render-seam control surface --------------------------------------------------------------------------------------------
int wtgpu_cancel(wtgpu_scene* s) {
    if (!s) return fail(WTGPU_ERR_INVALID, "null scene");
    s->cancel.store(1, std::memory_order_relaxed);
    s->paused.store(0, std::memory_order_relaxed);   // cancel is a full reset: a pause that was in force does not hold up the NEXT render (pause itself is sticky)
    return WTGPU_OK;
}
int wtgpu_pause(wtgpu_scene* s) {
    if (!s) return fail(WTGPU_ERR_INVALID, "null scene");
    s->paused.store(1, std::memory_order_relaxed);
    return WTGPU_OK;
}
int wtgpu_resume(wtgpu_scene* s) {
    if (!s) return fail(WTGPU_ERR_INVALID, "null scene");
    s->paused.store(0, std::memory_order_relaxed);
    return WTGPU_OK;
}
int wtgpu_capture_intermediate(wtgpu_scene* s, wtgpu_capture_cb capture, void* user) {
    if (!s || !capture) return fail(WTGPU_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> l(s->capture_mutex);
    s->capture_cb = capture;
    s->capture_user = user;
    return WTGPU_OK;
}
int wtgpu_render_progressive(wtgpu_scene* s, void* stream_, double* d_value, double* d_weight, double* d_light, uint64_t sb, uint64_t se, uint64_t seed,
                             uint32_t chunk_spp, wtgpu_progress_cb progress, void* user, uint64_t* spe_done) {
    if (!s || !s->uploaded) return fail(WTGPU_ERR_INVALID, "scene not uploaded");
    if (se < sb) return fail(WTGPU_ERR_INVALID, "bad sample range");
    if (spe_done) *spe_done = 0;
    s->cancel.store(0, std::memory_order_relaxed);
    const uint64_t step = chunk_spp ? chunk_spp : 1;
    const uint64_t npix = (uint64_t)s->host.sensor.width * s->host.sensor.height;
    device_guard_t guard(s->device);
    // a pending `capture intermediate` at a chunk boundary: the stream is idle, the films hold the completed chunks
    auto serve_capture = [&](uint64_t done) {
        wtgpu_capture_cb cb = nullptr;
        void* cu = nullptr;
        {
            std::lock_guard<std::mutex> l(s->capture_mutex);
            cb = s->capture_cb;
            cu = s->capture_user;
            s->capture_cb = nullptr;
        }
        if (cb) cb(done, cu);
    };
    for (uint64_t b = sb; b < se; b += step) {
        const uint64_t e = std::min(se, b + step);
        const int rc = wtgpu_render(s, stream_, d_value, d_weight, d_light, b, e, seed);
        if (rc) return rc;
        HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream_)));
        if (spe_done) *spe_done = e - sb;
        const bool stop = progress && progress((e - sb) * npix, (se - sb) * npix, user) != 0;
        serve_capture(e - sb);
        // paused: nothing is launched until wtgpu_resume (or a cancel); captures are still served (the reference's capture needs the paused state)
        while (s->paused.load(std::memory_order_relaxed) && !s->cancel.load(std::memory_order_relaxed) && !stop && e < se) {
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
            serve_capture(e - sb);
        }
        if ((stop || s->cancel.load(std::memory_order_relaxed)) && e < se) return fail(WTGPU_CANCELLED, "render cancelled");
    }
    return WTGPU_OK;
}

// ---- multi-GPU film reduction (RCCL) ----------------------------------------------------------------------------------------
#define NCCL_CHECK(x)                                                                                             \
    do {                                                                                                         \
        ncclResult_t r_ = (x);                                                                                   \
        if (r_ != ncclSuccess) return fail(WTGPU_ERR_COMM, std::string(#x) + ": " + ncclGetErrorString(r_));      \
    } while (0)
static_assert(sizeof(ncclUniqueId) == WTGPU_COMM_ID_BYTES, "ncclUniqueId size");
int wtgpu_comm_unique_id(void* id_out) {
    if (!id_out) return fail(WTGPU_ERR_INVALID, "null argument");
    ncclUniqueId id;
    NCCL_CHECK(ncclGetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return WTGPU_OK;
}
int wtgpu_comm_create(int world, int rank, int device, const void* id_, wtgpu_comm** out) {
    if (!id_ || !out || world < 1 || rank < 0 || rank >= world) return fail(WTGPU_ERR_INVALID, "bad communicator arguments");
    device_guard_t guard(device);
    auto c = std::make_unique<wtgpu_comm>();
    c->device = device;
    c->world = world;
    c->rank = rank;
    ncclUniqueId id;
    std::memcpy(&id, id_, sizeof(id));
    NCCL_CHECK(ncclCommInitRank(&c->comm, world, id, rank));
    *out = c.release();
    return WTGPU_OK;
}
int wtgpu_film_reduce(wtgpu_comm* c, void* stream_, double* d_value, double* d_weight, double* d_light, uint64_t n_value, uint64_t n_weight, int root) {
    if (!c || !c->comm || !d_value || !d_weight || !d_light || root < 0 || root >= c->world) return fail(WTGPU_ERR_INVALID, "bad reduce arguments");
    device_guard_t guard(c->device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    // one group: the three planes travel together (cornell 1440^2: 116 MB per rank, ~1.5 ms on a ring over xGMI)
    NCCL_CHECK(ncclGroupStart());
    NCCL_CHECK(ncclReduce(d_value, d_value, (size_t)n_value, ncclDouble, ncclSum, root, c->comm, st));
    NCCL_CHECK(ncclReduce(d_weight, d_weight, (size_t)n_weight, ncclDouble, ncclSum, root, c->comm, st));
    NCCL_CHECK(ncclReduce(d_light, d_light, (size_t)n_value, ncclDouble, ncclSum, root, c->comm, st));
    NCCL_CHECK(ncclGroupEnd());
    return WTGPU_OK;
}
void wtgpu_comm_destroy(wtgpu_comm* c) {
    if (!c) return;
    if (c->comm) {
        device_guard_t guard(c->device);
        (void)ncclCommDestroy(c->comm);
    }
    delete c;
}

}   // extern "C"
