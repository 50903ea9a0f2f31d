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
// All per-walk / per-sample state lives in HBM as word-interleaved SoA (wt::soa_load/soa_store) so that the 64
// lanes of a wavefront touch 64 consecutive dwords per field.
//
// There is no CPU fallback in this file: every entry point that computes requires a HIP device.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/wtgpu.h"
#include "host/scene_builder.h"
#include "wt/bdpt.h"
#include "wt/coop.h"

using namespace wt;

namespace {

constexpr uint32_t kMaxWalkIters = 96;   // must match oracle/oracle.cpp
constexpr int kBlock = 128;
constexpr int kLdsStack = 20;            // LDS-resident stack entries per lane
constexpr uint32_t kConeBudget = 96;     // cone-triangle tests one lane may spend on a query before it is handed to a wavefront
constexpr int kSpillStack = 44;          // scratch spill entries per lane (total 64, the reference's ray stack size)

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

struct device_state_t {
    uint64_t cap = 0;   // samples per batch
    uint32_t max_verts = 0;
    uint32_t* walks = nullptr;    // [kWalkWords][2cap]
    uint32_t* verts = nullptr;    // [max_verts*kVertexWords][2cap]
    uint32_t* ctx = nullptr;      // [kCtxWords][cap]
    uint32_t* trav = nullptr;     // [kTravWords][2cap]
    uint32_t* tris = nullptr;     // [kMaxConeTris][2cap]
    uint32_t* queue[2] = {nullptr, nullptr};
    uint32_t* qcount = nullptr;   // [2] device
    uint32_t* heavy_queue = nullptr;   // walks whose traversal exceeded the per-lane budget
    uint32_t* heavy_count = nullptr;   // [0] = number queued, [1] = dequeue head
    fsd_aperture_t* fsd_hdr = nullptr;
    fsd_edge_t* fsd_edges = nullptr;
    uint32_t* fsd_counter = nullptr;
    uint32_t fsd_cap = 0;
    unsigned long long* counters = nullptr;   // bdpt_counters_t + 2
    uint32_t* h_qcount = nullptr;             // pinned
    uint32_t* dbg = nullptr;                  // debug records (WTGPU_DEBUG_HEAVY builds)
};
constexpr size_t kWalkWords = sizeof(walk_t) / 4;
constexpr size_t kCtxWords = sizeof(sample_ctx_t) / 4;
constexpr size_t kTravWords = sizeof(trav_result_t) / 4;
constexpr size_t kNumCounters = sizeof(bdpt_counters_t) / sizeof(unsigned long long);

}   // namespace

struct wtgpu_scene {
    std::unique_ptr<wth::scene_builder_t> builder;   // owns the host arrays (named scenes)
    scene_t host{};                                  // host-pointer scene
    scene_t dev{};                                   // device-pointer scene
    std::vector<void*> dev_allocs;
    int device = -1;
    bool uploaded = false;
    device_state_t st;
    std::string stats;
    double lut_power[2] = {0, 0};
    float timings[8] = {0};
    uint64_t samples_rendered = 0;
    uint64_t cap_hits = 0;
    std::vector<hipEvent_t> events;
};

// ================================================ kernels ============================================================
namespace {

struct launch_args_t {
    scene_t sc;
    device_state_t st;
    film_t film;
    uint64_t seed;
    uint64_t j0;        // first global work item of this batch
    uint32_t nb;        // samples in this batch
    uint32_t npix;
    uint64_t sample_begin;
    uint32_t count_stats;
    uint32_t cone_budget;
};

__device__ inline void lds_stack(stack_entry_t* lds, stack_entry_t* spill, stack_ref_t& s) {
    s.p = lds + threadIdx.x;
    s.stride = blockDim.x;
    s.n_fast = kLdsStack;
    s.q = spill;
    s.cap = kLdsStack + kSpillStack;
}

__device__ inline void flush_counters(unsigned long long* g, const bdpt_counters_t& c) {
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(&c);
#pragma unroll
    for (size_t i = 0; i < kNumCounters; ++i) {
        unsigned long long v = p[i];
        // wave reduction (64 lanes)
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&g[i], v);
    }
}

__global__ void __launch_bounds__(kBlock) k_generate(launch_args_t a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nb) return;
    const uint64_t j = a.j0 + i;
    const uint32_t pix = (uint32_t)(j % a.npix);
    const uint64_t s = a.sample_begin + j / a.npix;
    const uint64_t sample_id = ((uint64_t)pix << 32) | (s & 0xFFFFFFFFull);
    const size_t W2 = 2 * (size_t)a.st.cap;
    sample_ctx_t ctx;
    walk_t sw, ew;
    const vertex_store_t svs{a.st.verts, W2, i}, evs{a.st.verts, W2, (size_t)a.st.cap + i};
    bdpt_generate(a.sc, a.seed, sample_id, pix % a.sc.sensor.width, pix / a.sc.sensor.width, ctx, sw, ew, svs, evs);
    soa_store(a.st.ctx, (size_t)a.st.cap, i, ctx);
    soa_store(a.st.walks, W2, i, sw);
    soa_store(a.st.walks, W2, (size_t)a.st.cap + i, ew);
}

// walk id -> (sample index, stream)
__device__ inline void walk_ident(const launch_args_t& a, uint32_t w, uint32_t& i, uint32_t& stream) {
    if (w < a.st.cap) {
        i = w;
        stream = STREAM_SENSOR_WALK;
    } else {
        i = w - (uint32_t)a.st.cap;
        stream = STREAM_EMITTER_WALK;
    }
}

__global__ void __launch_bounds__(kBlock) k_trace(launch_args_t a, const uint32_t* queue, uint32_t n, int first_round) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    const uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x;
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    if (qi < n) {
        const uint32_t w = first_round ? qi : queue[qi];
        const size_t W2 = 2 * (size_t)a.st.cap;
        walk_t wk;
        soa_load(a.st.walks, W2, w, wk);
        stack_entry_t spill[kSpillStack];
        stack_ref_t stack;
        lds_stack(lds, spill, stack);
        const uint_list_t tris{a.st.tris + w, (uint32_t)W2, kMaxConeTris};
        const cone_t env = walk_trace_envelope(a.sc, wk);
        const bool rt = a.sc.sensor.ray_trace_only || a.sc.opts.force_ray_tracing;
        const trav_result_t tr = traverse(a.sc, env, wavenum_to_wavelen_m(wk.beam.k), WT_INF, rt, stack, tris, nullptr, a.cone_budget, true);
        if (tr.aborted) {
            a.st.heavy_queue[atomicAdd(a.st.heavy_count, 1u)] = w;
        } else {
            soa_store(a.st.trav, W2, w, tr);
            ctr.segments = 1;
            ctr.ray_queries = tr.n_ray_queries;
            ctr.cone_queries = tr.n_cone_queries;
            ctr.cone_tri_overflow = tr.overflow;
        }
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}

// Heavy traversals: one wavefront (64-thread block) per walk, persistent blocks pulling from the heavy queue.
__global__ void __launch_bounds__(64) k_trace_heavy(launch_args_t a) {
    __shared__ coop_shared_t sh;
    __shared__ stack_entry_t lds[kLdsStack * 64];
    __shared__ uint32_t s_item;
    const uint32_t n = a.st.heavy_count[0];
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(a.st.heavy_count + 1, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        __syncthreads();
        if (item >= n) break;
        const uint32_t w = a.st.heavy_queue[item];
        const size_t W2 = 2 * (size_t)a.st.cap;
        walk_t wk;
        soa_load(a.st.walks, W2, w, wk);   // uniform address: broadcast
        stack_entry_t spill[kSpillStack];
        stack_ref_t stack;
        lds_stack(lds, spill, stack);
        const uint_list_t tris{a.st.tris + w, (uint32_t)W2, kMaxConeTris};
        const cone_t env = walk_trace_envelope(a.sc, wk);
        const bool rt = a.sc.sensor.ray_trace_only || a.sc.opts.force_ray_tracing;
#ifdef WTGPU_DEBUG_HEAVY
        const long long t0 = wall_clock64();
#endif
        const trav_result_t tr = coop_traverse(a.sc, env, wavenum_to_wavelen_m(wk.beam.k), WT_INF, rt, stack, sh, tris);
#ifdef WTGPU_DEBUG_HEAVY
        const long long dt = wall_clock64() - t0;   // 100 MHz ticks
        if (threadIdx.x == 0 && a.st.dbg) {
            const uint32_t k = atomicAdd(a.st.dbg, 1u);
            if (k < (1u << 20)) {
                uint32_t* r = a.st.dbg + 4 + 8 * (size_t)k;
                r[0] = w;
                r[1] = (uint32_t)dt;
                r[2] = tr.n_ray_queries | (tr.n_cone_queries << 8) | (tr.empty << 16) | (tr.ballistic << 17);
                r[3] = tr.ntris + tr.overflow;
                r[4] = __float_as_uint(env.x0);
                r[5] = __float_as_uint(tr.dist);
                r[6] = __float_as_uint(env.d.z);
                r[7] = __float_as_uint(env.o.z);
            }
        }
#endif
        if (threadIdx.x == 0) {
            soa_store(a.st.trav, W2, w, tr);
            ctr.segments += 1;
            ctr.ray_queries += tr.n_ray_queries;
            ctr.cone_queries += tr.n_cone_queries;
            ctr.cone_tri_overflow += tr.overflow;
        }
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}

__global__ void __launch_bounds__(kBlock) k_interact(launch_args_t a, const uint32_t* queue, uint32_t n, int first_round, uint32_t* next_queue,
                                                     uint32_t* next_count) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    const uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x;
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    if (qi < n) {
        const uint32_t w = first_round ? qi : queue[qi];
        const size_t W2 = 2 * (size_t)a.st.cap;
        uint32_t i, stream;
        walk_ident(a, w, i, stream);
        const uint64_t j = a.j0 + i;
        const uint32_t pix = (uint32_t)(j % a.npix);
        const uint64_t s = a.sample_begin + j / a.npix;
        const uint64_t sample_id = ((uint64_t)pix << 32) | (s & 0xFFFFFFFFull);
        walk_t wk;
        soa_load(a.st.walks, W2, w, wk);
        trav_result_t tr;
        soa_load(a.st.trav, W2, w, tr);
        const uint_list_t tris{a.st.tris + w, (uint32_t)W2, kMaxConeTris};
        const vertex_store_t vs{a.st.verts, W2, w};
        const fsd_pool_t pool{a.st.fsd_hdr, a.st.fsd_edges, a.st.fsd_counter, a.st.fsd_cap};
        stack_entry_t spill[kSpillStack];
        stack_ref_t stack;
        lds_stack(lds, spill, stack);
        const bool cont = bdpt_walk_step(a.sc, wk, tr, tris, vs, pool, a.seed, sample_id, stream, &ctr, &stack);
        wk.active = cont ? 1u : 0u;
        soa_store(a.st.walks, W2, w, wk);
        if (cont) next_queue[atomicAdd(next_count, 1u)] = w;
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}

__global__ void __launch_bounds__(kBlock) k_connect(launch_args_t a) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    if (i < a.nb) {
        const size_t W2 = 2 * (size_t)a.st.cap;
        const uint64_t j = a.j0 + i;
        const uint32_t pix = (uint32_t)(j % a.npix);
        const uint64_t s = a.sample_begin + j / a.npix;
        const uint64_t sample_id = ((uint64_t)pix << 32) | (s & 0xFFFFFFFFull);
        sample_ctx_t ctx;
        soa_load(a.st.ctx, (size_t)a.st.cap, i, ctx);
        const vertex_store_t svs{a.st.verts, W2, i}, evs{a.st.verts, W2, (size_t)a.st.cap + i};
        const uint32_t nT = a.st.walks[WT_WALK_NVERTS_WORD * W2 + i];
        const uint32_t nS = a.st.walks[WT_WALK_NVERTS_WORD * W2 + a.st.cap + i];
        stack_entry_t spill[kSpillStack];
        stack_ref_t stack;
        lds_stack(lds, spill, stack);
        const fsd_pool_t pool{a.st.fsd_hdr, a.st.fsd_edges, a.st.fsd_counter, a.st.fsd_cap};
#ifdef WTGPU_DEBUG_PRINT
        if (i == 0)
            printf("dbg k=%g recp=%g kd=%g el=(%u,%u) nT=%u nS=%u resp=%g %g %g spec0 type %d kmin %g kmax %g off %u cnt %u\n", ctx.k, ctx.recp_spectral_pd,
                   ctx.k_density, ctx.element.x, ctx.element.y, nT, nS, spectrum_f(a.sc, a.sc.sensor.response_spec[0], ctx.k),
                   spectrum_f(a.sc, a.sc.sensor.response_spec[1], ctx.k), spectrum_f(a.sc, a.sc.sensor.response_spec[2], ctx.k),
                   a.sc.spectra[a.sc.sensor.response_spec[0]].type, a.sc.spectra[a.sc.sensor.response_spec[0]].kmin,
                   a.sc.spectra[a.sc.sensor.response_spec[0]].kmax, a.sc.spectra[a.sc.sensor.response_spec[0]].offset,
                   a.sc.spectra[a.sc.sensor.response_spec[0]].count);
#endif
        bdpt_connect_all(a.sc, pool, a.film, svs, evs, (int)nT, (int)nS, ctx, a.seed, sample_id, stack, &ctr, nullptr);
#ifdef WTGPU_DEBUG_PRINT
        if (i == 0) {
            connect_ret_t cr;
            bdpt_connect(a.sc, pool, svs, evs, 0, 2, a.seed, sample_id, stack, cr, nullptr, nullptr);
            vertex_t last;
            svs.load(1, last);
            printf("dbg2 L02=%g type %u emitter_of_shape %d beam scale %g rad0 %g k %g rr %g film.value[0]=%g weight[0]=%g ptrs %p %p %p\n", cr.L.s[0], last.type,
                   last.emitter_of_shape, last.beam.scale, last.beam.rad[0], last.beam.k, last.rr_weight, a.film.value[0], a.film.weight[0], a.film.value,
                   a.film.weight, a.film.light);
        }
#endif
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}

// ---- per-query kernels (traversal parity tests) --------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_trace_rays(scene_t sc, const float* rays, uint32_t n, float* dist, uint32_t* tuid, float* bary, uint32_t* front) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack);
    const float* r = rays + 8 * (size_t)i;
    ray_hit_t h;
    ads_intersect_ray(sc, vec3{r[0], r[1], r[2]}, vec3{r[3], r[4], r[5]}, range_t{r[6], r[7]}, stack, h);
    dist[i] = h.dist;
    tuid[i] = h.tuid;
    bary[2 * i] = h.bx;
    bary[2 * i + 1] = h.by;
    front[i] = h.front_face;
}
__global__ void __launch_bounds__(kBlock) k_traverse_cones(scene_t sc, const float* cones, uint32_t n, uint32_t cap, float* dist, uint32_t* flags,
                                                           uint32_t* ntris, uint32_t* out_tris, uint32_t* scratch_tris) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack);
    const float* c = cones + 10 * (size_t)i;
    const vec3 d = normalize(vec3{c[3], c[4], c[5]});
    const cone_t env = make_cone(vec3{c[0], c[1], c[2]}, d, build_orthogonal_frame(d).t, c[6], c[8], c[7]);
    const uint_list_t tris{scratch_tris + i, n, kMaxConeTris};
    const trav_result_t tr = traverse(sc, env, c[9], WT_INF, false, stack, tris);
    dist[i] = tr.dist;
    flags[i] = (tr.empty ? 1u : 0u) | (tr.ballistic ? 2u : 0u) | (tr.front_face ? 4u : 0u);
    ntris[i] = tr.ballistic ? (tr.empty ? 0 : 1) : tr.ntris;
    for (uint32_t j = 0; j < cap; ++j) out_tris[(size_t)i * cap + j] = kInvalid;
    if (tr.ballistic) {
        if (!tr.empty) out_tris[(size_t)i * cap] = tr.tuid;
    } else {
        // insertion sort of the (short) list into the output
        uint32_t m = 0;
        for (uint32_t j = 0; j < tr.ntris; ++j) {
            const uint32_t v = tris[j];
            uint32_t pos = m < cap ? m : cap;
            while (pos > 0 && out_tris[(size_t)i * cap + pos - 1] > v) {
                if (pos < cap) out_tris[(size_t)i * cap + pos] = out_tris[(size_t)i * cap + pos - 1];
                --pos;
            }
            if (pos < cap) out_tris[(size_t)i * cap + pos] = v;
            if (m < cap) ++m;
        }
    }
}

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

int wtgpu_scene_create_named(const char* name, const wtgpu_scene_params* params, wtgpu_scene** out) {
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
        p.debug_only_s = params->debug_only_s;
        p.debug_only_t = params->debug_only_t;
        p.crop_of = params->crop_of;
        if (!wth::build_named_scene(name, p, *s->builder)) return fail(WTGPU_ERR_INVALID, std::string("unknown scene ") + name);
        s->host = s->builder->scene();
        s->stats = s->builder->stats();
        s->lut_power[0] = s->builder->fsd_lut_power(0);
        s->lut_power[1] = s->builder->fsd_lut_power(1);
        if ((uint32_t)s->host.opts.max_depth + 2 > kMaxVerts) return fail(WTGPU_ERR_INVALID, "max_depth exceeds the compiled vertex capacity (16)");
        *out = s.release();
        return WTGPU_OK;
    } catch (const std::exception& e) {
        return fail(WTGPU_ERR_INVALID, e.what());
    }
}

int wtgpu_scene_create_from_desc(const void* desc, wtgpu_scene** out) {
    if (!desc || !out) return fail(WTGPU_ERR_INVALID, "null argument");
    auto s = std::make_unique<wtgpu_scene>();
    s->host = *static_cast<const scene_t*>(desc);
    if ((uint32_t)s->host.opts.max_depth + 2 > kMaxVerts) return fail(WTGPU_ERR_INVALID, "max_depth exceeds the compiled vertex capacity (16)");
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

const void* wtgpu_scene_host_desc(const wtgpu_scene* s) { return s ? &s->host : nullptr; }
const char* wtgpu_scene_stats_json(const wtgpu_scene* s) { return s ? s->stats.c_str() : "{}"; }

int wtgpu_scene_upload(wtgpu_scene* s, int device, uint64_t max_batch) {
    if (!s) return fail(WTGPU_ERR_INVALID, "null scene");
    if (s->uploaded) return fail(WTGPU_ERR_INVALID, "scene already uploaded");
    int ndev = 0;
    const hipError_t dc = hipGetDeviceCount(&ndev);
    if (dc != hipSuccess || ndev == 0)
        return fail(WTGPU_ERR_NO_DEVICE, std::string("no HIP device present (there is no CPU fallback): hipGetDeviceCount -> ") + hipGetErrorString(dc) +
                                             ", count " + std::to_string(ndev));
    if (device < 0 || device >= ndev) return fail(WTGPU_ERR_NO_DEVICE, "invalid device index");
    HIP_CHECK(hipSetDevice(device));
    s->device = device;
    const scene_t& h = s->host;
    scene_t d = h;
    int rc;
#define UP(field, n) \
    if ((rc = upload(s, h.field, (size_t)(n), &d.field)) != WTGPU_OK) return rc;
    UP(tri_geo, h.n_tris)
    UP(tri_meta, h.n_tris)
    UP(tri_shade, h.n_tris)
    UP(edges, h.n_edges)
    UP(nodes, h.n_nodes)
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

    // per-batch path state
    device_state_t& st = s->st;
    const uint64_t npix = (uint64_t)h.sensor.width * h.sensor.height;
    st.cap = max_batch ? max_batch : std::min<uint64_t>(npix, 1u << 20);
    st.max_verts = (uint32_t)h.opts.max_depth + 2;
    const size_t W2 = 2 * (size_t)st.cap;
    if ((rc = dmalloc(s, &st.walks, kWalkWords * W2))) return rc;
    if ((rc = dmalloc(s, &st.verts, (size_t)st.max_verts * kVertexWords * W2))) return rc;
    if ((rc = dmalloc(s, &st.ctx, kCtxWords * (size_t)st.cap))) return rc;
    if ((rc = dmalloc(s, &st.trav, kTravWords * W2))) return rc;
    if ((rc = dmalloc(s, &st.tris, (size_t)kMaxConeTris * W2))) return rc;
    if ((rc = dmalloc(s, &st.queue[0], W2))) return rc;
    if ((rc = dmalloc(s, &st.queue[1], W2))) return rc;
    if ((rc = dmalloc(s, &st.qcount, 2))) return rc;
    if ((rc = dmalloc(s, &st.heavy_queue, W2))) return rc;
    if ((rc = dmalloc(s, &st.heavy_count, 2))) return rc;
    st.fsd_cap = (h.opts.FSD && !h.opts.force_ray_tracing) ? (uint32_t)std::min<uint64_t>(W2, 1u << 22) : 1u;
    if ((rc = dmalloc(s, &st.fsd_hdr, st.fsd_cap))) return rc;
    if ((rc = dmalloc(s, &st.fsd_edges, (size_t)st.fsd_cap * kFsdMaxEdges))) return rc;
    if ((rc = dmalloc(s, &st.fsd_counter, 1))) return rc;
    if ((rc = dmalloc(s, &st.counters, kNumCounters + 2))) return rc;
    HIP_CHECK(hipMemset(st.counters, 0, (kNumCounters + 2) * sizeof(unsigned long long)));
#ifdef WTGPU_DEBUG_HEAVY
    if ((rc = dmalloc(s, &st.dbg, 4 + 8 * (size_t)(1u << 20)))) return rc;
    HIP_CHECK(hipMemset(st.dbg, 0, 16));
#endif
    HIP_CHECK(hipHostMalloc((void**)&st.h_qcount, 2 * sizeof(uint32_t), hipHostMallocDefault));
    s->events.resize(4 * kMaxWalkIters + 8);
    for (auto& e : s->events) HIP_CHECK(hipEventCreate(&e));
    s->uploaded = true;
    return WTGPU_OK;
}

int wtgpu_render(wtgpu_scene* s, void* stream_, double* d_value, double* d_weight, double* d_light, uint64_t sb, uint64_t se, uint64_t seed) {
    if (!s || !s->uploaded) return fail(WTGPU_ERR_INVALID, "scene not uploaded");
    if (!d_value || !d_weight || !d_light || se < sb) return fail(WTGPU_ERR_INVALID, "bad film pointers / sample range");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_CHECK(hipSetDevice(s->device));
    const scene_t& h = s->host;
    device_state_t& st = s->st;
    const uint64_t npix = (uint64_t)h.sensor.width * h.sensor.height;
    const uint64_t total = npix * (se - sb);
    launch_args_t a;
    static_assert(sizeof(launch_args_t) <= 4096, "kernel argument too large");
    a.sc = s->dev;
    a.st = st;
    a.film = film_t{d_value, d_weight, d_light, h.sensor.width, h.sensor.height, h.sensor.channels};
    a.seed = seed;
    a.npix = (uint32_t)npix;
    a.sample_begin = sb;
    a.count_stats = 1;
    a.cone_budget = kConeBudget;
    if (const char* e = getenv("WTGPU_CONE_BUDGET")) a.cone_budget = (uint32_t)atoi(e);
    if (const char* e = getenv("WTGPU_COUNT_STATS")) a.count_stats = (uint32_t)atoi(e);
    float t_gen = 0, t_trace = 0, t_inter = 0, t_conn = 0, t_heavy = 0;
    uint32_t rounds_total = 0, n_trace_launches = 0;
    for (uint64_t j0 = 0; j0 < total; j0 += st.cap) {
        const uint32_t nb = (uint32_t)std::min<uint64_t>(st.cap, total - j0);
        a.j0 = j0;
        a.nb = nb;
        size_t ev = 0;
        auto rec = [&](void) { hipEventRecord(s->events[ev++], stream); };
        HIP_CHECK(hipMemsetAsync(st.fsd_counter, 0, sizeof(uint32_t), stream));
        rec();
        hipLaunchKernelGGL(k_generate, dim3((nb + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, a);
        HIP_CHECK(hipGetLastError());
        rec();
        // the first round's queue is the identity over both halves [0,nb) and [cap,cap+nb): materialise it only if nb<cap
        uint32_t n_active = 2 * nb;
        int cur = 0;
        bool first = (nb == st.cap);
        if (!first) {
            std::vector<uint32_t> q(2 * (size_t)nb);
            for (uint32_t i = 0; i < nb; ++i) {
                q[i] = i;
                q[nb + i] = (uint32_t)st.cap + i;
            }
            HIP_CHECK(hipMemcpyAsync(st.queue[0], q.data(), q.size() * 4, hipMemcpyHostToDevice, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
        }
        uint32_t round = 0;
        for (; round < kMaxWalkIters && n_active > 0; ++round) {
            HIP_CHECK(hipMemsetAsync(st.qcount + (1 - cur), 0, sizeof(uint32_t), stream));
            HIP_CHECK(hipMemsetAsync(st.heavy_count, 0, 2 * sizeof(uint32_t), stream));
            const dim3 grid((n_active + kBlock - 1) / kBlock);
            rec();
            hipLaunchKernelGGL(k_trace, grid, dim3(kBlock), 0, stream, a, st.queue[cur], n_active, first ? 1 : 0);
            HIP_CHECK(hipGetLastError());
            rec();
            {
                const uint32_t hb = std::min<uint32_t>(n_active, 256u * 16u);
                hipLaunchKernelGGL(k_trace_heavy, dim3(hb), dim3(64), 0, stream, a);
                HIP_CHECK(hipGetLastError());
            }
            rec();
            hipLaunchKernelGGL(k_interact, grid, dim3(kBlock), 0, stream, a, st.queue[cur], n_active, first ? 1 : 0, st.queue[1 - cur], st.qcount + (1 - cur));
            HIP_CHECK(hipGetLastError());
            rec();
            HIP_CHECK(hipMemcpyAsync(st.h_qcount, st.qcount + (1 - cur), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            n_active = st.h_qcount[0];
            cur = 1 - cur;
            first = false;
        }
        s->cap_hits += n_active;
        rec();
        hipLaunchKernelGGL(k_connect, dim3((nb + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, a);
        rec();
        HIP_CHECK(hipStreamSynchronize(stream));
        HIP_CHECK(hipGetLastError());
        // timings
        float ms = 0;
        hipEventElapsedTime(&ms, s->events[0], s->events[1]);
        t_gen += ms;
        size_t e = 2;
        for (uint32_t r = 0; r < round; ++r) {
            hipEventElapsedTime(&ms, s->events[e], s->events[e + 1]);
            t_trace += ms;
            hipEventElapsedTime(&ms, s->events[e + 1], s->events[e + 2]);
            t_heavy += ms;
            hipEventElapsedTime(&ms, s->events[e + 2], s->events[e + 3]);
            t_inter += ms;
            e += 4;
            ++n_trace_launches;
        }
        hipEventElapsedTime(&ms, s->events[e], s->events[e + 1]);
        t_conn += ms;
        rounds_total += round;
        // fsd pool overflow check
        uint32_t used = 0;
        HIP_CHECK(hipMemcpy(&used, st.fsd_counter, 4, hipMemcpyDeviceToHost));
        (void)used;
    }
#ifdef WTGPU_DEBUG_HEAVY
    {
        std::vector<uint32_t> d(4 + 8 * (size_t)(1u << 20));
        HIP_CHECK(hipMemcpy(d.data(), st.dbg, d.size() * 4, hipMemcpyDeviceToHost));
        const uint32_t n = std::min<uint32_t>(d[0], 1u << 20);
        std::vector<uint32_t> idx(n);
        for (uint32_t i = 0; i < n; ++i) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return d[4 + 8 * x + 1] > d[4 + 8 * y + 1]; });
        double tot = 0;
        for (uint32_t i = 0; i < n; ++i) tot += d[4 + 8 * i + 1];
        fprintf(stderr, "heavy items %u, total ticks %.3g (= %.1f wave-ms)\n", d[0], tot, tot / 1e5);
        for (uint32_t q : {0u, n / 1000, n / 100, n / 10, n / 2}) if (q < n) fprintf(stderr, "  rank %u ticks %u\n", q, d[4 + 8 * idx[q] + 1]);
        for (uint32_t i = 0; i < std::min<uint32_t>(n, 25); ++i) {
            const uint32_t* r = &d[4 + 8 * idx[i]];
            float x0, dist, dz, oz;
            memcpy(&x0, r + 4, 4); memcpy(&dist, r + 5, 4); memcpy(&dz, r + 6, 4); memcpy(&oz, r + 7, 4);
            fprintf(stderr, "  w=%u ms=%.2f nray=%u ncone=%u empty=%u ball=%u hits=%u x0=%g dist=%g dz=%g oz=%g\n", r[0], r[1] / 1e5, r[2] & 255, (r[2] >> 8) & 255,
                    (r[2] >> 16) & 1, (r[2] >> 17) & 1, r[3], x0, dist, dz, oz);
        }
        HIP_CHECK(hipMemset(st.dbg, 0, 16));
    }
#endif
    s->samples_rendered += total;
    s->timings[0] = t_gen;
    s->timings[1] = t_trace;
    s->timings[2] = t_inter;
    s->timings[3] = t_conn;
    s->timings[4] = (float)rounds_total;
    s->timings[5] = (float)n_trace_launches;
    s->timings[6] = (float)((total + st.cap - 1) / st.cap);
    s->timings[7] = t_heavy;
    return WTGPU_OK;
}

int wtgpu_last_render_timings(const wtgpu_scene* s, float out[8]) {
    if (!s || !out) return fail(WTGPU_ERR_INVALID, "null argument");
    std::memcpy(out, s->timings, sizeof(s->timings));
    return WTGPU_OK;
}

int wtgpu_get_counters(wtgpu_scene* s, wtgpu_counters* out) {
    if (!s || !out || !s->uploaded) return fail(WTGPU_ERR_INVALID, "scene not uploaded");
    bdpt_counters_t c;
    HIP_CHECK(hipMemcpy(&c, s->st.counters, sizeof(c), hipMemcpyDeviceToHost));
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
    return WTGPU_OK;
}
int wtgpu_reset_counters(wtgpu_scene* s) {
    if (!s || !s->uploaded) return fail(WTGPU_ERR_INVALID, "scene not uploaded");
    HIP_CHECK(hipMemset(s->st.counters, 0, (kNumCounters + 2) * sizeof(unsigned long long)));
    s->samples_rendered = 0;
    s->cap_hits = 0;
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
    uint32_t* scratch = nullptr;
    HIP_CHECK(hipMalloc((void**)&scratch, (size_t)n * kMaxConeTris * 4));
    hipLaunchKernelGGL(k_traverse_cones, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, s->dev, d_cones, n, cap, d_dist, d_flags, d_ntris,
                       d_tris, scratch);
    hipError_t e = hipStreamSynchronize(stream);
    hipFree(scratch);
    HIP_CHECK(e);
    return WTGPU_OK;
}

int wtgpu_develop(const wtgpu_scene* s, const double* value, const double* weight, const double* light, uint64_t spe, float* out) {
    if (!s || !value || !weight || !light || !out) return fail(WTGPU_ERR_INVALID, "null argument");
    const sensor_t& sn = s->host.sensor;
    const double sl = spe > 0 ? 1.0 / double(spe) : 0.0;
    for (size_t p = 0; p < (size_t)sn.width * sn.height; ++p)
        for (uint32_t c = 0; c < sn.channels; ++c) {
            const double w = weight[p];
            const double v = w != 0 ? value[p * sn.channels + c] / w : 0.0;
            out[p * sn.channels + c] = (float)(v + light[p * sn.channels + c] * sl);
        }
    return WTGPU_OK;
}

void wtgpu_scene_destroy(wtgpu_scene* s) {
    if (!s) return;
    if (s->uploaded) {
        hipSetDevice(s->device);
        for (void* p : s->dev_allocs) hipFree(p);
        if (s->st.h_qcount) hipHostFree(s->st.h_qcount);
        for (auto& e : s->events) hipEventDestroy(e);
    }
    delete s;
}

}   // extern "C"
