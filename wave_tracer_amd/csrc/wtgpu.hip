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

#include "../../include/wtgpu.h"
#include "wtgpu_test_hooks.h"
#include "scene_abi_check.h"
#include "host/scene_builder.h"
#include "wt/bdpt.h"
#include "wt/coop.h"
#include "wt/coop_fsd.h"
#include "wt/path.h"

using namespace wt;

namespace {

// Register budgets (second __launch_bounds__ argument = minimum waves per SIMD => 512 / n unified VGPRs per lane).
#ifndef WTGPU_LB_TRACE
#define WTGPU_LB_TRACE 3
#endif
#ifndef WTGPU_LB_HEAVY
#define WTGPU_LB_HEAVY 2   // 215 VGPRs, no spills, no scratch frame: as fast as 3 waves with 93 spilled registers, 27 GB per pass less HBM traffic
#endif
#ifndef WTGPU_LB_INTERACT
#define WTGPU_LB_INTERACT 4
#endif
#ifndef WTGPU_LB_INTERACT_B
#define WTGPU_LB_INTERACT_B 3
#endif
#ifndef WTGPU_LB_INTERACT_C
#define WTGPU_LB_INTERACT_C 3
#endif
#ifndef WTGPU_LB_FLUX
#define WTGPU_LB_FLUX 3
#endif
#ifndef WTGPU_LB_CONNECT
#define WTGPU_LB_CONNECT 2   // 355 -> 105 spilled registers (the rest of its frame are the two vertices and beams of a connection)
#endif
constexpr uint32_t kFluxTaskTris = 2048;   // default size of a region-sum task (k_flux_split / k_flux_tasks)
constexpr int kBlock = 128;
#ifndef WTGPU_LDS_STACK
#define WTGPU_LDS_STACK 20
#endif
constexpr int kLdsStack = WTGPU_LDS_STACK;   // LDS-resident stack entries per lane
constexpr uint32_t kConeBudget = 64;     // work units (1 per cone-triangle test, 2 per node) one lane may spend on a cone query before it is handed to a wavefront (with lane refill, round 3: 32 / 48 / 64 / 96 / 128 -> 14.6 / 14.8 / 15.2 / 14.4 / 12.8 Msamples/s; round 2's kernel without refill: optimum 28-32)
constexpr int kSpillStack = 64 - kLdsStack;   // scratch spill entries per lane (total 64, the reference's ray stack size)

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

// control block of one state slice (device memory)
enum : uint32_t { CTL_STRAT_HEAD_OPEN = 25, CTL_UTD_COUNT0 = 26, CTL_UTD_COUNT1 = 27, CTL_FSDQ_COUNT0 = 28, CTL_FSDQ_COUNT1 = 29, CTL_FSDQ_HEAD = 30, CTL_NEEQ_COUNT = 31, CTL_NEEQ_HEAD = 32, CTL_COUNT0 = 0, CTL_COUNT1 = 1, CTL_HEAD_TRACE = 2, CTL_HEAD_INTERACT = 3, CTL_HEAVY_COUNT = 4, CTL_HEAVY_HEAD = 5, CTL_FSD_COUNTER = 6,
                  CTL_ROUNDS = 7, CTL_STRAT_HEAD = 8, CTL_INTB_COUNT = 9, CTL_INTB_HEAD = 10, CTL_GATHER_COUNT = 11, CTL_GATHER_HEAD = 12, CTL_INTC_COUNT = 13, CTL_INTC_HEAD = 14, CTL_FTASK_COUNT = 15, CTL_FTASK_HEAD = 16, CTL_FSPLIT_HEAD = 17, CTL_EPOOL_COUNT = 18, CTL_FSD_ECOUNTER = 19, CTL_INTD_COUNT = 20, CTL_INTD_HEAD = 21, CTL_BACK0 = 22, CTL_BACK1 = 23, CTL_WORDS = 40 };   // (CTL_BACK*: see queue_append)   // (CTL_GATHER_*: queue of k_edges)
constexpr uint32_t kTriListWords = 128;   // per-walk list storage: 64 triangle ids, or (after coop_gather) up to 96 edge ids
constexpr uint32_t kGatherMarker = 0xFFFFFFFEu;   // trav.tuid of a walk whose interaction region was gathered
// ... and whose Fraunhofer aperture k_edges built as well (pool slot in trav.by): with segments — the walk is already queued for pass
// C — or without (pass B commits the restart)
constexpr uint32_t kApertureMarker = 0xFFFFFFFDu, kNullApertureMarker = 0xFFFFFFFCu;
__host__ __device__ inline bool is_region_marker(uint32_t t) { return t == kGatherMarker || t == kApertureMarker; }
// connection strategies (s,t) are bucketed by (min(t, kKeyDim-1), min(s, kKeyDim-1)): one bucket per strategy up to 18 vertices per subpath; a
// bucket of the last row / column holds every longer strategy of its sample (an item of such a bucket loops over them, k_connect_strat)
// (kMaxVerts + 2: up to max_depth = 16 — 18 vertices per subpath — every strategy has its own bucket and k_connect_strat_open is not launched;
// launching it for nothing cost 35 % of a pass with four streams: a 256-register, 22-KB-LDS grid that waits for free CUs holds up the other
// streams' dispatches)
constexpr uint32_t kKeyDim = kMaxVerts + 2, kNumKeys = kKeyDim * kKeyDim;

struct device_state_t {
    uint64_t cap = 0;   // samples per batch
    uint32_t max_verts = 0;
    uint32_t walk_words = 0;   // words of one walk record (walk_t, or path_walk_t for plt_path scenes)
    size_t vert_words = 0;     // words of one walk's vertex array (max_verts x kVertexWords)
    uint32_t* walks = nullptr;    // [kWalkWords][2cap]
    uint32_t* verts = nullptr;    // [max_verts*kVertexWords][2cap]
    uint32_t* ctx = nullptr;      // [kCtxWords][cap]
    uint32_t* trav = nullptr;     // [kTravWords][2cap]
    uint32_t* tris = nullptr;     // [kMaxConeTris][2cap]
    uint32_t* queue[2] = {nullptr, nullptr};
    uint32_t* heavy_queue = nullptr;   // walks whose traversal exceeded the per-lane budget
    uint32_t* intb_queue = nullptr;    // walks whose interaction takes the expensive (no primary triangle) path
    uint32_t* gather_queue = nullptr;  // ... of those, the ones whose triangle list overflowed (coop_gather first)
    uint32_t* intc_queue = nullptr;    // ... and the ones that built a Fraunhofer aperture with edges (sampled in pass C)
    uint32_t* intd_queue = nullptr;    // ... of those, the ones whose rejection sampling outlasts kEasyTries tries (k_interact_c_hard)
    uint2* ftasks = nullptr;           // (walk, subtree) tasks of the intercepted-power sums of overflowed regions (k_flux_split / k_flux_tasks)
    uint32_t ftask_cap = 0;
    double* facc = nullptr;            // [2cap] their accumulators
    uint32_t* epool = nullptr;         // edge-id lists of the gathered regions of one round (bump allocator, k_edges)
    uint32_t epool_cap = 0;
    uint32_t* ctl = nullptr;           // [CTL_WORDS] queue sizes, dequeue heads, FSD pool bump counter, rounds done
    fsd_aperture_t* fsd_hdr = nullptr;
    fsd_edge_t* fsd_edges = nullptr;
    uint32_t fsd_cap = 0;
    uint32_t fsd_ecap = 0;            // segment records of all apertures of a batch (bump allocator)
    uint32_t* strat_items = nullptr;    // [kNumKeys][cap] sample indices bucketed by connection strategy (s,t)
    uint32_t* strat_count = nullptr;    // [kNumKeys]
    uint32_t* strat_prefix = nullptr;   // [kNumKeys + 1]
    double* lacc = nullptr;             // [4][cap] per-sample sum of the t>1 strategies' fluxes
    unsigned long long* counters = nullptr;   // bdpt_counters_t + 2 (shared by all slices)
};
// plt_path only — a device-resident block the path kernels get a pointer to (launch_args_t stays below 1024 bytes: by-value kernel
// arguments beyond that cost 40 % of a plt_bdpt pass with four streams, measured: 976 -> 1048 bytes, 15.4 -> 11.1 Msamples/s).
struct path_state_t {
    // plt_path: wedge records of the walks' UTD apertures, two pools used alternately (round parity: an aperture built in round r is evaluated in
    // round r + 1), each reset when its round begins; queues of the wave-per-walk UTD kernels and what they exchange with k_path_interact
    utd_edge_rec_t* utd[2] = {nullptr, nullptr};
    uint32_t utd_cap = 0;
    uint32_t* fsdq[2] = {nullptr, nullptr};   // walks that carry an aperture into the next round (k_path_fsd evaluates it there)
    uint32_t* neeq = nullptr;                  // walks with a deferred next-event estimation of this round (k_path_nee)
    float* fsd_f = nullptr;                    // [cap] k_path_fsd's result per walk
    path_nee_rec_t* nee_recs = nullptr;        // [cap]
    uint2* gather_info = nullptr;              // [cap] k_path_edges' result per walk: (offset into the round's edge pool, number of ids)
};
constexpr size_t kWalkWords = sizeof(walk_t) / 4;
constexpr size_t kCtxWords = sizeof(sample_ctx_t) / 4;
constexpr size_t kTravWords = sizeof(trav_result_t) / 4;
#define WT_TRAV_WORD(field) (offsetof(trav_result_t, field) / 4)
constexpr size_t kNumCounters = sizeof(bdpt_counters_t) / sizeof(unsigned long long);
constexpr size_t kProfSlots = 128;   // WTGPU_PROFILE scratch counters behind the public ones
constexpr size_t kDroppedSlot = kNumCounters + kProfSlots;   // ... and behind those: children a full cooperative traversal stack could not hold (wt/coop.h)

}   // namespace

struct chunk_rec_t {
    std::vector<hipEvent_t> ev;   // [0] start, [1] after generate, then 6 per LAUNCHED round, last used: after connect
    uint32_t* h_ctl = nullptr;    // pinned snapshot of the slice's control block after the batch
    uint32_t* h_mid = nullptr;    // ... and after the rounds launched up front (render_first_part): is the queue empty?
    hipEvent_t ev_mid = nullptr;
    hipEvent_t ev_stagger = nullptr;   // recorded after the batch's round `stagger_round`: the next batch (on the next stream) starts there
    uint32_t rounds_launched = 0;
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
    // tuning knobs (environment, read ONCE at upload: wtgpu_scene_upload)
    struct knobs_t {
        uint32_t cone_budget = 0, count_stats = 1, profile = 0, no_lists = 0, stagger_round = 0, lane_cache = 1, heavy_cache = 1, split_queues = 1;
        uint32_t shrink_r1 = 8, shrink_f1 = 4, shrink_r2 = 16, shrink_f2 = 32, shrink_h1 = 4, decay_q = 0, decay_c = 4;   // persistent-grid sizes of the later rounds (see wtgpu_render_async)
        uint32_t heavy_waves_per_cu = 8, round_blocks_per_cu = 8, grid_div_b = 4, grid_div_c = 1, grid_div_hard = 4, grid_mul_flux = 2, coop_aperture_min = 8, heavy_probe = 1, flux_task_tris = kFluxTaskTris;
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
    uint32_t flux_task_tris;   // k_flux_split: largest subtree handed to one wavefront of k_flux_tasks
    uint32_t heavy_probe;   // k_trace_heavy: any-hit probe of the near slab before the handed-over cone query too
    uint32_t coop_aperture_min;   // regions with at least this many classified edges get their aperture built by k_edges' wavefront
    uint32_t profile;   // WTGPU_PROFILE=1: clock64() breakdown of the heavy traversals into counters[kNumCounters..]
    uint32_t split_queues;   // round queues keep sensor and emitter walks apart (queue_append); 0: one mixed queue (A/B)
    uint32_t lane_cache, heavy_cache;   // diagnostic switches of the remembered rejecting triangles (wt::traverse_axis / coop_traverse); default on
    uint32_t collect_list;    // plt_path: the cone queries keep the bounded triangle list of the interaction region (plt_bdpt: closest hit only)
};

// (block size as a constant: blockDim would pull 256 bytes of hidden kernel arguments into the kernel-argument segment)
__device__ inline void lds_stack(stack_entry_t* lds, stack_entry_t* spill, stack_ref_t& s, uint32_t block = kBlock) {
    s = make_stack_ref(lds + threadIdx.x, block, kLdsStack + kSpillStack, kLdsStack, spill);
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
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) {
        uint32_t* ctl = a.st.ctl;
        ctl[CTL_COUNT0] = 2 * a.nb;
        ctl[CTL_COUNT1] = 0;
        ctl[CTL_BACK0] = ctl[CTL_BACK1] = 0;
        ctl[CTL_HEAD_TRACE] = ctl[CTL_HEAD_INTERACT] = ctl[CTL_HEAVY_COUNT] = ctl[CTL_HEAVY_HEAD] = ctl[CTL_FSD_COUNTER] = ctl[CTL_ROUNDS] = 0;
        ctl[CTL_INTB_COUNT] = ctl[CTL_INTB_HEAD] = ctl[CTL_GATHER_COUNT] = ctl[CTL_GATHER_HEAD] = ctl[CTL_INTC_COUNT] = ctl[CTL_INTC_HEAD] = 0;
        ctl[CTL_FTASK_COUNT] = ctl[CTL_FTASK_HEAD] = ctl[CTL_FSPLIT_HEAD] = ctl[CTL_EPOOL_COUNT] = ctl[CTL_FSD_ECOUNTER] = 0;
        ctl[CTL_INTD_COUNT] = ctl[CTL_INTD_HEAD] = 0;
    }
    if (i >= a.nb) return;
    const uint64_t j = a.j0 + i;
    const uint32_t pix = (uint32_t)(j % a.npix);
    const uint64_t s = a.sample_begin + j / a.npix;
    const uint64_t sample_id = ((uint64_t)pix << 32) | (s & 0xFFFFFFFFull);
    const size_t W2 = 2 * (size_t)a.st.cap;
    sample_ctx_t ctx;
    walk_t sw, ew;
    const vertex_store_t svs{a.st.verts, a.st.vert_words, i}, evs{a.st.verts, a.st.vert_words, (size_t)a.st.cap + i};
    bdpt_generate(a.sc, a.seed, sample_id, pix % a.sc.sensor.width, pix / a.sc.sensor.width, ctx, sw, ew, svs, evs);
    soa_store(a.st.ctx, kCtxWords, i, ctx);
    soa_store(a.st.walks, a.st.walk_words, i, sw);
    soa_store(a.st.walks, a.st.walk_words, (size_t)a.st.cap + i, ew);
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
// The round queues hold the two kinds of walks apart: sensor walks are appended from the front of the array (count CTL_COUNT*), emitter
// walks from its end backwards (count CTL_BACK*).  A traversal costs an emitter walk of the headline workload 5-10x what it costs a
// sensor walk (wide beams from the spots against pixel-sized beams from the camera): wavefronts that hold one kind waste fewer lanes.
// queue item -> walk id; the first round's queue is the identity over [0,nb) (sensor walks) and [cap,cap+nb) (emitter walks)
__device__ inline uint32_t queue_count(const uint32_t* ctl, int in) { return ctl[CTL_COUNT0 + in] + ctl[CTL_BACK0 + in]; }
__device__ inline uint32_t queue_walk(const launch_args_t& a, const uint32_t* ctl, int in, uint32_t qi, int first_round) {
    if (first_round) return qi < a.nb ? qi : (uint32_t)a.st.cap + (qi - a.nb);
    const uint32_t front = ctl[CTL_COUNT0 + in];
    return qi < front ? a.st.queue[in][qi] : a.st.queue[in][2 * (size_t)a.st.cap - 1 - (qi - front)];
}
// one wavefront grabs the next 64 queue items
__device__ inline uint32_t wave_grab(uint32_t* head) {
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(head, 64u);
    return (uint32_t)__shfl((int)base, 0, 64);
}
// wave-aggregated append of `w` (for lanes with `pred`) to a device queue
__device__ inline void wave_append(uint32_t* queue, uint32_t* count, bool pred, uint32_t w) {
    const unsigned long long m = __ballot(pred);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (pred) queue[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = w;
}

// ... of walk `w` (for lanes with `pred`) to round queue `out`: sensor walks (and plt_path's) at the front, emitter walks at the back
__device__ inline void queue_append(const launch_args_t& a, uint32_t* ctl, int out, bool pred, uint32_t w) {
    const bool back = pred && w >= a.st.cap && a.split_queues;
    wave_append(a.st.queue[out], ctl + CTL_COUNT0 + out, pred && !back, w);
    const unsigned long long m = __ballot(back);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(ctl + CTL_BACK0 + out, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (back) a.st.queue[out][2 * (size_t)a.st.cap - 1 - (base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)))] = w;
}

#ifndef WTGPU_LEAF_NUM
#define WTGPU_LEAF_NUM 1   // leaf step when at least NUM / DEN of the running lanes hold a leaf (swept 1/3, 1/2, 2/3, 3/4: 99.0 / 97.6 / 96.6 / 97.9 ms per pass, noise 1 ms)
#define WTGPU_LEAF_DEN 2
#endif
// The per-lane trace kernel, with LANE REFILL.
// The cost of a walk's traversal varies by two orders of magnitude — one to seven cone queries of 2..cone_budget work units each —
// and a wavefront whose lanes ran the policy and their queries back to back would be as slow as its slowest lane (rounds 1-2: that kernel
// was kept as an A/B reference until round 4).  Here a lane is a slot that walks pass through.  The wavefront alternates between
//   * the traversal loop: every lane that holds a node descends (cq_node_step), every lane that holds a leaf tests its triangles
//     (cq_leaf_step) — the steps of wt/bvh.h, which the CPU checker drives one query at a time —
//   * and the service section, entered once enough lanes wait: a lane whose query ended gets the policy's next query (aw_query_done /
//     aw_next) or stores its record, and lanes without a walk fetch new ones from the queue (one atomic per wavefront), trace the beam
//     axis and start their first query.
// A slow query therefore occupies one lane, not 64, which is also what lets the work budget per query be larger (fewer walks
// handed to the wave-cooperative kernel).  Per walk the sequence of visits and the results are those of wt::traverse_axis.
//
// (GUIDED FETCH — a wavefront holds at most ceil(walks left in the queue / wavefronts of the grid) walks, so that the end of a round is as long
// as its longest single walk instead of a wavefront's 64 — was built and measured in round 4, dynamically and as a per-round target: the short
// rounds of a one-stream pass went from 1.5 to 1.0 ms each, but a wavefront that fetches one walk at a time runs its fetch section — the axis
// query — for one lane: the long rounds got 35 % slower, the pass 9 % (20.4 vs 22.4 Msamples/s).  With the per-round target: -4 % on the
// headline workload (21.5 vs 22.5), +3..6 % on the 720 x 540 film, -3 % with two-pass batches.  Not kept: what the ends of the rounds cost is paid per BATCH,
// and larger batches (bench.py: ~4 M samples) removed most of it: 720 x 540 18.8 -> 56 Msamples/s.)
#ifndef WTGPU_REFILL_MIN
#define WTGPU_REFILL_MIN 16
#endif
// the policy up to its next cone query (TRUE) or its end (FALSE: `r` is final); the tests of the remembered triangles run right here
__device__ inline bool policy_next_query(const scene_t& sc, const cone_t& env, bool rt, const stack_ref_t& stack, axis_walk_t& aw, cone_query_t& q, trav_result_t& r) {
    for (;;) {
        const int need = aw_next(sc, env, rt, stack, aw, q, r);
        if (need != AW_TEST) return need == AW_QUERY;
        aw_test_done(aw, cone_attempt_too_short_by(sc, env, aw.cand, aw.sr, aw.min_df_prog));
    }
}
__global__ void __launch_bounds__(kBlock, WTGPU_LB_TRACE) k_trace_refill(launch_args_t a, int in, int first_round, uint32_t round) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = queue_count(ctl, in);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl[CTL_COUNT0 + (1 - in)] = 0;   // output queue of this round's k_interact
        ctl[CTL_BACK0 + (1 - in)] = 0;
        ctl[CTL_HEAD_INTERACT] = 0;
        ctl[CTL_INTB_COUNT] = 0;
        ctl[CTL_INTB_HEAD] = 0;
        ctl[CTL_GATHER_COUNT] = 0;
        ctl[CTL_GATHER_HEAD] = 0;
        ctl[CTL_INTC_COUNT] = 0;
        ctl[CTL_INTC_HEAD] = 0;
        ctl[CTL_FTASK_COUNT] = 0;
        ctl[CTL_FTASK_HEAD] = 0;
        ctl[CTL_FSPLIT_HEAD] = 0;
        ctl[CTL_EPOOL_COUNT] = 0;
        ctl[CTL_INTD_COUNT] = 0;
        ctl[CTL_INTD_HEAD] = 0;
        // plt_path: this round's wedge pool and the queue it fills for the next round's k_path_fsd; this round's k_path_fsd / k_path_nee heads
        ctl[CTL_UTD_COUNT0 + (round & 1u)] = 0;
        ctl[CTL_FSDQ_COUNT0 + ((round + 1u) & 1u)] = 0;
        ctl[CTL_FSDQ_HEAD] = 0;
        ctl[CTL_NEEQ_COUNT] = 0;
        ctl[CTL_NEEQ_HEAD] = 0;
        if (n > 0) ctl[CTL_ROUNDS] = round + 1;
    }
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack);
    const size_t W2 = 2 * (size_t)a.st.cap;
    const bool rt = a.sc.sensor.ray_trace_only || a.sc.opts.force_ray_tracing;
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    // lane state: 0 = no walk, 1 = cone query running, 2 = cone query ended (to be served)
    int st = 0;
    uint32_t w = 0;
    cone_t env;
    axis_walk_t aw;
    cone_query_t q;
    uint_list_t tris{nullptr, 1u, 0u, nullptr};
    memset(&env, 0, sizeof(env));
    memset(&aw, 0, sizeof(aw));
    memset(&q, 0, sizeof(q));
    bool exhausted = false;   // wave-uniform: the queue holds no more walks
#ifdef WTGPU_REFILL_PROF
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0}, pl[6] = {0, 0, 0, 0, 0, 0};
    long long pt;
#define RP_BEGIN() pt = clock64()
#define RP_END(i, mask) do { const long long d_ = clock64() - pt; pc[i] += (unsigned long long)d_; pl[i] += (unsigned long long)d_ * (unsigned long long)__popcll(mask); } while (0)
#else
#define RP_BEGIN()
#define RP_END(i, mask)
#endif
    for (;;) {
        // ---- service section
        bool fin = false;
        trav_result_t r;
        RP_BEGIN();
        const unsigned long long m_srv = __ballot(st == 2);
        if (st == 2) {
            cq_end(env, tris, q);
            fin = aw_query_done(a.sc, env, aw, q.rec, r);
            if (!fin) fin = !policy_next_query(a.sc, env, rt, stack, aw, q, r);
            st = fin ? 0 : 1;
        }
        RP_END(0, m_srv);
        // (records of finished walks are stored below, together with those of freshly fetched walks that need no cone query)
        uint32_t w_fin = w;
        const int n_idle = __popcll(__ballot(st == 0 && !fin)), n_run = __popcll(__ballot(st == 1));
        bool fetched = false;
        const bool any_fin = __ballot(fin) != 0;   // (their records are stored first; they fetch in the next turn)
        if (!exhausted && !any_fin && (n_idle >= WTGPU_REFILL_MIN || n_run == 0)) {
            const unsigned long long im = __ballot(st == 0);
            const bool take = st == 0;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(ctl + CTL_HEAD_TRACE, (uint32_t)__popcll(im));
            base = (uint32_t)__shfl((int)base, 0, 64);
            if (base + (uint32_t)__popcll(im) >= n) exhausted = true;
            const uint32_t qi = base + (uint32_t)__popcll(im & below);
            RP_BEGIN();
            const unsigned long long m_f = __ballot(take && qi < n);
            if (take && qi < n) {
                w = queue_walk(a, ctl, in, qi, first_round);
                w_fin = w;
                const walk_trace_in_t wk = walk_load_trace_in(a.st.walks, a.st.walk_words, w);
                // plt_bdpt: the bounded list (64 triangles + their cone-hit distances) of the interaction region; see k_trace
                uint32_t* slot = a.st.tris + (size_t)w * kTriListWords;
                tris = uint_list_t{slot, 1u, a.collect_list ? kMaxConeTris : 0u, reinterpret_cast<float*>(slot + kMaxConeTris)};
                env = walk_trace_envelope(a.sc, wk);
                ray_hit_t ah;
                // (The axis query in a kernel of its own was built twice: round 3 as a grid-stride kernel — 60 vs 56 ms per pass — and round 4 as a
                // lane-refill kernel like this one (k_trace_axis: 111 registers, 4 waves per SIMD, 2.2 G rays/s in the long rounds: 3.9 ms where this
                // section spends ~3): the two kernels together took 24.2 ms of the long rounds against 23.4 ms with the query in here, 22.4 vs 22.5
                // Msamples/s — the fetch section's rays overlap other wavefronts' cone queries, which a separate kernel gives up.  Not kept.)
                const bool axis_hit = ads_intersect_ray(a.sc, env.o, env.d, range_t{0.f, WT_INF}, stack, ah);
                aw_begin(aw, wavenum_to_wavelen_m(wk.k), WT_INF, axis_hit, ah, a.cone_budget, true, !a.collect_list, a.lane_cache ? wk.prev_offset_tuid : kInvalid);
                aw.use_cache = a.lane_cache;
                fin = !policy_next_query(a.sc, env, rt, stack, aw, q, r);
                st = fin ? 0 : 1;
            }
            RP_END(1, m_f);
            fetched = true;
        }
        // store the records of the walks that ended in this section (convergent: the queue append is a wave operation)
        RP_BEGIN();
        const unsigned long long m_st = __ballot(fin);
        {
            const bool heavy = fin && r.aborted == 1;
            if (fin) {
                if (heavy) {
                    // resume state for k_trace_heavy (aw_query_done: dist / ntris = distance / segment of the query, the axis hit, the last
                    // rejecting triangle in `overflow`)
                    a.st.trav[(size_t)w_fin * kTravWords + WT_TRAV_WORD(dist)] = __float_as_uint(r.dist);
                    a.st.trav[(size_t)w_fin * kTravWords + WT_TRAV_WORD(ntris)] = r.ntris;
                    a.st.trav[(size_t)w_fin * kTravWords + WT_TRAV_WORD(n_ray_queries)] = r.n_ray_queries;
                    a.st.trav[(size_t)w_fin * kTravWords + WT_TRAV_WORD(n_cone_queries)] = r.n_cone_queries;
                    a.st.trav[(size_t)w_fin * kTravWords + WT_TRAV_WORD(tuid)] = r.tuid;
                    a.st.trav[(size_t)w_fin * kTravWords + WT_TRAV_WORD(bx)] = __float_as_uint(r.bx);
                    a.st.trav[(size_t)w_fin * kTravWords + WT_TRAV_WORD(by)] = __float_as_uint(r.by);
                    a.st.trav[(size_t)w_fin * kTravWords + WT_TRAV_WORD(pdist)] = __float_as_uint(r.pdist);
                    a.st.trav[(size_t)w_fin * kTravWords + WT_TRAV_WORD(front_face)] = r.front_face;
                    a.st.trav[(size_t)w_fin * kTravWords + WT_TRAV_WORD(overflow)] = r.overflow;
                } else {
                    soa_store(a.st.trav, kTravWords, w_fin, r);
                    ctr.segments += 1;
                    ctr.ray_queries += r.n_ray_queries;
                    ctr.cone_queries += r.n_cone_queries;
                    if (a.collect_list) ctr.cone_tri_overflow += r.overflow;
                }
            }
            wave_append(a.st.heavy_queue, ctl + CTL_HEAVY_COUNT, heavy, w_fin);
        }
        RP_END(2, m_st);
        // walks that ended left their lanes free: fetch (more) before traversing
        if (fetched || any_fin) continue;
        const int running = __popcll(__ballot(st == 1));
        if (running == 0) {
            if (exhausted) break;
            continue;
        }
        // ---- traversal loop: until a quarter of the lanes that entered it (at most WTGPU_REFILL_MIN) wait to be served
        const int leave_at = running < 4 * WTGPU_REFILL_MIN ? (running + 3) / 4 : WTGPU_REFILL_MIN;
        for (;;) {
            // nodes: every lane that holds no leaf descends, until the lanes with a leaf are the majority
            for (;;) {
                const bool at_node = st == 1 && q.leaf == 0 && q.s > 0;
                const unsigned long long nm = __ballot(at_node);
                if (!nm) break;
                RP_BEGIN();
                if (at_node) cq_node_step(a.sc, env, stack, q);
                RP_END(3, nm);
                if (WTGPU_LEAF_DEN * __popcll(__ballot(st == 1 && q.leaf != 0)) >= WTGPU_LEAF_NUM * running) break;
            }
            // (Deferring the exact cone-triangle tests of a leaf step — 3 % of its triangles, ~10x a filter test, 1-2 lanes busy — to a step of
            // their own, taken once 4 / 8 / 16 lanes wait for one, was built and measured in round 4: 5 % SLOWER per pass.  The kernel is bound by
            // dependent memory round trips, not by instruction issue, and the deferred test re-fetches its triangle: one more round trip per hit.)
            RP_BEGIN();
            const unsigned long long m_leaf = __ballot(st == 1 && q.leaf != 0);
            if (st == 1 && q.leaf != 0) cq_leaf_step(a.sc, env, stack, tris, q);
            RP_END(4, m_leaf);
            if (st == 1 && !cq_running(q)) st = 2;
            const int waiting = __popcll(__ballot(st == 2)) + (exhausted ? 0 : __popcll(__ballot(st == 0)));
            if (waiting >= leave_at || !__ballot(st == 1)) break;
        }
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
#ifdef WTGPU_REFILL_PROF
    if (lane == 0)
        for (int i = 0; i < 6; ++i) {
            atomicAdd(a.st.counters + kNumCounters + i, pc[i]);
            atomicAdd(a.st.counters + kNumCounters + 8 + i, pl[i]);
        }
    // (the ray timer runs in the first fetching lane: add what other lanes hold)
    if (lane != 0 && pc[5]) { atomicAdd(a.st.counters + kNumCounters + 5, pc[5]); atomicAdd(a.st.counters + kNumCounters + 8 + 5, pl[5]); }
#endif
}

// Heavy traversals: one wavefront (64-thread block) per walk, persistent blocks pulling from the heavy queue.
__global__ void __launch_bounds__(64, WTGPU_LB_HEAVY) k_trace_heavy(launch_args_t a) {
    __shared__ coop_shared_t sh;
    __shared__ uint32_t s_item;
    coop_set_dropped_counter(sh, a.st.counters + kDroppedSlot);
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_HEAVY_COUNT];
    const uint32_t* hq = a.st.heavy_queue;
    uint32_t* head = ctl + CTL_HEAVY_HEAD;
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    const size_t W2 = 2 * (size_t)a.st.cap;
    const bool rt = a.sc.sensor.ray_trace_only || a.sc.opts.force_ray_tracing;
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(head, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        __syncthreads();
        if (item >= n) break;
        const uint32_t w = hq[item];
        const walk_trace_in_t wk = walk_load_trace_in(a.st.walks, a.st.walk_words, w);   // uniform address: broadcast
        const uint_list_t tris{a.st.tris + (size_t)w * kTriListWords, 1u, a.collect_list ? kMaxConeTris : 0u};   // see k_trace
        const cone_t env = walk_trace_envelope(a.sc, wk);
        unsigned long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const long long tt0 = a.profile == 2 ? clock64() : 0;
        const float dist0 = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(dist)]);
        const uint32_t seg0 = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(ntris)];
        const uint32_t nray0 = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(n_ray_queries)], ncone0 = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(n_cone_queries)];
        ray_hit_t axis;   // the closest hit of the beam axis, found by k_trace (traverse_axis, wt/bvh.h)
        axis.tuid = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(tuid)];
        axis.bx = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(bx)]);
        axis.by = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(by)]);
        axis.dist = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(pdist)]);
        axis.front_face = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(front_face)];
        const uint32_t short0 = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(overflow)];
        const trav_result_t tr2 = coop_traverse(a.sc, env, wavenum_to_wavelen_m(wk.k), WT_INF, rt, sh, tris, a.profile == 2 ? prof : nullptr, true, seg0, dist0, nray0, ncone0,
                                                &axis, !a.collect_list, a.heavy_probe != 0, a.heavy_cache ? short0 : kInvalid, a.heavy_cache ? wk.prev_offset_tuid : kInvalid, a.heavy_cache != 0);
        if (a.profile == 2 && threadIdx.x == 0) {
            prof[3] = (unsigned long long)(clock64() - tt0);
            for (int q = 0; q < 4; ++q) atomicAdd(a.st.counters + kNumCounters + q, prof[q]);
            atomicAdd(a.st.counters + kNumCounters + 5, prof[5]);
            atomicAdd(a.st.counters + kNumCounters + 6, prof[6]);
            atomicAdd(a.st.counters + kNumCounters + 7, prof[7]);
            for (int q = 8; q < 12; ++q) atomicAdd(a.st.counters + kNumCounters + q, prof[q]);   // (WTGPU_COOP_PROF: batch counts)
            atomicAdd(a.st.counters + kNumCounters + 4, 1ull);
        }
        if (threadIdx.x == 0) {
            soa_store(a.st.trav, kTravWords, w, tr2);
            ctr.segments += 1;
            ctr.ray_queries += tr2.n_ray_queries;
            ctr.cone_queries += tr2.n_cone_queries;
            if (a.collect_list) ctr.cone_tri_overflow += tr2.overflow;
        }
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}

// Interaction step of the queued walks.  PASS 0 (A): the queue of the round — surface interactions; walks whose beam axis misses
// every triangle of the interaction region (8 % of them; what follows costs ~50x a surface interaction) are only appended to the
// pass-B queue.  k_edges then gathers the classified-edge set of their regions.  PASS 1 (B): Fraunhofer aperture construction,
// null interactions; the one walk in eight whose aperture has edges goes on to the pass-C queue (k_interact_c).
// No BVH query happens in these passes (the trace kernels resolved the primary triangle): they carry no traversal stack.
template <int PASS>
__device__ inline __attribute__((always_inline)) void interact_body(const launch_args_t& a, int in, int first_round) {
    constexpr bool PASS_B = PASS == 1;
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = PASS_B ? ctl[CTL_INTB_COUNT] : queue_count(ctl, in);
    if (!PASS_B && blockIdx.x == 0 && threadIdx.x == 0) {
        ctl[CTL_HEAVY_COUNT] = 0;   // for the next round's k_trace
        ctl[CTL_HEAVY_HEAD] = 0;
        ctl[CTL_HEAD_TRACE] = 0;
    }
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    const size_t W2 = 2 * (size_t)a.st.cap;
    const fsd_pool_t pool{a.st.fsd_hdr, a.st.fsd_edges, ctl + CTL_FSD_COUNTER, a.st.fsd_cap, ctl + CTL_FSD_ECOUNTER, a.st.fsd_ecap};
    for (;;) {
        const uint32_t qi = wave_grab(ctl + (PASS_B ? CTL_INTB_HEAD : CTL_HEAD_INTERACT)) + (threadIdx.x & 63);
        if (qi - (threadIdx.x & 63) >= n) break;
        bool cont = false;
        uint32_t w = 0;
        fsd_defer_t defer;
        defer.pending = defer.resolved = 0;
        defer.slot = defer.base = defer.next_try = defer.end_draws = 0;
        defer.defer_sampling = PASS_B ? 1u : 0u;
        defer.to_sampling_pass = 0;
        defer.have_aperture = 0;
        defer.split_no_primary = PASS_B ? 0u : 1u;
        defer.known_no_primary = PASS_B ? 1u : 0u;
        defer.no_primary = 0;
        defer.has_gather = defer.gather_n_edges = defer.gather_edge_overflow = 0;
        defer.gather_flux = 0.f;
        defer.gather_edges = nullptr;
        bool need_gather = false;
        if (qi < n) {
            w = PASS_B ? a.st.intb_queue[qi] : queue_walk(a, ctl, in, qi, first_round);
            uint32_t i, stream;
            walk_ident(a, w, i, stream);
            const uint64_t j = a.j0 + i;
            const uint32_t pix = (uint32_t)(j % a.npix);
            const uint64_t s = a.sample_begin + j / a.npix;
            const uint64_t sample_id = ((uint64_t)pix << 32) | (s & 0xFFFFFFFFull);
            walk_t wk;
            soa_load(a.st.walks, a.st.walk_words, w, wk);
            trav_result_t tr;
            soa_load(a.st.trav, kTravWords, w, tr);
            const uint_list_t tris{a.st.tris + (size_t)w * kTriListWords, 1u, kMaxConeTris};
            const vertex_store_t vs{a.st.verts, a.st.vert_words, w};
            const bool queued_for_c = PASS_B && tr.tuid == kApertureMarker;   // k_edges built the aperture and queued the walk for pass C
            if (PASS_B && tr.tuid == kNullApertureMarker) {   // k_edges built the aperture: no segments (the step restarts the beam)
                defer.have_aperture = 1;
                defer.slot = __float_as_uint(tr.by);
            }
            if (PASS_B && tr.tuid == kGatherMarker) {   // k_edges left the region's sorted classified-edge ids in the walk's list slot
                defer.has_gather = 1;
                defer.gather_n_edges = tr.n_ray_queries;
                defer.gather_edge_overflow = tr.n_cone_queries;
                const uint32_t off = __float_as_uint(tr.bx);   // offset into the round's edge pool
                defer.gather_edges = a.st.epool + off;
            }
            const long long pb0 = PASS_B && a.profile == 3 ? clock64() : 0;
            if (!queued_for_c) cont = bdpt_walk_step<PASS_B ? 2 : 1>(a.sc, wk, tr, tris, vs, pool, a.seed, sample_id, stream, &ctr, nullptr, &defer);
            if (PASS_B && defer.to_sampling_pass) a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(by)] = defer.slot;
            if (PASS_B && a.profile == 3) {   // pass-B cost by the number of gathered scene edges
                const int bin = defer.has_gather ? 32 - __clz((int)defer.gather_n_edges) : 0;   // 0: no gather / none
                atomicAdd(a.st.counters + kNumCounters + 56 + bin, 1ull);
                atomicAdd(a.st.counters + kNumCounters + 72 + bin, (unsigned long long)(clock64() - pb0));
            }
            // a region that did not fit the bounded list: its edge set comes from a walk of the whole region (k_edges)
            // (... or whose list holds more than kMaxEdgeIds / 3 triangles: the per-lane edge set of pass B is bounded)
            if (!PASS_B && defer.no_primary && !tr.ballistic && a.sc.opts.FSD && (tr.overflow > 0 || tr.ntris > kMaxEdgeIds / 3 || !a.collect_list)) need_gather = true;
            if (!defer.no_primary && !defer.to_sampling_pass && !queued_for_c) {
                wk.active = cont ? 1u : 0u;
                soa_store(a.st.walks, a.st.walk_words, w, wk);
            }
        }
        if (!PASS_B) wave_append(a.st.intb_queue, ctl + CTL_INTB_COUNT, defer.no_primary != 0, w);
        if (!PASS_B) wave_append(a.st.gather_queue, ctl + CTL_GATHER_COUNT, need_gather, w);
        if (PASS_B) wave_append(a.st.intc_queue, ctl + CTL_INTC_COUNT, defer.to_sampling_pass != 0, w);
        queue_append(a, ctl, 1 - in, cont, w);
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}

// The classified-edge set of the interaction regions that overflowed the bounded triangle list: the WHOLE region, whatever its
// triangle count — the reference's unbounded std::vector (include/wt/ads/traversal_common.hpp:124-148).  One wavefront per walk
// (coop_gather, edges only): only subtrees that hold classified edges are entered and only edge-bearing triangles are tested, 64 at
// a time (a wide beam over the whole scene still meets ~10^3 of them: a single lane needs milliseconds for that).
// Sorted ids -> the walk's list slot, marker + count -> its traversal record.
__global__ void __launch_bounds__(64, 3) k_edges(launch_args_t a) {
    __shared__ coop_gather_shared_t sh;
    __shared__ coop_edges_t eg;
    __shared__ uint32_t s_item;
    coop_set_dropped_counter(sh, a.st.counters + kDroppedSlot);
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_GATHER_COUNT];
    const size_t W2 = 2 * (size_t)a.st.cap;
    const fsd_pool_t pool{a.st.fsd_hdr, a.st.fsd_edges, ctl + CTL_FSD_COUNTER, a.st.fsd_cap, ctl + CTL_FSD_ECOUNTER, a.st.fsd_ecap};
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(ctl + CTL_GATHER_HEAD, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        __syncthreads();
        if (item >= n) break;
        const uint32_t w = a.st.gather_queue[item];
        walk_t wk;
        soa_load(a.st.walks, a.st.walk_words, w, wk);   // uniform address: broadcast
        const float beam_dist = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(dist)]);
        const float region_depth = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(region_depth)]);
        const range_t izr{beam_dist, beam_dist + region_depth};
        const cone_t tcone = walk_trace_envelope(a.sc, wk);
        const gather_out_t g = coop_gather(a.sc, tcone, izr, wk.beam.env, cone_frame(wk.beam.env), izr, vec2{1.f, 1.f}, false, sh, false, true, nullptr, 1, &eg);
        __syncthreads();
        // sorted ids -> the round's edge pool: any number of them in bitmap mode, the sorted 96-entry list for scenes with more than 32768 classified
        // edges.  (Until round 4 that list went into the walk's triangle-list slot, which a later pass may still read as triangles.)
        const bool bitmap = a.sc.n_edges <= kCoopEdgeBits;
        uint32_t n_edges = bitmap ? coop_edge_count(a.sc, eg) : g.n_edges, dropped = bitmap ? 0u : g.edge_overflow, off = 0;
        if (threadIdx.x == 0) s_item = n_edges ? atomicAdd(ctl + CTL_EPOOL_COUNT, n_edges) : 0u;
        __syncthreads();
        off = s_item;
        __syncthreads();
        if (off + n_edges > a.st.epool_cap) {   // pool exhausted (8M ids per round): reported, cannot happen in the shipped scenes
            dropped += n_edges;
            n_edges = 0;
        } else if (bitmap)
            coop_edge_write(a.sc, eg, a.st.epool + off, n_edges);
        else
            for (uint32_t j = threadIdx.x; j < n_edges; j += 64) a.st.epool[off + j] = eg.edge_ids[j];
        // Regions with many edges: the aperture is built right here, by the whole wavefront (wt/coop_fsd.h), instead of by one lane of
        // pass B; walks whose aperture has segments go straight to the pass-C queue.
        uint32_t marker = kGatherMarker, slot = 0;
        if (n_edges >= a.coop_aperture_min) {
            const uint32_t* eids = a.st.epool + off;
            __syncthreads();   // the ids were written by other lanes
            if (threadIdx.x == 0) s_item = fsd_pool_alloc(pool);
            __syncthreads();
            slot = s_item;
            __syncthreads();
            if (slot < pool.cap) {
                fsd_aperture_t ap;
                const vec3 sd3 = beam_footprint(wk.beam, beam_dist) / kBeamEnvelope;
                const bool ok = coop_build_aperture(a.sc, cone_frame(wk.beam.env), wk.beam.k, wk.beam.env, eids, n_edges, vec2{sd3.x, sd3.y}, pool, slot, ap);
                marker = ap.n_edges > 0 ? kApertureMarker : kNullApertureMarker;
                if (threadIdx.x == 0) {
                    pool.hdr[slot] = ap;
                    if (marker == kApertureMarker) a.st.intc_queue[atomicAdd(ctl + CTL_INTC_COUNT, 1u)] = w;
                    if (a.count_stats) {
                        if (dropped) atomicAdd(a.st.counters + offsetof(bdpt_counters_t, edge_overflow) / sizeof(unsigned long long), (unsigned long long)dropped);
                        if (ap.overflow) atomicAdd(a.st.counters + offsetof(bdpt_counters_t, fsd_edge_overflow) / sizeof(unsigned long long), (unsigned long long)ap.overflow);
                        if (!ok) atomicAdd(a.st.counters + offsetof(bdpt_counters_t, fsd_pool_overflow) / sizeof(unsigned long long), 1ull);
                    }
                }
            }
        }
        if (threadIdx.x == 0) {
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(tuid)] = marker;
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(by)] = slot;
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(bx)] = off;
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(n_ray_queries)] = n_edges;
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(n_cone_queries)] = dropped;
            if (a.profile) {
                atomicAdd(a.st.counters + kNumCounters + 5, 1ull);
                atomicAdd(a.st.counters + kNumCounters + 6, (unsigned long long)n_edges);
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kBlock, WTGPU_LB_INTERACT) k_interact(launch_args_t a, int in, int first_round) { interact_body<0>(a, in, first_round); }
__global__ void __launch_bounds__(kBlock, WTGPU_LB_INTERACT_B) k_interact_b(launch_args_t a, int in) { interact_body<1>(a, in, 0); }
// Intercepted power of interaction regions that overflowed the bounded list (find_closest_triangle's sum over ALL region triangles,
// plt_bdpt_detail.hpp:391-416) for the pass-C walks.  Such regions hold 10^3..10^5 triangles (a wide emitter beam over a finely
// tessellated mesh), 5000 on average in the headline workload: one wavefront per region would leave the round waiting for the
// largest one (measured: 27 ms for a 130,000-triangle region).  k_flux_split cuts the part of the tree that overlaps the region
// into subtrees of <= kFluxTaskTris (2048; swept 128 / 512 / 2048: 247 / 216 / 208 ms per pass) triangles, k_flux_tasks sums every subtree on whichever wavefront is free (f64 atomics).
__global__ void __launch_bounds__(64, 3) k_flux_split(launch_args_t a) {
    __shared__ coop_gather_shared_t sh;
    __shared__ uint32_t s_item;
    coop_set_dropped_counter(sh, a.st.counters + kDroppedSlot);
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_INTC_COUNT];
    const int lane = threadIdx.x & 63;
    // 64 queue items per grab: every lane looks at one walk's marker (most pass-C walks have a region that fitted its list and need no
    // split — bidir_room: 400,000 items a round, a few thousand to split; one item per grab was 11.5 ms of a 125-ms batch there), the
    // wavefront then cuts the regions of the flagged ones, one after the other
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(ctl + CTL_FSPLIT_HEAD, 64u);
        __syncthreads();
        const uint32_t base = s_item;
        __syncthreads();
        if (base >= n) break;
        uint32_t w_mine = 0;
        bool need = false;
        if (base + (uint32_t)lane < n) {
            w_mine = a.st.intc_queue[base + lane];
            need = is_region_marker(a.st.trav[(size_t)w_mine * kTravWords + WT_TRAV_WORD(tuid)]);
        }
        unsigned long long m = __ballot(need);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t w = (uint32_t)__shfl((int)w_mine, src, 64);   // block-uniform
            const walk_trace_in_t wk = walk_load_trace_in(a.st.walks, a.st.walk_words, w);
            const cone_t tcone = walk_trace_envelope(a.sc, wk);
            const float beam_dist = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(dist)]);
            const float region_depth = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(region_depth)]);
            if (threadIdx.x == 0) a.st.facc[w] = 0.0;
            coop_split(a.sc, tcone, range_t{beam_dist, beam_dist + region_depth}, sh, a.flux_task_tris, [&](int32_t ptr) {
                const uint32_t idx = atomicAdd(ctl + CTL_FTASK_COUNT, 1u);
                if (idx < a.st.ftask_cap)
                    a.st.ftasks[idx] = make_uint2(w, (uint32_t)ptr);
                else
                    atomicAdd(a.st.counters + offsetof(bdpt_counters_t, fsd_pool_overflow) / sizeof(unsigned long long), 1ull);   // reported; cannot happen below 4M tasks per batch
            });
            __syncthreads();
        }
    }
}
__global__ void __launch_bounds__(64, WTGPU_LB_FLUX) k_flux_tasks(launch_args_t a) {
    __shared__ coop_gather_shared_t sh;
    __shared__ uint32_t s_item;
    coop_set_dropped_counter(sh, a.st.counters + kDroppedSlot);
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = min(ctl[CTL_FTASK_COUNT], a.st.ftask_cap);
    const size_t W2 = 2 * (size_t)a.st.cap;
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(ctl + CTL_FTASK_HEAD, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        __syncthreads();
        if (item >= n) break;
        const uint2 task = a.st.ftasks[item];
        const uint32_t w = task.x;
        walk_t wk;
        soa_load(a.st.walks, a.st.walk_words, w, wk);   // uniform address: broadcast
        const float beam_dist = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(dist)]);
        const float region_depth = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(region_depth)]);
        const bool want_front = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(front_face)] != 0;
        const range_t izr{beam_dist, beam_dist + region_depth};
        const vec3 sd3 = beam_footprint(wk.beam, beam_dist) / kBeamEnvelope;
        const cone_t tcone = walk_trace_envelope(a.sc, wk);
        unsigned long long gst[2] = {0, 0};
        const double flux = coop_gather(a.sc, tcone, izr, wk.beam.env, cone_frame(wk.beam.env), izr, vec2{sd3.x, sd3.y}, want_front, sh, true, false,
                                        a.profile ? gst : nullptr, (int32_t)task.y).flux;
        if (threadIdx.x == 0) {
            if (flux != 0.0) unsafeAtomicAdd(&a.st.facc[w], flux);
            if (a.profile) {   // WTGPU_PROFILE=1: sizes of the gathered regions
                atomicAdd(a.st.counters + kNumCounters + 0, 1ull);
                atomicAdd(a.st.counters + kNumCounters + 1, gst[0]);
                atomicAdd(a.st.counters + kNumCounters + 2, gst[1]);
                atomicMax(a.st.counters + kNumCounters + 4, gst[0]);
            }
        }
        __syncthreads();
    }
}

// Pass C: the walks of pass B whose Fraunhofer aperture has edges, ONE WAVEFRONT PER WALK.  What a single lane would do serially
// is spread over the 64 lanes: the intercepted-power integral over every triangle of the interaction region (find_closest_triangle,
// plt_bdpt_detail.hpp:391-416 — coop_gather walks the WHOLE region, however many triangles it holds: the reference's unbounded list)
// and the rejection sampling (64 tries per step; tries own their random draws, the lowest accepted try wins like in the sequential
// loop); lane 0 then re-enters bdpt_walk_step with the outcome (vertex append, beam transform, Russian roulette).
//
// The number of tries is wildly non-uniform: most apertures accept within the first 64, but the acceptance probability is
// |sum of amplitudes|^2 / (n x sum of |amplitudes|^2) and the loop runs up to n x 1024 tries (fsd.h: fsd_max_tries, the reference's
// cap) — in the headline workload apertures of 8..15 segments average 1,650 tries and account for 2/3 of this pass's arithmetic
// (WTGPU_PROFILE=3), with single walks keeping one wavefront busy for a millisecond while the round waits.  BLOCK = 64 (k_interact_c)
// therefore gives up after kEasyTries tries and queues the walk for BLOCK = 256 (k_interact_c_hard): four wavefronts per walk, 256
// tries per step, continuing at try kEasyTries.
constexpr uint32_t kEasyTries = 512;
constexpr uint32_t kStageSegs = 256;
#ifndef WTGPU_HARD_BLOCK
#define WTGPU_HARD_BLOCK 256
#endif
template <int BLOCK>
__device__ inline __attribute__((always_inline)) void interact_c_body(const launch_args_t& a, int in) {
    constexpr bool HARD = BLOCK > 64;
    __shared__ uint32_t s_item;
    __shared__ uint32_t s_tmin;
    __shared__ float s_res[3];
    __shared__ stack_entry_t lds[8];   // the resumed step does no BVH queries; lane 0's stack is a formality
    __shared__ fsd_edge_t s_seg[kStageSegs];   // the walk's aperture segments (7 KB; larger apertures are read from the pool)
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[HARD ? CTL_INTD_COUNT : CTL_INTC_COUNT];
    const uint32_t* queue_in = HARD ? a.st.intd_queue : a.st.intc_queue;
    const int tid = threadIdx.x, lane = threadIdx.x & 63;
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    const size_t W2 = 2 * (size_t)a.st.cap;
    const fsd_pool_t pool{a.st.fsd_hdr, a.st.fsd_edges, ctl + CTL_FSD_COUNTER, a.st.fsd_cap, ctl + CTL_FSD_ECOUNTER, a.st.fsd_ecap};
    for (;;) {
        const long long pl0 = a.profile == 3 ? clock64() : 0;
        if (tid == 0) s_item = atomicAdd(ctl + (HARD ? CTL_INTD_HEAD : CTL_INTC_HEAD), 1u);
        __syncthreads();
        const uint32_t item = s_item;
        __syncthreads();
        if (item >= n) break;
        const uint32_t w = queue_in[item];
        uint32_t i, stream;
        walk_ident(a, w, i, stream);
        const uint64_t j = a.j0 + i;
        const uint32_t pix = (uint32_t)(j % a.npix);
        const uint64_t sample_id = ((uint64_t)pix << 32) | ((a.sample_begin + j / a.npix) & 0xFFFFFFFFull);
        const uint32_t rng_draws = a.st.walks[(size_t)w * a.st.walk_words + WT_WALK_WORD(rng_draws)];
        const uint32_t slot = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(by)];   // left by pass B
        fsd_aperture_t ap = pool.hdr[slot];
        const fsd_edges_ref_t ed = fsd_pool_edges(pool, slot);
        const long long pc0 = a.profile == 3 ? clock64() : 0;
        if (!HARD) {
            // ---- intercepted power of the whole region (same z-slab, cone and facing as the reference's list-based sum)
            const walk_trace_in_t wk = walk_load_trace_in(a.st.walks, a.st.walk_words, w);   // uniform address: broadcast
            cone_t benv = wk.env;   // the beam's own envelope (not offset for tracing)
            const float tr_dist = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(dist)]);
            const float tr_depth = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(region_depth)]);
            const uint32_t tr_tuid = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(tuid)], tr_ntris = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(ntris)];
            const bool tr_front = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(front_face)] != 0;
            const range_t izr{tr_dist, tr_dist + tr_depth};
            const vec2 axes = cone_axes(benv, tr_dist);
            const vec2 sigma{axes.x / kBeamEnvelope, axes.y / kBeamEnvelope};
            double flux;
            if (is_region_marker(tr_tuid)) {   // the region overflowed the bounded list: summed over all of it by k_flux_split / k_flux_tasks
                flux = a.st.facc[w];
            } else {   // lane = triangle of the (complete) list, wave reduction (bdpt_walk_step computes the same sum triangle by triangle)
                const uint32_t* tl = a.st.tris + (size_t)w * kTriListWords;
                flux = (uint32_t)lane < tr_ntris ? (double)region_triangle_flux(a.sc, cone_frame(benv), benv, izr, sigma, tl[lane], tr_front) : 0.0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) flux += __shfl_xor(flux, off, 64);
            }
            const float I = (float)(1.0 - flux);
            ap.recp_I = I > 0.f ? 1.f / I : 0.f;
            if (lane == 0) pool.hdr[slot] = ap;
        }
        // ---- rejection sampling.  A try reads every segment of the aperture twice (segment selection by a linear scan of the pdfs,
        // then the density / scattering-function sums): the segments are staged in LDS once per walk.
        const sampler_t ss = make_sampler(a.seed, sample_id, stream, 0);
        const uint32_t base = fsd_tries_base(make_sampler(a.seed, sample_id, stream, rng_draws));
        const uint32_t max_tries = fsd_max_tries(ap);
        const uint32_t t_begin = HARD ? kEasyTries : 0u, t_end = HARD ? max_tries : (max_tries < kEasyTries ? max_tries : kEasyTries);
        bool acc = false;
        uint32_t t_acc = 0;
        float rx = 0.f, ry = 0.f, rf = 0.f;
        auto run_tries = [&](const fsd_edges_ref_t& edr) {
            for (uint32_t t0 = t_begin; t0 < t_end && !acc; t0 += BLOCK) {
                const uint32_t t = t0 + (uint32_t)tid;
                fsd_try_t r{{0.f, 0.f}, 0.f, 0u};
                if (t < t_end) r = fsd_try(a.sc, ap, edr, sampler_at(ss, base + t * kFsdDrawsPerTry));
                if (!HARD) {
                    const unsigned long long am = __ballot(r.accept != 0);
                    if (am) {
                        const int wl = __ffsll((long long)am) - 1;
                        rx = __shfl(r.x.x, wl, 64);
                        ry = __shfl(r.x.y, wl, 64);
                        rf = __shfl(r.f, wl, 64);
                        t_acc = t0 + (uint32_t)wl;
                        acc = true;
                    }
                } else {   // the lowest accepted try of the block
                    if (tid == 0) s_tmin = 0xFFFFFFFFu;
                    __syncthreads();
                    if (r.accept) atomicMin(&s_tmin, t);
                    __syncthreads();
                    const uint32_t tm = s_tmin;
                    if (tm != 0xFFFFFFFFu) {
                        if (t == tm) {
                            s_res[0] = r.x.x;
                            s_res[1] = r.x.y;
                            s_res[2] = r.f;
                        }
                        __syncthreads();
                        rx = s_res[0];
                        ry = s_res[1];
                        rf = s_res[2];
                        t_acc = tm;
                        acc = true;
                    }
                    __syncthreads();
                }
            }
        };
        if (ap.n_edges <= kStageSegs) {
            __syncthreads();   // (the previous walk's tries are done with the buffer)
            for (uint32_t k = (uint32_t)tid; k < ap.n_edges; k += BLOCK) s_seg[k] = ed.p[k];
            __syncthreads();
            run_tries(fsd_edges_ref_t{s_seg, 1});
        } else
            run_tries(ed);
        if (a.profile == 3 && tid == 0) {   // WTGPU_PROFILE=3: pass-C cost by aperture size (bin = floor(log2(segments)))
            const int bin = 31 - __clz((int)max(ap.n_edges, 1u));
            atomicAdd(a.st.counters + kNumCounters + 8 + bin, 1ull);
            atomicAdd(a.st.counters + kNumCounters + 24 + bin, (unsigned long long)(acc ? t_acc + 1u - t_begin : t_end - t_begin));
            atomicAdd(a.st.counters + kNumCounters + 40 + bin, (unsigned long long)(clock64() - pc0));
            atomicAdd(a.st.counters + kNumCounters + 112 + bin, (unsigned long long)(pc0 - pl0));
        }
        const long long pm0 = a.profile == 3 ? clock64() : 0;
        if (!HARD && !acc && t_end < max_tries) {   // none of the first kEasyTries tries accepted: four wavefronts take over
            if (lane == 0) a.st.intd_queue[atomicAdd(ctl + CTL_INTD_COUNT, 1u)] = w;
            continue;
        }
        // ---- commit: thread 0 resumes the step with the outcome
        bool cont = false;
        if (tid == 0) {
            walk_t wk;
            soa_load(a.st.walks, a.st.walk_words, w, wk);
            trav_result_t tr;
            soa_load(a.st.trav, kTravWords, w, tr);
            fsd_defer_t defer;
            defer.defer_sampling = defer.to_sampling_pass = 0;
            defer.have_aperture = 1;
            defer.split_no_primary = 0;
            defer.known_no_primary = 1;
            defer.no_primary = 0;
            defer.has_gather = defer.gather_n_edges = defer.gather_edge_overflow = 0;
            defer.gather_flux = 0.f;
            defer.gather_edges = nullptr;
            defer.pending = 0;
            defer.resolved = 1;
            defer.slot = slot;
            defer.base = base;
            defer.next_try = 0;
            defer.fs = fsd_finalize(ap, acc, vec2{rx, ry}, rf);
            defer.end_draws = fsd_draws_after(base, acc ? t_acc : max_tries - 1u);
            const uint_list_t tris{a.st.tris + (size_t)w * kTriListWords, 1u, kMaxConeTris};
            const vertex_store_t vs{a.st.verts, a.st.vert_words, w};
            stack_ref_t stack = make_stack_ref(lds, 1, 8, 8, nullptr);
            cont = bdpt_walk_step<2>(a.sc, wk, tr, tris, vs, pool, a.seed, sample_id, stream, &ctr, &stack, &defer);
            wk.active = cont ? 1u : 0u;
            soa_store(a.st.walks, a.st.walk_words, w, wk);
            if (a.profile == 3) atomicAdd(a.st.counters + kNumCounters + 96 + (31 - __clz((int)max(ap.n_edges, 1u))), (unsigned long long)(clock64() - pm0));
        }
        if (tid < 64) queue_append(a, ctl, 1 - in, cont, w);
    }
    if (a.count_stats && tid < 64) flush_counters(a.st.counters, ctr);
}
__global__ void __launch_bounds__(64, WTGPU_LB_INTERACT_C) k_interact_c(launch_args_t a, int in) { interact_c_body<64>(a, in); }
__global__ void __launch_bounds__(WTGPU_HARD_BLOCK) k_interact_c_hard(launch_args_t a, int in) { interact_c_body<WTGPU_HARD_BLOCK>(a, in); }

// ---- plt_path: one walk per sample; k_trace / k_trace_heavy are shared with plt_bdpt (they only read the walk_t
// prefix of the walk record), the interaction step is path_walk_step (wt/path.h): UTD evaluation of the previous aperture (shadow
// rays through the LDS stack), primary triangle, edge query, aperture construction, NEE / sensing splats (f64 atomics), sampling.
__global__ void __launch_bounds__(kBlock) k_path_generate(launch_args_t a) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) {
        uint32_t* ctl = a.st.ctl;
        ctl[CTL_COUNT0] = a.nb;
        ctl[CTL_COUNT1] = 0;
        ctl[CTL_BACK0] = ctl[CTL_BACK1] = 0;
        ctl[CTL_UTD_COUNT0] = ctl[CTL_UTD_COUNT1] = ctl[CTL_FSDQ_COUNT0] = ctl[CTL_FSDQ_COUNT1] = ctl[CTL_FSDQ_HEAD] = ctl[CTL_NEEQ_COUNT] = ctl[CTL_NEEQ_HEAD] = 0;
        ctl[CTL_HEAD_TRACE] = ctl[CTL_HEAD_INTERACT] = ctl[CTL_HEAVY_COUNT] = ctl[CTL_HEAVY_HEAD] = ctl[CTL_FSD_COUNTER] = ctl[CTL_ROUNDS] = 0;
        ctl[CTL_INTB_COUNT] = ctl[CTL_INTB_HEAD] = ctl[CTL_GATHER_COUNT] = ctl[CTL_GATHER_HEAD] = ctl[CTL_INTC_COUNT] = ctl[CTL_INTC_HEAD] = 0;
        ctl[CTL_FTASK_COUNT] = ctl[CTL_FTASK_HEAD] = ctl[CTL_FSPLIT_HEAD] = ctl[CTL_EPOOL_COUNT] = ctl[CTL_FSD_ECOUNTER] = 0;
        ctl[CTL_INTD_COUNT] = ctl[CTL_INTD_HEAD] = 0;
    }
    if (i >= a.nb) return;
    const uint64_t j = a.j0 + i;
    const uint32_t pix = (uint32_t)(j % a.npix);
    const uint64_t s = a.sample_begin + j / a.npix;
    const uint64_t sample_id = ((uint64_t)pix << 32) | (s & 0xFFFFFFFFull);
    path_walk_t pw;
    path_generate(a.sc, a.seed, sample_id, pix % a.sc.sensor.width, pix / a.sc.sensor.width, pw);
    soa_store(a.st.walks, a.st.walk_words, i, pw);
}

// do_fsd (plt_path_detail.hpp:311-346) by ONE WAVEFRONT: lane = wedge (strided over apertures of any size) — the Fermat point on the wedge, the
// UTD coefficients and the two shadow rays (per-lane any-hit traversals on the lane's LDS stack) — coherent sums in f64 by wave reduction; the
// direct path is evaluated redundantly by all lanes (uniform control flow).  Returns (|ts|^2 + |th|^2) / 2.
__device__ inline float coop_do_fsd(const scene_t& sc, const cone_t& cone_from_src, const path_geo_t& src_geo, vec3 dst, const utd_aperture_t& ap, const utd_edge_rec_t* recs,
                                    float k, const stack_ref_t& stack, bdpt_counters_t* ctr) {
    const int lane = threadIdx.x & 63;
    const vec3 src = cone_from_src.o;
    const path_geo_t dst_geo = path_geo_point(dst);
    double tsr = 0, tsi = 0, thr = 0, thi = 0;
    for (uint32_t i = (uint32_t)lane; i < ap.n_edges; i += 64u) {
        utd_diffracting_edge_t f;
        if (!utd_f_edge(sc, ap, recs[i], src, dst, f)) continue;
        const path_geo_t eintr = path_geo_edge(f.edge, f.p);
        if (path_shadow(sc, eintr, src_geo, stack, ctr) || path_shadow(sc, eintr, dst_geo, stack, ctr)) continue;
        const cplx phase = cpolar(1.f, -k_times_len(k, f.ro + f.ri));
        const cplx a = phase * f.utd.Ds, b = phase * f.utd.Dh;
        tsr += a.re;
        tsi += a.im;
        thr += b.re;
        thi += b.im;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tsr += __shfl_xor(tsr, off, 64);
        tsi += __shfl_xor(tsi, off, 64);
        thr += __shfl_xor(thr, off, 64);
        thi += __shfl_xor(thi, off, 64);
    }
    cplx ts{(float)tsr, (float)tsi}, th{(float)thr, (float)thi};
    if (cone_contains(cone_from_src, dst)) {
        bdpt_counters_t* c0 = lane == 0 ? ctr : nullptr;
        if (!path_shadow(sc, src_geo, dst_geo, stack, c0)) {
            const cplx phase = cpolar(1.f, -k_times_len(k, length(dst - src)));
            ts = ts + phase;
            th = th + phase;
        }
    }
    return (cnorm(ts) + cnorm(th)) / 2.f;
}

// plt_path, before the interaction step: the coherent UTD sum of the aperture the walk built in the previous round towards this round's
// interaction point (plt_path_detail.hpp:616-636) — one wavefront per walk, queue filled by the previous round's k_path_interact.
__global__ void __launch_bounds__(64, 3) k_path_fsd(launch_args_t a, const path_state_t* __restrict__ ps, uint32_t round) {
    const path_state_t& P = *ps;
    __shared__ stack_entry_t lds[kLdsStack * 64];
    __shared__ uint32_t s_item;
    uint32_t* ctl = a.st.ctl;
    const uint32_t qin = round & 1u;
    const uint32_t n = ctl[CTL_FSDQ_COUNT0 + qin];
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack, 64u);
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    const utd_edge_rec_t* prev_pool = P.utd[(round + 1u) & 1u];
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(ctl + CTL_FSDQ_HEAD, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        __syncthreads();
        if (item >= n) break;
        const uint32_t w = P.fsdq[qin][item];
        const uint32_t empty = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(empty)];
        if (empty) continue;   // (the step ends before do_fsd: plt_path_detail.hpp:577-581)
        path_walk_t pw;
        soa_load(a.st.walks, a.st.walk_words, w, pw);   // uniform address: broadcast
        const vec3 origin{__uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(origin.x)]), __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(origin.y)]),
                          __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(origin.z)])};
        const float dist = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(dist)]);
        const vec3 interaction_wp = origin + dist * pw.w.beam.env.d;
        const float f = coop_do_fsd(a.sc, pw.prev_beam.env, path_geo_prev(pw.w), interaction_wp, pw.ap, prev_pool + pw.ap.edge_offset, pw.w.beam.k, stack, &ctr);
        if (threadIdx.x == 0) P.fsd_f[w] = f;
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}

// plt_path, the classified-edge set of regions the per-lane means cannot hold (path_defer_t::need_gather): one wavefront per walk.  Non-ballistic
// hit: the triangles of the interaction region [dist, dist + depth] of the traced cone.  Ballistic hit: the reference's cone query around the hit
// (plt_path_detail.hpp:645-650: closest cone hit inside dist -+ z / 2, then every triangle inside the final slab) — closest hit by the
// wave-cooperative query, then a walk of that slab.  Edge ids through the LDS bitmap: any number, sorted, into the round's edge pool.
__global__ void __launch_bounds__(64, 2) k_path_edges(launch_args_t a, const path_state_t* __restrict__ ps) {
    const path_state_t& P = *ps;
    __shared__ coop_shared_t csh;
    __shared__ coop_gather_shared_t sh;
    __shared__ coop_edges_t eg;
    __shared__ uint32_t s_item;
    coop_set_dropped_counter(csh, a.st.counters + kDroppedSlot);
    coop_set_dropped_counter(sh, a.st.counters + kDroppedSlot);
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_GATHER_COUNT];
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(ctl + CTL_GATHER_HEAD, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        __syncthreads();
        if (item >= n) break;
        const uint32_t w = a.st.gather_queue[item];
        const walk_trace_in_t wk = walk_load_trace_in(a.st.walks, a.st.walk_words, w);   // uniform address: broadcast
        const uint32_t tr_ballistic = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(ballistic)];
        const float dist = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(dist)]);
        const float depth = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(region_depth)]);
        const vec3 origin{__uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(origin.x)]), __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(origin.y)]),
                          __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(origin.z)])};
        const bool ballistic = tr_ballistic || cone_is_ray(wk.env);
        cone_t cone = wk.env;
        range_t slab{dist, dist + depth};
        bool any = true;
        if (!ballistic)
            cone.o = origin;   // the traced (self-intersection-offset) cone, like the record's triangles
        else {
            const float zdist = cone_axes(cone, dist).x * kMajorAxisToZScale;
            const range_t sr{dist - zdist / 2.f, dist + zdist / 2.f};
            cone_hit_t ch;
            const uint_list_t none{nullptr, 1u, 0u};
            coop_cone(a.sc, cone, sr, 1.f, csh, none, ch);
            any = ch.ntris + ch.overflow > 0;
            slab = cone_search_range(cone, sr, ch.dist, 1.f);
            __syncthreads();
        }
        uint32_t n_edges = 0, off = 0, dropped = 0;
        if (any) {
            const gather_out_t g = coop_gather(a.sc, cone, slab, cone, cone_frame(cone), slab, vec2{1.f, 1.f}, false, sh, false, true, nullptr, 1, &eg);
            __syncthreads();
            // (every id list goes into the round's edge pool: the walk's triangle-list slot is read again as triangles by PASS 1)
            const bool bitmap = a.sc.n_edges <= kCoopEdgeBits;
            n_edges = bitmap ? coop_edge_count(a.sc, eg) : g.n_edges;
            dropped = bitmap ? 0u : g.edge_overflow;
            if (threadIdx.x == 0) s_item = n_edges ? atomicAdd(ctl + CTL_EPOOL_COUNT, n_edges) : 0u;
            __syncthreads();
            off = s_item;
            __syncthreads();
            if (off + n_edges > a.st.epool_cap) {   // pool exhausted (8M ids per round): reported
                dropped += n_edges;
                n_edges = 0;
            } else if (bitmap)
                coop_edge_write(a.sc, eg, a.st.epool + off, n_edges);
            else
                for (uint32_t j = threadIdx.x; j < n_edges; j += 64) a.st.epool[off + j] = eg.edge_ids[j];
        }
        if (threadIdx.x == 0) {
            P.gather_info[w] = make_uint2(off, n_edges);
            if (dropped && a.count_stats) atomicAdd(a.st.counters + offsetof(bdpt_counters_t, edge_overflow) / sizeof(unsigned long long), (unsigned long long)dropped);
        }
        __syncthreads();
    }
}

// PASS 0: the round's queue; walks whose classified-edge set needs a wavefront are only queued for k_path_edges.  PASS 1: those walks, with it.
template <int PASS>
__device__ inline __attribute__((always_inline)) void path_interact_body(const launch_args_t& a, const path_state_t& P, int in, int first_round, uint32_t round) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = PASS ? ctl[CTL_GATHER_COUNT] : queue_count(ctl, in);
    if (!PASS && blockIdx.x == 0 && threadIdx.x == 0) {
        ctl[CTL_HEAVY_COUNT] = 0;   // for the next round's k_trace
        ctl[CTL_HEAVY_HEAD] = 0;
        ctl[CTL_HEAD_TRACE] = 0;
    }
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack);
    const uint32_t stream = a.sc.opts.integrator == INTEGRATOR_PATH_FORWARD ? STREAM_EMITTER_WALK : STREAM_SENSOR_WALK;
    const utd_pool_t pool{P.utd[round & 1u], ctl + CTL_UTD_COUNT0 + (round & 1u), P.utd_cap};
    const utd_edge_rec_t* prev_pool = P.utd[(round + 1u) & 1u];
    for (;;) {
        const uint32_t qi = wave_grab(ctl + (PASS ? CTL_INTB_HEAD : CTL_HEAD_INTERACT)) + (threadIdx.x & 63);
        if (qi - (threadIdx.x & 63) >= n) break;
        bool cont = false, carries_fsd = false, nee = false, gather = false;
        uint32_t w = 0;
        if (qi < n) {
            w = PASS ? a.st.gather_queue[qi] : queue_walk(a, ctl, in, qi, first_round);
            const uint64_t j = a.j0 + w;
            const uint32_t pix = (uint32_t)(j % a.npix);
            const uint64_t s = a.sample_begin + j / a.npix;
            const uint64_t sample_id = ((uint64_t)pix << 32) | (s & 0xFFFFFFFFull);
            path_walk_t pw;
            soa_load(a.st.walks, a.st.walk_words, w, pw);
            trav_result_t tr;
            soa_load(a.st.trav, kTravWords, w, tr);
            uint32_t* slot = a.st.tris + (size_t)w * kTriListWords;
            const uint_list_t tris{slot, 1u, kMaxConeTris, reinterpret_cast<float*>(slot + kMaxConeTris)};
            path_defer_t defer;
            defer.have_prev_f = pw.has_fsd;   // evaluated by k_path_fsd (this round), one lane per wedge
            defer.prev_f = pw.has_fsd ? P.fsd_f[w] : 0.f;
            defer.defer_nee = 1;
            defer.nee_pending = 0;
            defer.split_gather = PASS ? 0u : 1u;
            defer.need_gather = 0;
            defer.has_gather = PASS ? 1u : 0u;
            defer.gather_n = 0;
            defer.gather_edges = nullptr;
            if (PASS) {
                const uint2 gi = P.gather_info[w];
                defer.gather_n = gi.y;
                defer.gather_edges = a.st.epool + gi.x;
            }
            cont = path_walk_step(a.sc, pw, tr, tris, prev_pool, pool, a.film, a.seed, sample_id, stream, stack, &ctr, &defer);
            gather = defer.need_gather != 0;
            if (!gather) {
                if (!cont) path_finish(a.sc, a.film, pw);
                pw.w.active = cont ? 1u : 0u;
                soa_store(a.st.walks, a.st.walk_words, w, pw);
                carries_fsd = cont && pw.has_fsd;
                nee = defer.nee_pending != 0;
                if (nee) P.nee_recs[w] = defer.nee;
            }
        }
        if (!PASS) wave_append(a.st.gather_queue, ctl + CTL_GATHER_COUNT, gather, w);
        queue_append(a, ctl, 1 - in, cont && !gather, w);
        wave_append(P.fsdq[(round + 1u) & 1u], ctl + CTL_FSDQ_COUNT0 + ((round + 1u) & 1u), carries_fsd, w);
        wave_append(P.neeq, ctl + CTL_NEEQ_COUNT, nee, w);
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}
__global__ void __launch_bounds__(kBlock, 2) k_path_interact(launch_args_t a, const path_state_t* __restrict__ ps, int in, int first_round, uint32_t round) { path_interact_body<0>(a, *ps, in, first_round, round); }
__global__ void __launch_bounds__(kBlock, 2) k_path_interact_b(launch_args_t a, const path_state_t* __restrict__ ps, int in, uint32_t round) { path_interact_body<1>(a, *ps, in, 0, round); }

// plt_path, after the interaction step: next-event estimation towards the virtual sensor through the aperture the step just built (nee_forward,
// plt_path_detail.hpp:474-518) — one wavefront per walk: coherent UTD sum (coop_do_fsd), beam transform, integrate_beams, light-image splat.
__global__ void __launch_bounds__(64, 3) k_path_nee(launch_args_t a, const path_state_t* __restrict__ ps, uint32_t round) {
    const path_state_t& P = *ps;
    __shared__ stack_entry_t lds[kLdsStack * 64];
    __shared__ uint32_t s_item;
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_NEEQ_COUNT];
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack, 64u);
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    const utd_edge_rec_t* cur_pool = P.utd[round & 1u];
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(ctl + CTL_NEEQ_HEAD, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        __syncthreads();
        if (item >= n) break;
        const uint32_t w = P.neeq[item];
        const path_nee_rec_t r = P.nee_recs[w];   // uniform address
        utd_aperture_t ap;
        soa_load(a.st.walks + offsetof(path_walk_t, ap) / 4, a.st.walk_words, w, ap);   // the aperture k_path_interact just stored
        const path_geo_t src_geo{r.src_wp, r.src_kind, r.src_ng, r.src_tuid};
        const float k = r.beam.k;
        const float f = coop_do_fsd(a.sc, r.beam.env, src_geo, r.sd_beam.env.o, ap, cur_pool + ap.edge_offset, k, stack, &ctr);
        if (threadIdx.x == 0 && f != 0.f) {
            beam_t fsd_beam = r.beam;
            beam_transform_region_interaction(fsd_beam, r.interaction_wp, r.dist, -r.sd_beam.env.d, f);
            const stokes_t sL = integrate_beams(r.sd_beam, fsd_beam);
            film_splat_direct(a.sc, a.film, r.element, sL * r.recp_spectral_pd, k);
            ctr.connections++;
            ctr.light_splats++;
        }
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}

// walks still active after the last round (iteration cap): backward transport splats what they gathered
constexpr uint32_t kFlushGrid = 64;
__global__ void __launch_bounds__(kBlock) k_path_flush(launch_args_t a, int in) {
    const uint32_t n = queue_count(a.st.ctl, in);
    const size_t W2 = 2 * (size_t)a.st.cap;
    for (uint32_t qi = blockIdx.x * kBlock + threadIdx.x; qi < n; qi += kFlushGrid * kBlock) {
        const uint32_t w = queue_walk(a, a.st.ctl, in, qi, 0);
        path_walk_t pw;
        soa_load(a.st.walks, a.st.walk_words, w, pw);
        path_finish(a.sc, a.film, pw);
    }
}

// ---- connections: strategy-major -------------------------------------------------------------------------------------
// plt_bdpt.cpp:105-146 loops over all (s,t) pairs of a sample.  One thread per sample would leave a wavefront executing the
// UNION of its 64 samples' pairs (~80 iterations with ~10 lanes' worth of work: subpath lengths are geometric).  Instead:
//   k_connect_enum  : every sample appends its index to one bucket per valid (s,t) pair (block-aggregated: LDS counts, one global
//                     atomic per bucket and block),
//   k_connect_scan  : prefix sum over the 19x19 bucket sizes,
//   k_connect_strat : persistent; 64 consecutive items of the flattened bucket space = 64 samples with the SAME (s,t): uniform
//                     control flow, coalesced vertex loads; the t>1 fluxes are summed per sample (f64 atomics), t<=1 strategies
//                     splat into the light image directly,
//   k_connect_splat : one film splat per sample with the summed flux (film.hpp:214-342).
__device__ inline bool strategy_valid(const integrator_opts_t& o, int s, int t, int nS, int nT) {
    const int depth = t + s - 2;
    if (t > nT || s > nS) return false;
    if ((t == 1 && s == 1) || depth < 0 || depth > o.max_depth) return false;
    if (!o.emitter_direct && s == 1) return false;
    if (!o.sensor_direct && t == 1) return false;
    if (o.debug_only_s && (int)o.debug_only_s - 1 != s) return false;
    if (o.debug_only_t && (int)o.debug_only_t - 1 != t) return false;
    return true;
}
// bucket (sk, tk): does it hold a valid strategy of a sample with nS / nT vertices?  (the last row / column stands for every s / t >= kKeyDim-1)
__device__ inline bool strategy_class_valid(const integrator_opts_t& o, int sk, int tk, int nS, int nT) {
    const int K = (int)kKeyDim - 1;
    const int t1 = tk < K ? tk : nT, s1 = sk < K ? sk : nS;
    for (int t = tk; t <= t1; ++t)
        for (int s = sk; s <= s1; ++s)
            if (strategy_valid(o, s, t, nS, nT)) return true;
    return false;
}
__device__ inline int wave_max_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

// Block-aggregated bucket append: 1024 samples per block count their valid (s,t) pairs per bucket in LDS, reserve one range per
// bucket with ONE global atomic each, and fill it.  (Wave-aggregated global atomics on the ~30 hot bucket counters serialised in
// L2: PMC SQ_WAIT_ANY 99 % of this kernel's wave cycles, 9.5 ms per pass.)
constexpr int kEnumBlock = 1024;
__global__ void __launch_bounds__(kEnumBlock) k_connect_enum(launch_args_t a) {
    __shared__ uint32_t s_cnt[kNumKeys], s_base[kNumKeys];
    const uint32_t i = blockIdx.x * kEnumBlock + threadIdx.x;
    const size_t W2 = 2 * (size_t)a.st.cap;
    for (uint32_t k = threadIdx.x; k < kNumKeys; k += kEnumBlock) s_cnt[k] = 0;
    int nT = -1, nS = -1;
    if (i < a.nb) {
        nT = (int)a.st.walks[(size_t)(i) * a.st.walk_words + WT_WALK_NVERTS_WORD];
        nS = (int)a.st.walks[(size_t)(a.st.cap + i) * a.st.walk_words + WT_WALK_NVERTS_WORD];
#pragma unroll
        for (int c = 0; c < 4; ++c) a.st.lacc[(size_t)c * a.st.cap + i] = 0.0;
    }
    __syncthreads();
    const int K = (int)kKeyDim - 1;
    const int kT = nT < K ? nT : K, kS = nS < K ? nS : K;
    for (int tk = 0; tk <= kT; ++tk)
        for (int sk = 0; sk <= kS; ++sk)
            if (strategy_class_valid(a.sc.opts, sk, tk, nS, nT)) atomicAdd(&s_cnt[(uint32_t)tk * kKeyDim + (uint32_t)sk], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kNumKeys; k += kEnumBlock) {
        const uint32_t c = s_cnt[k];
        s_base[k] = c ? atomicAdd(a.st.strat_count + k, c) : 0u;
        s_cnt[k] = 0;
    }
    __syncthreads();
    for (int tk = 0; tk <= kT; ++tk)
        for (int sk = 0; sk <= kS; ++sk)
            if (strategy_class_valid(a.sc.opts, sk, tk, nS, nT)) {
                const uint32_t key = (uint32_t)tk * kKeyDim + (uint32_t)sk;
                a.st.strat_items[(size_t)key * a.st.cap + s_base[key] + atomicAdd(&s_cnt[key], 1u)] = i;
            }
}
__global__ void __launch_bounds__(64) k_connect_scan(launch_args_t a) {
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t k = 0; k < kNumKeys; ++k) {
            const uint32_t c = a.st.strat_count[k];
            a.st.strat_prefix[k] = acc;
            acc += c;
            a.st.strat_count[k] = 0;   // ready for the next batch
        }
        a.st.strat_prefix[kNumKeys] = acc;
        a.st.ctl[CTL_STRAT_HEAD] = 0;
        a.st.ctl[CTL_STRAT_HEAD_OPEN] = 0;
    }
}
// OPEN = false: the buckets with one strategy each (all of them while no subpath exceeds 17 vertices).  OPEN = true (k_connect_strat_open): the
// buckets of the last row / column, whose items loop over every longer strategy of their sample — a kernel of its own so that the loop and
// the subpath lengths it needs do not weigh on the common case's registers.
template <bool OPEN>
__device__ inline __attribute__((always_inline)) void connect_strat_body(const launch_args_t& a) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    __shared__ uint32_t s_prefix[kNumKeys + 1];
    constexpr int K = (int)kKeyDim - 1;
    // flattened item space: OPEN = false all buckets (items of the open ones are skipped), OPEN = true the open buckets only
    if (!OPEN) {
        for (uint32_t k = threadIdx.x; k <= kNumKeys; k += kBlock) s_prefix[k] = a.st.strat_prefix[k];
    } else if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t k = 0; k < kNumKeys; ++k) {
            s_prefix[k] = acc;
            if ((int)(k / kKeyDim) == K || (int)(k % kKeyDim) == K) acc += a.st.strat_prefix[k + 1] - a.st.strat_prefix[k];
        }
        s_prefix[kNumKeys] = acc;
    }
    __syncthreads();
    const uint32_t total = s_prefix[kNumKeys];
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack);
    const fsd_pool_t pool{a.st.fsd_hdr, a.st.fsd_edges, a.st.ctl + CTL_FSD_COUNTER, a.st.fsd_cap, a.st.ctl + CTL_FSD_ECOUNTER, a.st.fsd_ecap};
    for (;;) {
        const uint32_t idx = wave_grab(a.st.ctl + (OPEN ? CTL_STRAT_HEAD_OPEN : CTL_STRAT_HEAD)) + (threadIdx.x & 63);
        if (idx - (threadIdx.x & 63) >= total) break;
        if (idx < total) {
            // bucket of this item: last key with prefix <= idx
            uint32_t lo = 0, hi = kNumKeys;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_prefix[mid] <= idx)
                    lo = mid;
                else
                    hi = mid;
            }
            const uint32_t key = lo;
            const int tk = (int)(key / kKeyDim), sk = (int)(key % kKeyDim);
            if (!OPEN && (tk == K || sk == K)) continue;   // (k_connect_strat_open's)
            const uint32_t i = a.st.strat_items[(size_t)key * a.st.cap + (idx - s_prefix[key])];
            const uint64_t j = a.j0 + i;
            const uint32_t pix = (uint32_t)(j % a.npix);
            const uint64_t smp = a.sample_begin + j / a.npix;
            const uint64_t sample_id = ((uint64_t)pix << 32) | (smp & 0xFFFFFFFFull);
            sample_ctx_t ctx;
            soa_load(a.st.ctx, kCtxWords, i, ctx);
            const vertex_store_t svs{a.st.verts, a.st.vert_words, i}, evs{a.st.verts, a.st.vert_words, (size_t)a.st.cap + i};
            auto one = [&](int s, int t) __attribute__((always_inline)) {
                const stokes_t flux = bdpt_strategy(a.sc, pool, a.film, svs, evs, s, t, ctx, a.seed, sample_id, stack, &ctr, nullptr);
                if (t > 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (flux.s[c] != 0.f) unsafeAtomicAdd(&a.st.lacc[(size_t)c * a.st.cap + i], (double)flux.s[c]);
                }
            };
            if constexpr (!OPEN) {
                one(sk, tk);
            } else {
                const int nT = (int)a.st.walks[(size_t)i * a.st.walk_words + WT_WALK_NVERTS_WORD];
                const int nS = (int)a.st.walks[((size_t)a.st.cap + i) * a.st.walk_words + WT_WALK_NVERTS_WORD];
                const int t1 = tk == K ? nT : tk, s1 = sk == K ? nS : sk;
                for (int t = tk; t <= t1; ++t)
                    for (int s = sk; s <= s1; ++s)
                        if (strategy_valid(a.sc.opts, s, t, nS, nT)) one(s, t);
            }
        }
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}
__global__ void __launch_bounds__(kBlock, WTGPU_LB_CONNECT) k_connect_strat(launch_args_t a) { connect_strat_body<false>(a); }
__global__ void __launch_bounds__(kBlock, WTGPU_LB_CONNECT) k_connect_strat_open(launch_args_t a) { connect_strat_body<true>(a); }
__global__ void __launch_bounds__(kBlock) k_connect_splat(launch_args_t a) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.nb) return;
    sample_ctx_t ctx;
    soa_load(a.st.ctx, kCtxWords, i, ctx);
    stokes_t L;
#pragma unroll
    for (int c = 0; c < 4; ++c) L.s[c] = (float)a.st.lacc[(size_t)c * a.st.cap + i];
    film_splat(a.sc, a.film, ctx.element, L, ctx.k);
}

// The same splat for batches that cover (most of) the film: one block per 128-element row segment accumulates the footprints of its
// elements' samples in an LDS tile (3 rows x 130 columns x (planes + 1) f64, plane-major) and adds the tile to the film once.  Per sample the
// plain kernel issues 9 x (planes + 1) f64 atomics on addresses its neighbours in the wavefront hit too — 117 for the Stokes film of the
// polarimetric workload, where it took 19.7 ms of a 204-ms batch (run r4t) — the tile turns them into LDS atomics and one global add per tile
// entry.  Same weights, same products, f64 sums in another order.  Reconstruction-filter radius <= 1 (the host launches the plain kernel
// otherwise); a sample whose element is not where the block expects it (never, for the sensors built so far) goes to the film directly.
constexpr uint32_t kSplatCols = kBlock + 2;
__global__ void __launch_bounds__(kBlock) k_connect_splat_tiled(launch_args_t a) {
    extern __shared__ double tile[];   // [planes + 1][3][kSplatCols]
    const sensor_t& sn = a.sc.sensor;
    const uint32_t W = a.film.width, H = a.film.height;
    const uint32_t S = film_stokes(sn), P = sn.channels * S, PL = P + 1;
    const uint32_t bpr = (W + kBlock - 1) / kBlock;
    const uint32_t row = blockIdx.x / bpr, x0 = (blockIdx.x % bpr) * kBlock;
    const int r = sn.rf_radius;
    const uint32_t n_px = 3 * kSplatCols;
    for (uint32_t q = threadIdx.x; q < n_px * PL; q += kBlock) tile[q] = 0.0;
    __syncthreads();
    const uint32_t x = x0 + threadIdx.x;
    if (x < W && row < H) {
        const uint64_t p = (uint64_t)row * W + x;
        // the samples of this batch that belong to element p: work items i with (j0 + i) % npix == p
        const uint64_t first = (p + a.npix - (a.j0 % a.npix)) % a.npix;
        for (uint64_t i = first; i < a.nb; i += a.npix) {
            sample_ctx_t ctx;
            soa_load(a.st.ctx, kCtxWords, i, ctx);
            stokes_t L;
#pragma unroll
            for (int c = 0; c < 4; ++c) L.s[c] = (float)a.st.lacc[(size_t)c * a.st.cap + i];
            // (what follows is film_splat, wt/film.h, with the tile in place of the film)
            const rfilter_weights_t rw = film_rfilter_weights(sn, ctx.element.offset);
            float val[16];
            for (uint32_t c = 0; c < sn.channels; ++c) {
                const float f = spectrum_f(a.sc, sn.response_spec[c], ctx.k);
                bool ok = true;
                for (uint32_t q = 0; q < S; ++q) {
                    val[c * S + q] = L.s[q] * f;
                    ok = ok && finitef(val[c * S + q]);
                }
                ok = ok && val[c * S] >= 0.f;
                if (!ok)
                    for (uint32_t q = 0; q < S; ++q) val[c * S + q] = 0.f;
            }
            for (int dy = -r; dy <= r; ++dy) {
                const int y = (int)ctx.element.y + dy;
                if (y < 0 || y >= (int)H) continue;
                for (int dx = -r; dx <= r; ++dx) {
                    const int xx = (int)ctx.element.x + dx;
                    if (xx < 0 || xx >= (int)W) continue;
                    const float w = fmaxf_(0.f, rw.wx[dx + r] * rw.wy[dy + r]) * rw.recp_total;
                    const int ty = y - ((int)row - 1), tx = xx - ((int)x0 - 1);
                    if (ty >= 0 && ty < 3 && tx >= 0 && tx < (int)kSplatCols) {
                        const uint32_t q = (uint32_t)ty * kSplatCols + (uint32_t)tx;
                        unsafeAtomicAdd(&tile[q], (double)w);
                        for (uint32_t c = 0; c < P; ++c) unsafeAtomicAdd(&tile[(size_t)(1 + c) * n_px + q], (double)(w * val[c]));
                    } else {
                        const size_t pix = (size_t)y * W + xx;
                        film_add(&a.film.weight[pix], (double)w);
                        for (uint32_t c = 0; c < P; ++c) film_add(&a.film.value[pix * P + c], (double)(w * val[c]));
                    }
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < n_px; q += kBlock) {
        const int y = (int)row - 1 + (int)(q / kSplatCols), xx = (int)x0 - 1 + (int)(q % kSplatCols);
        if (y < 0 || y >= (int)H || xx < 0 || xx >= (int)W) continue;
        const size_t pix = (size_t)y * W + xx;
        const double wsum = tile[q];
        if (wsum != 0.0) film_add(&a.film.weight[pix], wsum);
        for (uint32_t c = 0; c < P; ++c) {
            const double v = tile[(size_t)(1 + c) * n_px + q];
            if (v != 0.0) film_add(&a.film.value[pix * P + c], v);
        }
    }
}

// ---- PMC calibration: a streaming copy with the access width of the SoA state (one dword per lane, fully coalesced) and a known
// byte count, so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be scaled to bytes for THIS access pattern (tools/profile_round.sh)
__global__ void __launch_bounds__(256) k_calib_copy(const uint32_t* in, uint32_t* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i] + 1u;
}

// ---- per-query kernels (traversal parity tests) --------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_trace_rays(scene_t sc, const float* rays, uint32_t n, float* dist, uint32_t* tuid, float* bary, uint32_t* front) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
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
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    // (the CPU checker's 128-entry stack: this kernel answers every query by itself — in the pipeline a lane whose 64-entry stack
    // fills up hands the query to a wavefront, k_trace_heavy)
    stack_entry_t spill[128 - kLdsStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack);
    stack.cap = 128;
    const float* c = cones + 10 * (size_t)i;
    const vec3 d = normalize(vec3{c[3], c[4], c[5]});
    const cone_t env = make_cone(vec3{c[0], c[1], c[2]}, d, build_orthogonal_frame(d).t, c[6], c[8], c[7]);
    const uint_list_t tris{scratch_tris + i, n, kMaxConeTris, reinterpret_cast<float*>(scratch_tris + (size_t)n * kMaxConeTris) + i};
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

// Region summaries of cone queries of ANY size (parity tests of the whole-region machinery): one wavefront per cone runs the
// traversal policy with closest-hit-only cone queries, then — for a diffusive hit — resolves the triangle under the axis and walks
// the region [dist, dist + 2 x major axis] for its triangle count, sorted classified-edge set and intercepted power (sigma = axes/3).
__global__ void __launch_bounds__(64) k_query_regions(scene_t sc, const float* cones, uint32_t n, uint32_t edge_cap, float* dist, uint32_t* flags,
                                                      uint32_t* primary, uint32_t* ntris, uint32_t* nedges, uint32_t* edges, float* flux, unsigned long long* dropped) {
    __shared__ coop_shared_t sh;
    __shared__ coop_gather_shared_t gsh;
    __shared__ coop_edges_t eg;
    coop_set_dropped_counter(sh, dropped);
    coop_set_dropped_counter(gsh, dropped);
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const float* c = cones + 10 * (size_t)i;
    const vec3 d = normalize(vec3{c[3], c[4], c[5]});
    const cone_t env = make_cone(vec3{c[0], c[1], c[2]}, d, build_orthogonal_frame(d).t, c[6], c[8], c[7]);
    const uint_list_t none{nullptr, 1u, 0u};
    const trav_result_t tr = coop_traverse(sc, env, c[9], WT_INF, false, sh, none, nullptr, false, 0, 0.f, 0, 0, nullptr, true);
    uint32_t prim = kInvalid;
    gather_out_t ge{0.0, 0u, 0u, 0u}, gf{0.0, 0u, 0u, 0u};
    if (tr.ballistic) {
        prim = tr.tuid;
    } else if (!tr.empty) {
        const range_t izr{tr.dist, tr.dist + tr.region_depth};
        prim = tr.tuid;   // primary_from_axis (kInvalid: the axis misses the region)
        const vec2 ax = cone_axes(env, tr.dist);
        ge = coop_gather(sc, env, izr, env, cone_frame(env), izr, vec2{1.f, 1.f}, false, gsh, false, true, nullptr, 1, &eg);
        __syncthreads();
        if (sc.n_edges <= kCoopEdgeBits) {
            ge.n_edges = coop_edge_count(sc, eg);
            coop_edge_write(sc, eg, edges + (size_t)i * edge_cap, edge_cap);
        } else
            for (uint32_t j = threadIdx.x; j < ge.n_edges && j < edge_cap; j += 64) edges[(size_t)i * edge_cap + j] = eg.edge_ids[j];
        __syncthreads();
        gf = coop_gather(sc, env, izr, env, cone_frame(env), izr, vec2{ax.x / kBeamEnvelope, ax.y / kBeamEnvelope}, tr.front_face != 0, gsh, true, false);
    }
    if (threadIdx.x == 0) {
        dist[i] = tr.dist;
        flags[i] = (tr.empty ? 1u : 0u) | (tr.ballistic ? 2u : 0u) | (tr.front_face ? 4u : 0u);
        primary[i] = prim;
        ntris[i] = gf.n_tris;
        nedges[i] = ge.n_edges + ge.edge_overflow;
        flux[i] = (float)gf.flux;
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
    UP(textures, h.n_textures)
    size_t tex_words = 0;
    for (uint32_t i = 0; i < h.n_textures; ++i)
        if (h.textures[i].type == TEX_BITMAP)
            tex_words = std::max(tex_words, (size_t)h.textures[i].offset + (size_t)h.textures[i].width * h.textures[i].height * h.textures[i].channels);
        else if (h.textures[i].type == TEX_FUNCTION)
            tex_words = std::max(tex_words, (size_t)h.textures[i].offset + (size_t)h.textures[i].width);
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
        uint64_t per_sample = 4ull * (2 * ((pm ? kPathWalkWords : kWalkWords) + (pm ? 0 : mv * kVertexWords) + kTravWords + kTriListWords) + kCtxWords) + 64ull * 28ull + 2048ull;
        // plt_path: two wedge pools of 48 records per walk, the deferred-NEE records, the queues of the wave-per-walk kernels
        if (pm) per_sample += 2ull * 48ull * sizeof(utd_edge_rec_t) + sizeof(path_nee_rec_t) + 3ull * 4ull + 4ull + sizeof(uint2);
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
        if ((rc = dmalloc(s, &st.verts, path_mode ? 1 : (size_t)st.max_verts * kVertexWords * W2))) return rc;
        if ((rc = dmalloc(s, &st.ctx, kCtxWords * (size_t)st.cap))) return rc;
        if ((rc = dmalloc(s, &st.trav, kTravWords * W2))) return rc;
        if ((rc = dmalloc(s, &st.tris, (size_t)kTriListWords * W2))) return rc;
        if ((rc = dmalloc(s, &st.queue[0], W2))) return rc;
        if ((rc = dmalloc(s, &st.queue[1], W2))) return rc;
        if ((rc = dmalloc(s, &st.heavy_queue, W2))) return rc;
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
    if (r.rounds_launched == kMaxWalkIters) note_rounds(s, rounds);   // (a batch that got every round: what it really needed)
    if (s->timing) {
        // (an event pair that cannot be resolved contributes 0 ms: timings are diagnostics, the render itself has completed)
        auto elapsed = [](hipEvent_t a, hipEvent_t b) {
            float ms = 0.f;
            return hipEventElapsedTime(&ms, a, b) == hipSuccess ? ms : 0.f;
        };
        s->acc[0] += elapsed(r.ev[0], r.ev[1]);
        size_t e = 1;
        for (uint32_t k = 0; k < r.rounds_launched; ++k, e += 6) {
            static const int slot[6] = {1, 7, 2, 8, 9, 10};   // trace, heavy trace, pass A, edges + pass B, region flux, pass C
            for (int q = 0; q < 6; ++q) s->acc[slot[q]] += elapsed(r.ev[e + q], r.ev[e + q + 1]);
        }
        s->acc[3] += elapsed(r.ev[e], r.ev[e + 1]);
    }
    r.busy = false;
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
    bool hp_on = false;
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
    }
#define HP_LAUNCH(slot, ...)                                                                                              \
    do {                                                                                                                  \
        if (hp_on) {                                                                                                      \
            const auto t0_ = std::chrono::steady_clock::now();                                                            \
            hipLaunchKernelGGL(__VA_ARGS__);                                                                              \
            hp_t[slot] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0_).count();      \
            hp_n[slot]++;                                                                                                 \
        } else                                                                                                            \
            hipLaunchKernelGGL(__VA_ARGS__);                                                                              \
    } while (0)
    void rec(chunk_rec_t& r, hipStream_t st_) {
        const auto hp0_ = std::chrono::steady_clock::now();
        if (s->timing && hipEventRecord(r.ev[r.ev_used++], st_) != hipSuccess) ev_fail = true;
        if (hp_on) { hp_t[31] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - hp0_).count(); hp_n[31]++; }
    }
    uint32_t g_full(uint32_t nb) const { return std::min<uint32_t>(grid_round, ((path_mode ? 1u : 2u) * nb + kBlock - 1) / kBlock); }
    void generate(const launch_args_t& a, chunk_rec_t& r, hipStream_t st_) {
        r.ev_used = 0;
        r.rounds_launched = 0;
        rec(r, st_);
        if (path_mode)
            HP_LAUNCH(0, k_path_generate, dim3((a.nb + kBlock - 1) / kBlock), dim3(kBlock), 0, st_, a);
        else
            HP_LAUNCH(1, k_generate, dim3((a.nb + kBlock - 1) / kBlock), dim3(kBlock), 0, st_, a);
        rec(r, st_);
    }
    int rounds(const launch_args_t& a, const path_state_t* ps, chunk_rec_t& r, hipStream_t st_, uint32_t r_begin, uint32_t r_end) {
        const uint32_t nb = a.nb, walks_per_sample = path_mode ? 1u : 2u, gf = g_full(nb);
        const uint32_t grid_div_b = K.grid_div_b, grid_div_c = K.grid_div_c, grid_mul_flux = K.grid_mul_flux;   // persistent grids of the expensive-interaction passes relative to the round's
        const int dbg_stage = K.dbg_stage;
        for (uint32_t round = r_begin; round < r_end; ++round) {
            const int in = (int)(round & 1u), first = round == 0 ? 1 : 0;
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
            if (dbg_stage >= 2 + 3 * (int)round) {
                HP_LAUNCH(3, k_trace_refill, dim3(g0), dim3(kBlock), 0, st_, a, in, first, round);
            }
            rec(r, st_);
            if (dbg_stage >= 3 + 3 * (int)round) HP_LAUNCH(5, k_trace_heavy, dim3(gh), dim3(64), 0, st_, a);
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
            HP_LAUNCH(11, k_interact, dim3(g0), dim3(kBlock), 0, st_, a, in, first);
            rec(r, st_);
            HP_LAUNCH(12, k_edges, dim3(gh), dim3(64), 0, st_, a);
            HP_LAUNCH(13, k_interact_b, dim3(std::max<uint32_t>(1u, g0 / grid_div_b)), dim3(kBlock), 0, st_, a, in);
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
    // after the batch's last round: connections (plt_bdpt) / what is left of the walks (plt_path), the control block's snapshot, the closing event
    int tail(const launch_args_t& a, chunk_rec_t& r, hipStream_t st_) {
        const uint32_t nb = a.nb, gf = g_full(nb);
        if (path_mode) {
            HP_LAUNCH(18, k_path_flush, dim3(kFlushGrid), dim3(kBlock), 0, st_, a, (int)(r.rounds_launched & 1u));
        } else {
            HP_LAUNCH(19, k_connect_enum, dim3((nb + kEnumBlock - 1) / kEnumBlock), dim3(kEnumBlock), 0, st_, a);
            HP_LAUNCH(20, k_connect_scan, dim3(1), dim3(64), 0, st_, a);
            HP_LAUNCH(21, k_connect_strat, dim3(gf), dim3(kBlock), 0, st_, a);
            if ((uint32_t)s->host.opts.max_depth + 2 >= kKeyDim - 1) HP_LAUNCH(22, k_connect_strat_open, dim3(std::max<uint32_t>(1u, gf / 8u)), dim3(kBlock), 0, st_, a);
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
        static const char* hp_names[] = {"k_path_generate","k_generate","(unused)","k_trace_refill","(unused)","k_trace_heavy","k_path_fsd","k_path_interact","k_path_edges","k_path_interact_b","k_path_nee","k_interact","k_edges","k_interact_b","k_flux_split","k_flux_tasks","k_interact_c","k_interact_c_hard","k_path_flush","k_connect_enum","k_connect_scan","k_connect_strat","k_connect_strat_open","k_connect_splat"};
        for (int i = 0; i < 24; ++i)
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
    uint32_t sum = 0;
    for (uint32_t i = 0; i < n; ++i) sum += s->rounds_hist[i];
    return std::min<uint32_t>(kMaxWalkIters, (sum + n - 1) / n + s->knobs.rounds_margin);
}
// the second part of the batch pending on slice k (see batch_launcher_t); blocks the calling thread until its first part has run
static int render_finish_part(wtgpu_scene* s, size_t k, batch_launcher_t& L) {
    wtgpu_scene::pending_t& p = s->pending[k];
    if (!p.active) return WTGPU_OK;
    p.active = false;
    chunk_rec_t& r = *p.rec;
    launch_args_t a;
    std::memcpy(&a, p.args, sizeof(a));
    hipStream_t st_ = s->streams[k];
    // Is the round queue empty?  If not — a batch whose walks outlasted the expectation — another kRoundsStep rounds, and look again.
    constexpr uint32_t kRoundsStep = 8;
    uint32_t launched = p.rounds_first;
    while (launched < kMaxWalkIters) {
        HIP_CHECK(hipEventSynchronize(r.ev_mid));
        const uint32_t q = launched & 1u;
        const uint32_t left = r.h_mid[CTL_COUNT0 + q] + r.h_mid[CTL_BACK0 + q];
        if (left == 0) {
            note_rounds(s, r.h_mid[CTL_ROUNDS]);
            break;
        }
        s->round_fallbacks++;
        const uint32_t next = std::min<uint32_t>(kMaxWalkIters, launched + kRoundsStep);
        const int rc = L.rounds(a, s->d_path_slices[k], r, st_, launched, next);
        if (rc) return rc;
        launched = next;
        if (launched < kMaxWalkIters) {
            HIP_CHECK(hipMemcpyAsync(r.h_mid, a.st.ctl, CTL_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
            HIP_CHECK(hipEventRecord(r.ev_mid, st_));
        }   // (else: every round has been launched; what the batch needed is noted when it is drained)
    }
    return L.tail(a, r, st_);
}
static int finish_all_pending(wtgpu_scene* s) {
    bool any = false;
    for (const auto& p : s->pending) any = any || p.active;
    if (!any) return WTGPU_OK;
    batch_launcher_t L(s);
    for (size_t k = 0; k < s->pending.size(); ++k) {
        const int rc = render_finish_part(s, k, L);
        if (rc) return rc;
    }
    return WTGPU_OK;
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
    a.collect_list = (h.opts.integrator != INTEGRATOR_BDPT || !K.no_lists) ? 1u : 0u;
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
        if (r1 < kMaxWalkIters) {
            HIP_CHECK(hipMemcpyAsync(r.h_mid, a.st.ctl, CTL_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
            HIP_CHECK(hipEventRecord(r.ev_mid, st_));
        }
        wtgpu_scene::pending_t& p = s->pending[k];
        std::memcpy(p.args, &a, sizeof(a));
        p.rec = &r;
        p.rounds_first = r1;
        p.active = true;
        if (r1 >= kMaxWalkIters) {   // nothing to wait for: the whole batch goes out at once, as before round 4
            rc = render_finish_part(s, k, L);
            if (rc) return rc;
        }
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
