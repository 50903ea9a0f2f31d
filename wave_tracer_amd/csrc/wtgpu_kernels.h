// wave_tracer_amd — what the kernel translation units (kernels_*.hip) and the host side (wtgpu.hip) share: the slice state, the launch block,
// the device-side queue helpers and the kernels' declarations.  The kernels of one batch, in launch order:
//   kernels_walk.hip    k_generate, k_interact (pass A, by material class), k_edges, k_interact_b (pass B)
//   kernels_trace.hip   k_trace_refill, k_trace_heavy (+ the per-query kernels of the traversal parity tests, the PMC calibration copy)
//   kernels_fsd.hip     k_flux_split, k_flux_tasks, k_interact_c, k_interact_c_hard (Fraunhofer interactions: power sums, rejection sampling)
//   kernels_path.hip    k_path_* (plt_path)
//   kernels_connect.hip k_connect_* (strategy buckets, connections, MIS, film splat)
// One translation unit per group: they compile in parallel (the single file took four minutes) and a kernel's registers are not at the mercy of
// its neighbours' inlining decisions.  Kernels are launched across translation units through their host-side handles (external linkage: hence
// the NAMED namespace).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstring>

#include "../../include/wtgpu.h"
#include "wt/bdpt.h"
#include "wt/coop.h"
#include "wt/coop_fsd.h"
#include "wt/path.h"

using namespace wt;

namespace wtk {


// Register budgets (second __launch_bounds__ argument = minimum waves per SIMD => 512 / n unified VGPRs per lane).
#ifndef WTGPU_LB_TRACE
#define WTGPU_LB_TRACE 3
#endif
#ifndef WTGPU_LB_HEAVY
#define WTGPU_LB_HEAVY 2   // 215 VGPRs, no spills, no scratch frame: as fast as 3 waves with 93 spilled registers, 27 GB per pass less HBM traffic
#endif
#ifndef WTGPU_LB_INTERACT
#define WTGPU_LB_INTERACT 3   // (round 6, with the bicubic texture path in the kernel: 4 / 3 / 2 waves per SIMD -> 29.1-29.3 / 29.9-30.2 / 30.2 Msamples/s; until round 5: 4)
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
constexpr uint32_t kConeBudget = 96;     // work units (1 per cone-triangle test, 2 per node) one lane may spend on a cone query before it is handed to a wavefront (with lane refill, round 3: 32 / 48 / 64 / 96 / 128 -> 14.6 / 14.8 / 15.2 / 14.4 / 12.8 Msamples/s; round 2's kernel without refill: optimum 28-32)
constexpr int kSpillStack = 64 - kLdsStack;   // scratch spill entries per lane (total 64, the reference's ray stack size)

// control block of one state slice (device memory)
enum : uint32_t { CTL_STRAT_HEAD_OPEN = 25, CTL_UTD_COUNT0 = 26, CTL_UTD_COUNT1 = 27, CTL_FSDQ_COUNT0 = 28, CTL_FSDQ_COUNT1 = 29, CTL_FSDQ_HEAD = 30, CTL_NEEQ_COUNT = 31, CTL_NEEQ_HEAD = 32, CTL_COUNT0 = 0, CTL_COUNT1 = 1, CTL_HEAD_TRACE = 2, CTL_HEAD_INTERACT = 3, CTL_HEAVY_COUNT = 4, CTL_HEAVY_HEAD = 5, CTL_FSD_COUNTER = 6,
                  CTL_ROUNDS = 7, CTL_STRAT_HEAD = 8, CTL_INTB_COUNT = 9, CTL_INTB_HEAD = 10, CTL_GATHER_COUNT = 11, CTL_GATHER_HEAD = 12, CTL_INTC_COUNT = 13, CTL_INTC_HEAD = 14, CTL_FTASK_COUNT = 15, CTL_FTASK_HEAD = 16, CTL_FSPLIT_HEAD = 17, CTL_EPOOL_COUNT = 18, CTL_FSD_ECOUNTER = 19, CTL_INTD_COUNT = 20, CTL_INTD_HEAD = 21, CTL_BACK0 = 22, CTL_BACK1 = 23,
                  CTL_TPOL_COUNT0 = 41, CTL_TPOL_COUNT1 = 42, CTL_TPOL_HEAD = 43, CTL_TCONE_COUNT0 = 44, CTL_TCONE_COUNT1 = 45, CTL_TCONE_HEAD = 46,   // the staged trace kernels' queues (trace_stage_t)
                  CTL_CLS_COUNT0 = 33, CTL_CLS_HEAD0 = 37,   // the material-sorted pass A: sizes and dequeue heads of the kNumWalkClasses class queues (k_classify / k_interact_cls)
                  CTL_LIGHT_DONE = 47, CTL_LIGHT_STOP = 48,   // k_light_rounds: whole rounds it ran, the stage the host has to continue the next one from (0: none)
                  CTL_WORDS = 56 };   // (CTL_BACK*: see queue_append)   // (CTL_GATHER_*: queue of k_edges)
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
    const struct bdpt_ext_t* ext = nullptr;   // more plt_bdpt state behind one pointer (device memory; the launch block must stay below 1 KB, see path_state_t)
};
// plt_bdpt only — like path_state_t a device-resident block that kernels reach through one pointer of the launch block.
struct bdpt_ext_t {
    // material-sorted pass A: walk class of every triangle (wt/bdpt.h: walk_class_of_triangle, built at upload) and the round's class queues
    const unsigned char* tri_class = nullptr;
    uint32_t* cls_queue = nullptr;   // [kNumWalkClasses][2 cap]
    // staged connections: the connections that wait for their shadow ray (k_connect_eval -> k_connect_shadow -> k_connect_mis)
    // The strategy items of a batch are connected in CHUNKS of pend_cap items (in bucket order), each chunk through the three kernels in turn: a
    // chunk's connections cannot outnumber its items, so the pending list cannot overflow whatever the scene's depth.  chunk_ctl: kChunkCtlWords
    // counters per chunk (zeroed by k_connect_scan).
    struct conn_pending_t* pend = nullptr;
    uint32_t* surv = nullptr;        // [pend_cap] the pending connections whose ray arrived (indices into pend)
    uint32_t* chunk_ctl = nullptr;   // [n_chunks][kChunkCtlWords]
    uint32_t pend_cap = 0, n_chunks = 0;
};
enum : uint32_t { CHUNK_EVAL_HEAD = 0, CHUNK_PEND_COUNT = 1, CHUNK_PEND_HEAD = 2, CHUNK_SURV_COUNT = 3, CHUNK_MIS_HEAD = 4, kChunkCtlWords = 8 };
struct conn_pending_t {   // 52 B
    uint32_t i;    // sample of the batch
    uint32_t st;   // s | t << 16 | kPendShadow
    float L[4];    // flux of the unoccluded connection
    float o[3], d[3], dist;   // the shadow ray
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
// The staged trace kernels (k_tr_axis / k_tr_cone / k_tr_policy / k_tr_tail, kernels_trace.hip) keep a walk's traversal state in memory between
// their stages: the traced envelope and the policy's state.  The records live BEHIND the slice's traversal records, the two queues behind its heavy
// queue (same allocations: the launch block is at its 984-byte limit, see path_state_t).
struct trace_stage_t {
    cone_t env;
    axis_walk_t aw;
};
constexpr size_t kStageWords = sizeof(trace_stage_t) / 4;
constexpr size_t kNumCounters = sizeof(bdpt_counters_t) / sizeof(unsigned long long);
constexpr size_t kProfSlots = 128;   // WTGPU_PROFILE scratch counters behind the public ones
constexpr size_t kDroppedSlot = kNumCounters + kProfSlots;   // ... and behind those: children a full cooperative traversal stack could not hold (wt/coop.h)


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
    uint32_t collect_list;    // bit 0: the cone queries keep the bounded triangle list of the interaction region; bit 1: primary triangles from the axis query always (wtgpu.hip)
};

WT_D uint32_t* trace_stage_words(const launch_args_t& a) { return a.st.trav + kTravWords * 2 * (size_t)a.st.cap; }   // [2 cap][kStageWords]
WT_D uint32_t* trace_pol_queue(const launch_args_t& a) { return a.st.heavy_queue + 2 * (size_t)a.st.cap; }
WT_D uint32_t* trace_cone_queue(const launch_args_t& a) { return a.st.heavy_queue + 4 * (size_t)a.st.cap; }
// (block size as a constant: blockDim would pull 256 bytes of hidden kernel arguments into the kernel-argument segment)
WT_D void lds_stack(stack_entry_t* lds, stack_entry_t* spill, stack_ref_t& s, uint32_t block = kBlock) {
    s = make_stack_ref(lds + threadIdx.x, block, kLdsStack + kSpillStack, kLdsStack, spill);
}

WT_D void flush_counters(unsigned long long* g, const bdpt_counters_t& c) {
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(&c);
#pragma unroll
    for (size_t i = 0; i < kNumCounters; ++i) {
        unsigned long long v = p[i];
        // wave reduction (64 lanes)
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&g[i], v);
    }
}

// walk id -> (sample index, stream)
WT_D void walk_ident(const launch_args_t& a, uint32_t w, uint32_t& i, uint32_t& stream) {
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
WT_D uint32_t queue_count(const uint32_t* ctl, int in) { return ctl[CTL_COUNT0 + in] + ctl[CTL_BACK0 + in]; }
WT_D uint32_t queue_walk(const launch_args_t& a, const uint32_t* ctl, int in, uint32_t qi, int first_round) {
    if (first_round) return qi < a.nb ? qi : (uint32_t)a.st.cap + (qi - a.nb);
    const uint32_t front = ctl[CTL_COUNT0 + in];
    return qi < front ? a.st.queue[in][qi] : a.st.queue[in][2 * (size_t)a.st.cap - 1 - (qi - front)];
}
// The next `inc` entries of a device queue for one WAVEFRONT: lane 0 moves the head, every lane gets the old value.
// Until round 5 every persistent loop began with `if (threadIdx.x == 0) word = atomicAdd(head, 1); __syncthreads(); item = word; __syncthreads();`
// (one-wavefront blocks) or the same through a shuffle, and in round 5 that stopped k_path_fsd in one of two build layouts.  What
// the compiler did, read off the ISA: the loop body ENDS with `if (threadIdx.x == 0) result[w] = f;` and BEGINS with `if (threadIdx.x == 0)
// …atomicAdd…`; the two branches on the same condition were threaded across the back edge (lane 0: store, then atomic; the others: neither), the
// join — the read of the shared word / the readfirstlane — became the header of a loop with TWO back edges, these were split into nested loops,
// and the structuriser ran the inner one (lanes 1-63, which skip the atomic) to completion while lane 0 waited outside: 63 lanes re-read the
// same item (the stale shared word, or lane 1's zero) for ever.  The barriers of a one-wavefront block compile to nothing, so nothing stood in
// the way.  Two things stand in the way here: a convergent marker IN FRONT of the branch (a block that holds one is not duplicated, and
// threading an edge through the loop header means duplicating it), and a lane index the optimiser does not connect with threadIdx.x.
// (Measured alternatives, run r5n / r5o: every lane issues the atomic, lane 0 adding `inc` and the others 0 — correct, and 1 % of a pass slower:
// the compiler's scan over lane-dependent addends costs a wavefront-per-item kernel ~1 us per item; every lane adding 1 — free, but correct
// only where the compiler folds the 64 atomics into one, which it does for pointers it knows to be global and not for the chunk words of the
// staged connections: five parity tests failed.)
WT_D bool wave_first_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u; }
WT_D uint32_t wave_grab0(uint32_t* head, uint32_t inc) {
    __builtin_amdgcn_wave_barrier();
    uint32_t old = 0;
    if (wave_first_lane()) old = atomicAdd(head, inc);
    __builtin_amdgcn_wave_barrier();
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
}
WT_D uint32_t wave_grab(uint32_t* head) { return wave_grab0(head, 64u); }        // 64 queue items, one per lane: the first lane's
WT_D uint32_t wave_grab_item(uint32_t* head) { return wave_grab0(head, 1u); }    // one queue item for a one-wavefront block
// One value from lane 0 of a one-wavefront block to all of its lanes, for values that only lane 0 may compute (an allocation).  Call sites keep
// a convergent operation (a barrier, a shuffle) between this and any earlier `if (threadIdx.x == 0)`, see above.
WT_D uint32_t wave_bcast0(uint32_t v) {
    __builtin_amdgcn_wave_barrier();
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// wave-aggregated append of `w` (for lanes with `pred`) to a device queue
WT_D void wave_append(uint32_t* queue, uint32_t* count, bool pred, uint32_t w) {
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
WT_D void queue_append(const launch_args_t& a, uint32_t* ctl, int out, bool pred, uint32_t w) {
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

// ---- wave-cooperative record transfer -------------------------------------------------------------------------------------------------------
// The per-walk records are record-major in memory (one contiguous record per walk) and the walks a wavefront holds are scattered over the batch.
// A lane that loads its own record word by word makes every load INSTRUCTION touch 64 different cache lines for 4 bytes each (57 instructions x
// 64 lines for a walk); a lane that stores its record word by word leaves 64 partially written lines behind every instruction, which the L2
// evicts before the other 85 words of a vertex arrive (measured, run r5g: k_interact moves 82 GB per step for 13 GB of records).  Here the
// wavefront moves the records of its lanes ONE RECORD AT A TIME, consecutive lanes on consecutive words (a 228-byte walk = one instruction
// touching 2-3 lines), through an LDS buffer of kIoRows rows that the owning lanes then read / have written row-wise (odd pitch: no bank
// conflicts): 64 x 3 line visits instead of 57 x 64.  Half a wavefront at a time, so that the buffer is 11 KB per wavefront.
constexpr int kIoRows = 32;
template <int W>
constexpr int io_pitch() { return (W & 1) ? W : W + 1; }
// `idx`: the lane's record, `valid`: the lane takes part; lds: this wavefront's buffer (>= kIoRows * io_pitch<W>() words)
template <int W, class T>
WT_D void wave_load_records(const uint32_t* base, size_t stride, uint32_t idx, bool valid, uint32_t* lds, T& out) {
    static_assert(sizeof(T) == 4 * W, "record size");
    constexpr int P = io_pitch<W>();
    const int lane = threadIdx.x & 63;
    uint32_t* o = reinterpret_cast<uint32_t*>(&out);
    const unsigned long long all = __ballot(valid);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        unsigned long long m = all & (h ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull);
        while (m) {
            const int r = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t* src = base + (size_t)__builtin_amdgcn_readlane(idx, r) * stride;
#pragma unroll
            for (int w0 = 0; w0 < W; w0 += 64)
                if (w0 + lane < W) lds[(r & 31) * P + w0 + lane] = src[w0 + lane];
        }
        __builtin_amdgcn_wave_barrier();
        if (valid && (lane >> 5) == h) {
#pragma unroll
            for (int i = 0; i < W; ++i) o[i] = lds[(lane & 31) * P + i];
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// ... `off`: word offset of the lane's record behind base + idx * stride (a vertex of the walk's vertex array)
template <int W, class T>
WT_D void wave_store_records(uint32_t* base, size_t stride, uint32_t idx, uint32_t off, bool valid, uint32_t* lds, const T& in) {
    static_assert(sizeof(T) == 4 * W, "record size");
    constexpr int P = io_pitch<W>();
    const int lane = threadIdx.x & 63;
    const uint32_t* o = reinterpret_cast<const uint32_t*>(&in);
    const unsigned long long all = __ballot(valid);
    if (!all) return;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        if (valid && (lane >> 5) == h) {
#pragma unroll
            for (int i = 0; i < W; ++i) lds[(lane & 31) * P + i] = o[i];
        }
        __builtin_amdgcn_wave_barrier();
        unsigned long long m = all & (h ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull);
        while (m) {
            const int r = __ffsll((long long)m) - 1;
            m &= m - 1;
            uint32_t* dst = base + (size_t)__builtin_amdgcn_readlane(idx, r) * stride + __builtin_amdgcn_readlane(off, r);
#pragma unroll
            for (int w0 = 0; w0 < W; w0 += 64)
                if (w0 + lane < W) dst[w0 + lane] = lds[(r & 31) * P + w0 + lane];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- kernels (definitions: kernels_*.hip)
constexpr uint32_t kFlushGrid = 64;          // k_path_flush
constexpr int kEnumBlock = 1024;             // k_connect_enum
constexpr uint32_t kSplatCols = kBlock + 2;  // k_connect_splat_tiled: columns of its LDS tile
#ifndef WTGPU_HARD_BLOCK
#define WTGPU_HARD_BLOCK 256
#endif
__global__ void k_generate(launch_args_t a);
__global__ void k_trace_refill(launch_args_t a, int in, int first_round, uint32_t round);
__global__ void k_trace_sm(launch_args_t a, int in, int first_round, uint32_t round);
__global__ void k_tr_axis(launch_args_t a, int in, int first_round, uint32_t round);
__global__ void k_tr_cone(launch_args_t a, uint32_t it);
__global__ void k_tr_policy(launch_args_t a, uint32_t it);
__global__ void k_tr_tail(launch_args_t a, uint32_t it);
__global__ void k_trace_heavy(launch_args_t a);
__global__ void k_interact(launch_args_t a, int in, int first_round);
__global__ void k_classify(launch_args_t a, int in, int first_round);
__global__ void k_interact_diffuse(launch_args_t a, int in);
__global__ void k_interact_dielectric(launch_args_t a, int in);
__global__ void k_interact_spm(launch_args_t a, int in);
__global__ void k_interact_any(launch_args_t a, int in);
__global__ void k_interact_sorted(launch_args_t a, int in);
__global__ void k_interact_coop(launch_args_t a, int in, int first_round);
__global__ void k_edges(launch_args_t a);
__global__ void k_interact_b(launch_args_t a, int in);
__global__ void k_light_rounds(launch_args_t a, int in, uint32_t round, uint32_t max_rounds);
__global__ void k_flux_split(launch_args_t a);
__global__ void k_flux_tasks(launch_args_t a);
__global__ void k_interact_c(launch_args_t a, int in);
__global__ void k_interact_c_hard(launch_args_t a, int in);
__global__ void k_path_generate(launch_args_t a);
__global__ void k_path_fsd(launch_args_t a, const path_state_t* __restrict__ ps, uint32_t round);
__global__ void k_path_edges(launch_args_t a, const path_state_t* __restrict__ ps);
__global__ void k_path_interact(launch_args_t a, const path_state_t* __restrict__ ps, int in, int first_round, uint32_t round);
__global__ void k_path_interact_b(launch_args_t a, const path_state_t* __restrict__ ps, int in, uint32_t round);
__global__ void k_path_nee(launch_args_t a, const path_state_t* __restrict__ ps, uint32_t round);
__global__ void k_path_flush(launch_args_t a, int in);
__global__ void k_connect_enum(launch_args_t a);
__global__ void k_connect_scan(launch_args_t a);
__global__ void k_connect_strat(launch_args_t a);
__global__ void k_connect_strat_open(launch_args_t a);
__global__ void k_connect_eval(launch_args_t a, uint32_t chunk);
__global__ void k_connect_shadow(launch_args_t a, uint32_t chunk);
__global__ void k_connect_mis(launch_args_t a, uint32_t chunk);
__global__ void k_connect_splat(launch_args_t a);
__global__ void k_connect_splat_tiled(launch_args_t a);
__global__ void k_calib_copy(const uint32_t* in, uint32_t* out, size_t n);
__global__ void k_trace_rays(scene_t sc, const float* rays, uint32_t n, float* dist, uint32_t* tuid, float* bary, uint32_t* front);
__global__ void k_traverse_cones(scene_t sc, const float* cones, uint32_t n, uint32_t cap, float* dist, uint32_t* flags, uint32_t* ntris, uint32_t* out_tris,
                                 uint32_t* scratch_tris);
__global__ void k_query_regions(scene_t sc, const float* cones, uint32_t n, uint32_t edge_cap, float* dist, uint32_t* flags, uint32_t* primary, uint32_t* ntris,
                                uint32_t* nedges, uint32_t* edges, float* flux, unsigned long long* dropped);

}   // namespace wtk
using namespace wtk;
