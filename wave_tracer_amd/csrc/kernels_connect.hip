// wave_tracer_amd — subpath connections of plt_bdpt: strategy buckets, connections + MIS, film splat (see wtgpu_kernels.h for the list of kernel translation units).
#include "wtgpu_kernels.h"

namespace wtk {

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
WT_D bool strategy_valid(const integrator_opts_t& o, int s, int t, int nS, int nT) {
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
WT_D bool strategy_class_valid(const integrator_opts_t& o, int sk, int tk, int nS, int nT) {
    const int K = (int)kKeyDim - 1;
    const int t1 = tk < K ? tk : nT, s1 = sk < K ? sk : nS;
    for (int t = tk; t <= t1; ++t)
        for (int s = sk; s <= s1; ++s)
            if (strategy_valid(o, s, t, nS, nT)) return true;
    return false;
}
WT_D int wave_max_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

// Block-aggregated bucket append: 1024 samples per block count their valid (s,t) pairs per bucket in LDS, reserve one range per
// bucket with ONE global atomic each, and fill it.  (Wave-aggregated global atomics on the ~30 hot bucket counters serialised in
// L2: PMC SQ_WAIT_ANY 99 % of this kernel's wave cycles, 9.5 ms per pass.)
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
    if (a.st.ext)   // the counters of the staged connections' chunks
        for (uint32_t q = threadIdx.x; q < a.st.ext->n_chunks * kChunkCtlWords; q += 64) a.st.ext->chunk_ctl[q] = 0;
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
WT_D void connect_strat_body(const launch_args_t& a) {
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
// ---- staged connections (the default; k_connect_strat, one kernel for the whole strategy, stays as the A/B reference: WTGPU_STAGED_CONNECT=0) ---
// A connection is three things with very different shapes: forming the two connecting beams (two vertex loads, two BSDF evaluations, ~200
// registers), one any-hit ray (a BVH stack, ~90 registers, a run time that varies by two orders of magnitude), and the MIS weight (a stream
// over both subpaths).  One kernel for all three (k_connect_strat) carries the union of their registers — 256 + a 1.5-KB frame, two wavefronts
// per SIMD — and runs the ray with whatever lanes still have a connection.  Here:
//   k_connect_eval    lane / (sample, s, t), items bucketed by strategy as before: bdpt_connect<DEFER> — everything of connect_subpaths but the
//                     ray; connections with flux > 0 go into the pending list (52 B: sample, (s,t), flux, ray).  The items of a batch are taken
//                     in CHUNKS of as many items as the list has records, each chunk through the three kernels in turn: the list cannot
//                     overflow, whatever the depth of the scene (bdpt_ext_t);
//   k_connect_shadow  lane / pending connection, ALL lanes: the any-hit ray (src/ads/bvh8w.cpp:556-603), survivors compacted;
//   k_connect_mis     lane / survivor: the temporary vertex of the s = 1 / t = 1 / virtual-sensor strategies formed again from the same random
//                     numbers (bdpt_connect_temp), streaming MIS weight (plt_bdpt_detail.hpp:604-720), flux sum or light-image splat.
// Same functions, same numbers as the one-piece form (the CPU checker runs both: tests/test_oracle.py::test_staged_connections_are_the_connections).
constexpr uint32_t kPendShadow = 0x80000000u;   // conn_pending_t::st: the connection waits for its ray (t in bits 16..30, s in bits 0..15)
WT_D void pending_append(const launch_args_t& a, uint32_t* cctl, bool keep, uint32_t i, int s, int t, const connect_ret_t& cr) {
    const bdpt_ext_t& x = *a.st.ext;
    const unsigned long long m = __ballot(keep);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(cctl + CHUNK_PEND_COUNT, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (!keep || slot >= x.pend_cap) return;   // (cannot happen: a chunk holds pend_cap items, each yields at most one connection)
    conn_pending_t r;
    r.i = i;
    r.st = (uint32_t)s | ((uint32_t)t << 16) | (cr.need_shadow ? kPendShadow : 0u);
#pragma unroll
    for (int c = 0; c < 4; ++c) r.L[c] = cr.L.s[c];
    r.o[0] = cr.ray.o.x, r.o[1] = cr.ray.o.y, r.o[2] = cr.ray.o.z;
    r.d[0] = cr.ray.d.x, r.d[1] = cr.ray.d.y, r.d[2] = cr.ray.d.z;
    r.dist = cr.ray.dist;
    x.pend[slot] = r;
}
#ifndef WTGPU_LB_EVAL
#define WTGPU_LB_EVAL 2
#endif
__global__ void __launch_bounds__(kBlock, WTGPU_LB_EVAL) k_connect_eval(launch_args_t a, uint32_t chunk) {
    __shared__ uint32_t s_prefix[kNumKeys + 1];
    for (uint32_t k = threadIdx.x; k <= kNumKeys; k += kBlock) s_prefix[k] = a.st.strat_prefix[k];
    __syncthreads();
    const bdpt_ext_t& x = *a.st.ext;
    uint32_t* cctl = x.chunk_ctl + (size_t)chunk * kChunkCtlWords;
    const uint32_t total = s_prefix[kNumKeys];
    const uint64_t begin = (uint64_t)chunk * x.pend_cap;
    if (begin >= total) return;   // (the host launches the chunks a batch of this depth can need at most: most are empty)
    const uint32_t span = (uint32_t)min<uint64_t>(x.pend_cap, total - begin);
    bdpt_counters_t ctr;
    ctr.connections = ctr.shadow_rays = 0;
    const stack_ref_t no_stack = make_stack_ref(nullptr, 0, 0, 0, nullptr);   // (bdpt_connect<true> traces nothing)
    const fsd_pool_t pool{a.st.fsd_hdr, a.st.fsd_edges, a.st.ctl + CTL_FSD_COUNTER, a.st.fsd_cap, a.st.ctl + CTL_FSD_ECOUNTER, a.st.fsd_ecap};
    for (;;) {
        const uint32_t off = wave_grab(cctl + CHUNK_EVAL_HEAD) + (threadIdx.x & 63);
        if (off - (threadIdx.x & 63) >= span) break;
        if (off < span) {
            const uint32_t idx = (uint32_t)begin + off;
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
            const int t = (int)(key / kKeyDim), s = (int)(key % kKeyDim);
            const uint32_t i = a.st.strat_items[(size_t)key * a.st.cap + (idx - s_prefix[key])];
            const uint64_t j = a.j0 + i;
            const uint32_t pix = (uint32_t)(j % a.npix);
            const uint64_t smp = a.sample_begin + j / a.npix;
            const uint64_t sample_id = ((uint64_t)pix << 32) | (smp & 0xFFFFFFFFull);
            const vertex_store_t svs{a.st.verts, a.st.vert_words, i}, evs{a.st.verts, a.st.vert_words, (size_t)a.st.cap + i};
            connect_ret_t cr;
            bdpt_connect<true>(a.sc, pool, svs, evs, s, t, a.seed, sample_id, no_stack, cr, &ctr, nullptr);
            pending_append(a, cctl, cr.L.s[0] > 0.f, i, s, t, cr);
        }
    }
    if (a.count_stats) {
        unsigned long long vc = ctr.connections, vr = ctr.shadow_rays;
        for (int off = 32; off > 0; off >>= 1) {
            vc += __shfl_down(vc, off, 64);
            vr += __shfl_down(vr, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            if (vc) atomicAdd(a.st.counters + offsetof(bdpt_counters_t, connections) / sizeof(unsigned long long), vc);
            if (vr) atomicAdd(a.st.counters + offsetof(bdpt_counters_t, shadow_rays) / sizeof(unsigned long long), vr);
        }
    }
}

__global__ void __launch_bounds__(kBlock, 4) k_connect_shadow(launch_args_t a, uint32_t chunk) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack);
    const bdpt_ext_t& x = *a.st.ext;
    uint32_t* cctl = x.chunk_ctl + (size_t)chunk * kChunkCtlWords;
    const uint32_t n = min(cctl[CHUNK_PEND_COUNT], x.pend_cap);
    if (n == 0) return;
    for (;;) {
        const uint32_t idx = wave_grab(cctl + CHUNK_PEND_HEAD) + (threadIdx.x & 63);
        if (idx - (threadIdx.x & 63) >= n) break;
        bool alive = false;
        if (idx < n) {
            const conn_pending_t& r = x.pend[idx];
            alive = true;
            if (r.st & kPendShadow) alive = !ads_shadow_ray(a.sc, vec3{r.o[0], r.o[1], r.o[2]}, vec3{r.d[0], r.d[1], r.d[2]}, range_t{0.f, r.dist}, stack, nullptr);
        }
        wave_append(x.surv, cctl + CHUNK_SURV_COUNT, alive, idx);
    }
}

#ifndef WTGPU_LB_MIS
#define WTGPU_LB_MIS 2
#endif
__global__ void __launch_bounds__(kBlock, WTGPU_LB_MIS) k_connect_mis(launch_args_t a, uint32_t chunk) {
    const bdpt_ext_t& x = *a.st.ext;
    uint32_t* ctl = a.st.ctl;
    uint32_t* cctl = x.chunk_ctl + (size_t)chunk * kChunkCtlWords;
    const uint32_t n = cctl[CHUNK_SURV_COUNT];
    if (n == 0) return;
    const fsd_pool_t pool{a.st.fsd_hdr, a.st.fsd_edges, ctl + CTL_FSD_COUNTER, a.st.fsd_cap, ctl + CTL_FSD_ECOUNTER, a.st.fsd_ecap};
    bdpt_counters_t ctr;
    ctr.light_splats = 0;
    for (;;) {
        const uint32_t k = wave_grab(cctl + CHUNK_MIS_HEAD) + (threadIdx.x & 63);
        if (k - (threadIdx.x & 63) >= n) break;
        if (k < n) {   // (no divergent `continue` in front of the loop header's grab: wave_grab0 assumes a converged wavefront)
            const conn_pending_t& r = x.pend[x.surv[k]];
            const uint32_t i = r.i, st = r.st;
            const int s = (int)(st & 0xFFFFu), t = (int)((st >> 16) & 0x7FFFu);
            stokes_t L;
    #pragma unroll
            for (int c = 0; c < 4; ++c) L.s[c] = r.L[c];
            const uint64_t j = a.j0 + i;
            const uint32_t pix = (uint32_t)(j % a.npix);
            const uint64_t smp = a.sample_begin + j / a.npix;
            const uint64_t sample_id = ((uint64_t)pix << 32) | (smp & 0xFFFFFFFFull);
            sample_ctx_t ctx;
            soa_load(a.st.ctx, kCtxWords, i, ctx);
            const vertex_store_t svs{a.st.verts, a.st.vert_words, i}, evs{a.st.verts, a.st.vert_words, (size_t)a.st.cap + i};
            vertex_nb_t tv;
            sensor_element_t element;
            bool has_element = false;
            if (s <= 1 || t <= 1) bdpt_connect_temp(a.sc, svs, evs, s, t, a.seed, sample_id, tv, element, has_element);
            const stokes_t flux = bdpt_strategy_finish(a.sc, pool, a.film, svs, evs, s, t, ctx, L, tv, element, has_element, &ctr);
            if (t > 1) {
    #pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (flux.s[c] != 0.f) unsafeAtomicAdd(&a.st.lacc[(size_t)c * a.st.cap + i], (double)flux.s[c]);
            }
        }
    }
    if (a.count_stats) {
        unsigned long long v = ctr.light_splats;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(a.st.counters + offsetof(bdpt_counters_t, light_splats) / sizeof(unsigned long long), v);
    }
}

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

}   // namespace wtk
