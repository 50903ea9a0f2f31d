// wave_tracer_amd — generation and the per-lane interaction passes A / B of plt_bdpt, the classified-edge gather (k_edges) (see wtgpu_kernels.h for the list of kernel translation units).
#include "wtgpu_kernels.h"
#include "kernels_trace_refill.h"

namespace wtk {

__global__ void __launch_bounds__(kBlock) k_generate(launch_args_t a) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) {
        uint32_t* ctl = a.st.ctl;
        ctl[CTL_COUNT0] = 2 * a.nb;
        ctl[CTL_COUNT1] = 0;
        ctl[CTL_BACK0] = ctl[CTL_BACK1] = 0;
        ctl[CTL_HEAD_TRACE] = ctl[CTL_HEAD_INTERACT] = ctl[CTL_HEAVY_COUNT] = ctl[CTL_HEAVY_HEAD] = ctl[CTL_FSD_COUNTER] = ctl[CTL_ROUNDS] = 0;
        ctl[CTL_TPOL_COUNT0] = ctl[CTL_TPOL_COUNT1] = ctl[CTL_TPOL_HEAD] = ctl[CTL_TCONE_COUNT0] = ctl[CTL_TCONE_COUNT1] = ctl[CTL_TCONE_HEAD] = 0;   // the staged trace kernels' queues
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

// Interaction step of the queued walks.  PASS 0 (A): the queue of the round — surface interactions; walks whose beam axis misses
// every triangle of the interaction region (8 % of them; what follows costs ~50x a surface interaction) are only appended to the
// pass-B queue.  k_edges then gathers the classified-edge set of their regions.  PASS 1 (B): Fraunhofer aperture construction,
// null interactions; the one walk in eight whose aperture has edges goes on to the pass-C queue (k_interact_c).
// No BVH query happens in these passes (the trace kernels resolved the primary triangle): they carry no traversal stack.
// COOP: the walk and traversal records come in, and the appended vertex and the walk record go out, by wave-cooperative transfers through LDS
// (wtgpu_kernels.h: wave_load_records / wave_store_records) instead of lane by lane.
template <int PASS, bool COOP = false>
WT_D void interact_body(const launch_args_t& a, int in, int first_round) {
    constexpr bool PASS_B = PASS == 1;
    __shared__ uint32_t s_io[COOP ? (kBlock / 64) * kIoRows * io_pitch<(int)kVertexWords>() : 1];
    uint32_t* io = s_io + (COOP ? (threadIdx.x >> 6) * kIoRows * io_pitch<(int)kVertexWords>() : 0);
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
    // (Taking 4 x 64 / 16 x 64 queue items per atomic on the queue's head, to see whether the hot-address atomics of the persistent loops hold
    // the kernel up: exclusive time of k_interact 95.4 -> 100.9 / 163.6 ms per three steps, run r5j — they do not; larger grabs only unbalance the tail.)
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
        const bool valid = qi < n;
        if (valid) w = PASS_B ? a.st.intb_queue[qi] : queue_walk(a, ctl, in, qi, first_round);
        walk_t wk;
        trav_result_t tr;
        if constexpr (COOP) {
            wave_load_records<(int)kWalkWords>(a.st.walks, a.st.walk_words, w, valid, io, wk);
            wave_load_records<(int)kTravWords>(a.st.trav, kTravWords, w, valid, io, tr);
        }
        vertex_t staged;
        uint32_t staged_idx = kInvalid;
        bool store_walk = false;
        if (valid) {
            uint32_t i, stream;
            walk_ident(a, w, i, stream);
            const uint64_t j = a.j0 + i;
            const uint32_t pix = (uint32_t)(j % a.npix);
            const uint64_t s = a.sample_begin + j / a.npix;
            const uint64_t sample_id = ((uint64_t)pix << 32) | (s & 0xFFFFFFFFull);
            if constexpr (!COOP) {
                soa_load(a.st.walks, a.st.walk_words, w, wk);
                soa_load(a.st.trav, kTravWords, w, tr);
            }
            const uint_list_t tris{a.st.tris + (size_t)w * kTriListWords, 1u, kMaxConeTris};
            const vertex_store_t vs{a.st.verts, a.st.vert_words, w, COOP ? &staged : nullptr, COOP ? &staged_idx : nullptr};
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
            if (!PASS_B && defer.no_primary && !tr.ballistic && a.sc.opts.FSD && (tr.overflow > 0 || tr.ntris > kMaxEdgeIds / 3 || !(a.collect_list & 1u))) need_gather = true;
            if (!defer.no_primary && !defer.to_sampling_pass && !queued_for_c) {
                wk.active = cont ? 1u : 0u;
                store_walk = true;
                if constexpr (!COOP) soa_store(a.st.walks, a.st.walk_words, w, wk);
            }
        }
        if constexpr (COOP) {
            wave_store_records<(int)kVertexWords>(a.st.verts, a.st.vert_words, w, staged_idx == kInvalid ? 0u : staged_idx * (uint32_t)kVertexWords, valid && staged_idx != kInvalid, io, staged);
            wave_store_records<(int)kWalkWords>(a.st.walks, a.st.walk_words, w, 0u, store_walk, io, wk);
        }
        if (!PASS_B) wave_append(a.st.intb_queue, ctl + CTL_INTB_COUNT, defer.no_primary != 0, w);
        if (!PASS_B) wave_append(a.st.gather_queue, ctl + CTL_GATHER_COUNT, need_gather, w);
        if (PASS_B) wave_append(a.st.intc_queue, ctl + CTL_INTC_COUNT, defer.to_sampling_pass != 0, w);
        queue_append(a, ctl, 1 - in, cont, w);
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}

// ---- pass A, material-sorted (WTGPU_SORTED_INTERACT=1|2; the default is k_interact above, one kernel for every walk: DESIGN.md §4) ------------------------------
// k_classify (lane / walk of the round's queue): the primary triangle (wt/bdpt.h: bdpt_classify — ballistic hit, the trace kernels' axis hit of an
// overflowed region, or a scan of the region's list) goes back into the walk's traversal record, and the walk goes into the queue of its CLASS — the
// BSDF type of the hit shape's material (one byte per triangle, built at upload), WCLS_ANY for wrapped materials — or, without a primary triangle,
// into pass B's queue (and k_edges').  41 registers, no frame: the dependent loads of this part (record -> list -> triangles) run at full occupancy.
// k_interact_<class> (lane / walk of one class queue): bdpt_surface_step<CLS> — one BSDF's code per kernel (src/bsdf/diffuse.cpp:23-71,
// dielectric.cpp:26-72, surface_spm.cpp:40-201), the walk record and the new vertex read / written field by field.
__global__ void __launch_bounds__(kBlock) k_classify(launch_args_t a, int in, int first_round) {
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = queue_count(ctl, in);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl[CTL_HEAVY_COUNT] = 0;   // for the next round's k_trace
        ctl[CTL_HEAVY_HEAD] = 0;
        ctl[CTL_HEAD_TRACE] = 0;
    }
    const bdpt_ext_t& x = *a.st.ext;
    const size_t W2 = 2 * (size_t)a.st.cap;
    for (;;) {
        const uint32_t qi = wave_grab(ctl + CTL_HEAD_INTERACT) + (threadIdx.x & 63);
        if (qi - (threadIdx.x & 63) >= n) break;
        uint32_t cls = WCLS_END, w = 0;
        bool need_gather = false;
        if (qi < n) {
            w = queue_walk(a, ctl, in, qi, first_round);
            trav_result_t tr;
            soa_load(a.st.trav, kTravWords, w, tr);
            const uint32_t* wp = a.st.walks + (size_t)w * a.st.walk_words;
            const vec3 d{__uint_as_float(wp[WT_WALK_WORD(beam.env.d.x)]), __uint_as_float(wp[WT_WALK_WORD(beam.env.d.y)]), __uint_as_float(wp[WT_WALK_WORD(beam.env.d.z)])};
            const bool beam_ray = __uint_as_float(wp[WT_WALK_WORD(beam.env.tan_alpha)]) == 0.f && __uint_as_float(wp[WT_WALK_WORD(beam.env.x0)]) == 0.f;
            const uint_list_t tris{a.st.tris + (size_t)w * kTriListWords, 1u, kMaxConeTris};
            primary_hit_t ph;
            cls = bdpt_classify(a.sc, d, beam_ray, tr, tris, x.tri_class, ph);
            if (cls < kNumWalkClasses) {
                uint32_t* tv = a.st.trav + (size_t)w * kTravWords;
                tv[WT_TRAV_WORD(tuid)] = ph.tuid;
                tv[WT_TRAV_WORD(bx)] = __float_as_uint(ph.bx);
                tv[WT_TRAV_WORD(by)] = __float_as_uint(ph.by);
                tv[WT_TRAV_WORD(pdist)] = __float_as_uint(ph.dist);
            }
            // a region that did not fit the bounded list (or whose list holds more than kMaxEdgeIds / 3 triangles): its edge set comes from k_edges
            need_gather = cls == WCLS_NO_PRIMARY && !tr.ballistic && a.sc.opts.FSD && (tr.overflow > 0 || tr.ntris > kMaxEdgeIds / 3 || !(a.collect_list & 1u));
        }
#pragma unroll
        for (uint32_t c = 0; c < kNumWalkClasses; ++c) wave_append(x.cls_queue + (size_t)c * W2, ctl + CTL_CLS_COUNT0 + c, cls == c, w);
        wave_append(a.st.intb_queue, ctl + CTL_INTB_COUNT, cls == WCLS_NO_PRIMARY, w);
        wave_append(a.st.gather_queue, ctl + CTL_GATHER_COUNT, need_gather, w);
    }
}
template <int CLS>
WT_D void interact_cls_body(const launch_args_t& a, int in) {
    constexpr uint32_t cls = CLS < 0 ? (uint32_t)WCLS_ANY : (uint32_t)CLS;
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_CLS_COUNT0 + cls];
    const uint32_t* queue = a.st.ext->cls_queue + (size_t)cls * 2 * (size_t)a.st.cap;
    bdpt_counters_t ctr;
    ctr.surface_interactions = ctr.vertices = 0;
    for (;;) {
        const uint32_t qi = wave_grab(ctl + CTL_CLS_HEAD0 + cls) + (threadIdx.x & 63);
        if (qi - (threadIdx.x & 63) >= n) break;
        bool cont = false;
        uint32_t w = 0;
        if (qi < n) {
            w = queue[qi];
            uint32_t i, stream;
            walk_ident(a, w, i, stream);
            const uint64_t j = a.j0 + i;
            const uint32_t pix = (uint32_t)(j % a.npix);
            const uint64_t s = a.sample_begin + j / a.npix;
            const uint64_t sample_id = ((uint64_t)pix << 32) | (s & 0xFFFFFFFFull);
            const uint32_t* tv = a.st.trav + (size_t)w * kTravWords;
            const vec3 origin{__uint_as_float(tv[WT_TRAV_WORD(origin.x)]), __uint_as_float(tv[WT_TRAV_WORD(origin.y)]), __uint_as_float(tv[WT_TRAV_WORD(origin.z)])};
            const float beam_dist = __uint_as_float(tv[WT_TRAV_WORD(dist)]);
            const primary_hit_t ph{tv[WT_TRAV_WORD(tuid)], __uint_as_float(tv[WT_TRAV_WORD(pdist)]), __uint_as_float(tv[WT_TRAV_WORD(bx)]), __uint_as_float(tv[WT_TRAV_WORD(by)])};
            const walk_rec_t wr{a.st.walks + (size_t)w * a.st.walk_words};
            const vertex_store_t vs{a.st.verts, a.st.vert_words, w};
            cont = bdpt_surface_step<CLS>(a.sc, wr, origin, beam_dist, ph, vs, a.seed, sample_id, stream, &ctr);
        }
        queue_append(a, ctl, 1 - in, cont, w);
    }
    if (a.count_stats) {
        unsigned long long vsi = ctr.surface_interactions, vv = ctr.vertices;
        for (int off = 32; off > 0; off >>= 1) {
            vsi += __shfl_down(vsi, off, 64);
            vv += __shfl_down(vv, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            if (vsi) atomicAdd(a.st.counters + offsetof(bdpt_counters_t, surface_interactions) / sizeof(unsigned long long), vsi);
            if (vv) atomicAdd(a.st.counters + offsetof(bdpt_counters_t, vertices) / sizeof(unsigned long long), vv);
        }
    }
}
#ifndef WTGPU_LB_CLS
#define WTGPU_LB_CLS 3
#endif
#ifndef WTGPU_LB_CLS_SPM
#define WTGPU_LB_CLS_SPM 3
#endif
// ... and all four in ONE launch (the default): a block starts with the class its index names — in proportion to what the classes typically hold — and,
// once that queue is empty, goes on to the next (every block visits every class, so every queue is drained whatever the mix).  Four launches in a
// row each pay the ramp-up and the tail of a persistent grid on 256 CUs — per round, 40 rounds a batch — and cannot overlap one another on their
// stream: measured (run r5a) the four-kernel form lost what the sorting gained.  Here a wavefront still runs ONE class at a time (its own code, no
// divergence across BSDFs); the kernel carries the registers of the widest class.
#ifndef WTGPU_LB_SORTED
#define WTGPU_LB_SORTED 2
#endif
__global__ void __launch_bounds__(kBlock, WTGPU_LB_SORTED) k_interact_sorted(launch_args_t a, int in) {
    static constexpr unsigned char start_of[8] = {0, 0, 2, 0, 1, 0, 2, 3};
    const uint32_t first = start_of[blockIdx.x & 7u];
#pragma unroll 1
    for (uint32_t k = 0; k < kNumWalkClasses; ++k) {
        switch ((first + k) & 3u) {
        case WCLS_DIFFUSE: interact_cls_body<MAT_DIFFUSE>(a, in); break;
        case WCLS_DIELECTRIC: interact_cls_body<MAT_DIELECTRIC>(a, in); break;
        case WCLS_SPM: interact_cls_body<MAT_SURFACE_SPM>(a, in); break;
        default: interact_cls_body<-1>(a, in); break;
        }
    }
}
__global__ void __launch_bounds__(kBlock, WTGPU_LB_CLS) k_interact_diffuse(launch_args_t a, int in) { interact_cls_body<MAT_DIFFUSE>(a, in); }
__global__ void __launch_bounds__(kBlock, WTGPU_LB_CLS) k_interact_dielectric(launch_args_t a, int in) { interact_cls_body<MAT_DIELECTRIC>(a, in); }
__global__ void __launch_bounds__(kBlock, WTGPU_LB_CLS_SPM) k_interact_spm(launch_args_t a, int in) { interact_cls_body<MAT_SURFACE_SPM>(a, in); }
__global__ void __launch_bounds__(kBlock, WTGPU_LB_CLS_SPM) k_interact_any(launch_args_t a, int in) { interact_cls_body<-1>(a, in); }

// The classified-edge set of the interaction regions that overflowed the bounded triangle list: the WHOLE region, whatever its
// triangle count — the reference's unbounded std::vector (include/wt/ads/traversal_common.hpp:124-148).  One wavefront per walk
// (coop_gather, edges only): only subtrees that hold classified edges are entered and only edge-bearing triangles are tested, 64 at
// a time (a wide beam over the whole scene still meets ~10^3 of them: a single lane needs milliseconds for that).
// Sorted ids -> the walk's list slot, marker + count -> its traversal record.
__global__ void __launch_bounds__(64, 3) k_edges(launch_args_t a) {
    __shared__ coop_gather_shared_t sh;
    __shared__ coop_edges_t eg;
    coop_set_dropped_counter(sh, a.st.counters + kDroppedSlot);
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_GATHER_COUNT];
    const size_t W2 = 2 * (size_t)a.st.cap;
    const fsd_pool_t pool{a.st.fsd_hdr, a.st.fsd_edges, ctl + CTL_FSD_COUNTER, a.st.fsd_cap, ctl + CTL_FSD_ECOUNTER, a.st.fsd_ecap};
    for (;;) {
        const uint32_t item = wave_grab_item(ctl + CTL_GATHER_HEAD);
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
        off = wave_grab0(ctl + CTL_EPOOL_COUNT, n_edges);
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
            if (threadIdx.x == 0) slot = fsd_pool_alloc(pool);
            slot = wave_bcast0(slot);
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
#ifndef WTGPU_LB_INTERACT_COOP
#define WTGPU_LB_INTERACT_COOP 3
#endif
__global__ void __launch_bounds__(kBlock, WTGPU_LB_INTERACT_COOP) k_interact_coop(launch_args_t a, int in, int first_round) { interact_body<0, true>(a, in, first_round); }
__global__ void __launch_bounds__(kBlock, WTGPU_LB_INTERACT_B) k_interact_b(launch_args_t a, int in) { interact_body<1>(a, in, 0); }

#ifndef WTGPU_LIGHT_MAX_WALKS
#define WTGPU_LIGHT_MAX_WALKS 2048
#endif
constexpr uint32_t kLightMaxWalks = WTGPU_LIGHT_MAX_WALKS;
// ---- LIGHT ROUNDS: many rounds of a (nearly) empty queue in ONE launch.
// A handful of walks of some scenes outlive their batch by THOUSANDS of rounds (bidir_room: beams that graze a finely tessellated object restart behind
// empty apertures 1800-3800 times; the reference's walk has no iteration cap, plt_bdpt_detail.hpp:421-526, and until round 6 such walks were dropped
// after 96 rounds).  Run as rounds they cost nine launches each, paced by the host: 240 ms of host time per batch, 36 -> 16 Msamples/s.  What those
// rounds DO is three lane-per-walk stages — trace, pass A, pass B (a null interaction is committed by pass B) — on a dozen walks.  Here ONE block
// runs exactly those three stage bodies, round after round, with a block barrier where the stream order stood, until the queue is empty, `max_rounds`
// are done, or a walk needs a stage this kernel does not hold — the wave-cooperative traversal (heavy queue), the whole-region edge walk (k_edges),
// the Fraunhofer sampling passes: then it stops BEFORE that stage, says which (CTL_LIGHT_STOP), and the host continues that round with the ordinary
// kernels (wtgpu.hip: render_finish_part).  Same stage code, same order per walk: the same results as rounds.
// Between two stages of a light round: what the lanes of this block wrote must be read by the lanes of this block — ONE block, one CU, one XCD.
// The block barrier orders the stores (written through to the XCD's L2) and an agent-scope ACQUIRE drops the CU's L1 lines that queue counters updated
// by L2 atomics would otherwise be read from; the RELEASE half of __threadfence() — a write-back of the XCD's whole L2 (buffer_wbl2: microseconds,
// MI355X_MICROARCH.md), needed only for readers on other XCDs — is what a round does not pay three times (WTGPU_LIGHT_FENCE=1: the full fence; bidir_room 30.4 -> 30.9 Msamples/s, the same films).
#ifndef WTGPU_LIGHT_FENCE
#define WTGPU_LIGHT_FENCE 0
#endif
WT_D void light_stage_barrier() {
#if WTGPU_LIGHT_FENCE
    __threadfence();
    __syncthreads();
#else
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
__global__ void __launch_bounds__(kBlock, 2) k_light_rounds(launch_args_t a, int in, uint32_t round, uint32_t max_rounds) {
    uint32_t* ctl = a.st.ctl;
    uint32_t done = 0, stop = 0;
    // (one block: a queue that is not nearly empty — a batch that outlives its expected rounds wholesale — is the host's to continue with full grids)
    if (queue_count(ctl, in) > kLightMaxWalks) {
        if (threadIdx.x == 0) {
            ctl[CTL_LIGHT_DONE] = 0;
            ctl[CTL_LIGHT_STOP] = 4;
        }
        return;
    }
    if (queue_count(ctl, in) == 0) max_rounds = 0;
    for (; done < max_rounds; ++done) {
        trace_refill_body(a, in, 0, round + done);
        light_stage_barrier();
        if (ctl[CTL_HEAVY_COUNT] != 0) {
            stop = 1;
            break;
        }
        interact_body<0>(a, in, 0);
        light_stage_barrier();
        if (ctl[CTL_GATHER_COUNT] != 0) {
            stop = 2;
            break;
        }
        interact_body<1>(a, in, 0);
        light_stage_barrier();
        if (ctl[CTL_INTC_COUNT] != 0) {
            stop = 3;
            break;
        }
        in = 1 - in;
        if (queue_count(ctl, in) == 0) {
            ++done;
            break;
        }
    }
    if (threadIdx.x == 0) {
        ctl[CTL_LIGHT_DONE] = done;
        ctl[CTL_LIGHT_STOP] = stop;
    }
}

}   // namespace wtk
