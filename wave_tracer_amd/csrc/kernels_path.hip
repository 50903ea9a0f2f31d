// wave_tracer_amd — plt_path (see wtgpu_kernels.h for the list of kernel translation units).
#include "wtgpu_kernels.h"

namespace wtk {

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
    path_walk_t pw;
    path_generate(a.sc, a.seed, sample_id, pix % a.sc.sensor.width, pix / a.sc.sensor.width, pw);
    soa_store(a.st.walks, a.st.walk_words, i, pw);
}

// do_fsd (plt_path_detail.hpp:311-346) by ONE WAVEFRONT: lane = wedge (strided over apertures of any size) — the Fermat point on the wedge, the
// UTD coefficients and the two shadow rays (per-lane any-hit traversals on the lane's LDS stack) — coherent sums in f64 by wave reduction; the
// direct path is evaluated redundantly by all lanes (uniform control flow).  Returns (|ts|^2 + |th|^2) / 2.
#ifdef WTGPU_FSD_WATCH
extern "C" int wtgpu_debug_set_watch(unsigned int* host_mapped) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_watch), &host_mapped, sizeof(host_mapped));
}
#endif
// G: lanes per aperture (a power of two <= 64; the 64 / G apertures of a wavefront are independent: shuffles stay inside an aligned group of G lanes).
// The apertures of the city-block workload hold 11 wedges on average: one aperture per wavefront leaves 5 of 6 lanes without a wedge.
#ifndef WTGPU_FSD_GROUP
#define WTGPU_FSD_GROUP 8   // (etoile 720 x 540: 64 / 32 / 16 / 8 lanes per aperture -> 61.8 / 73.3 / 80.6 / 84.6 Msamples/s, profiles/r06_ab_experiments.log)
#endif
template <int G>
WT_D float coop_do_fsd(const scene_t& sc, const cone_t& cone_from_src, const path_geo_t& src_geo, vec3 dst, const utd_aperture_t& ap, const utd_edge_rec_t* recs,
                                    float k, const stack_ref_t& stack, bdpt_counters_t* ctr, bool have = true) {   // have = false: the group holds no aperture (it only takes part in the shuffles)
    const int lane = threadIdx.x & (G - 1);
    const vec3 src = cone_from_src.o;
    const path_geo_t dst_geo = path_geo_point(dst);
    double tsr = 0, tsi = 0, thr = 0, thi = 0;
    for (uint32_t i = (uint32_t)lane; have && i < ap.n_edges; i += (uint32_t)G) {
        utd_diffracting_edge_t f;
        WT_WATCH_ADD(12);
        if (lane == 0) WT_WATCH(13, i);
        if (lane == 0) WT_WATCH(14, ap.n_edges);
        if (lane == 0) WT_WATCH(5, i);
        if (!utd_f_edge(sc, ap, recs[i], src, dst, f)) continue;
        if (lane == 0) WT_WATCH(6, i);
        const path_geo_t eintr = path_geo_edge(f.edge, f.p);
        WT_WATCH_ADD(15);
        if (path_shadow(sc, eintr, src_geo, stack, ctr) || path_shadow(sc, eintr, dst_geo, stack, ctr)) continue;
        const cplx phase = cpolar(1.f, -k_times_len(k, f.ro + f.ri));
        const cplx a = phase * f.utd.Ds, b = phase * f.utd.Dh;
        tsr += a.re;
        tsi += a.im;
        thr += b.re;
        thi += b.im;
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        tsr += __shfl_xor(tsr, off, 64);
        tsi += __shfl_xor(tsi, off, 64);
        thr += __shfl_xor(thr, off, 64);
        thi += __shfl_xor(thi, off, 64);
    }
    cplx ts{(float)tsr, (float)tsi}, th{(float)thr, (float)thi};
    if (have && cone_contains(cone_from_src, dst)) {
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
    uint32_t* ctl = a.st.ctl;
    const uint32_t qin = round & 1u;
    const uint32_t n = ctl[CTL_FSDQ_COUNT0 + qin];
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack, 64u);
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    const utd_edge_rec_t* prev_pool = P.utd[(round + 1u) & 1u];
    constexpr int G = WTGPU_FSD_GROUP, NG = 64 / G;
    const uint32_t group = (threadIdx.x & 63u) / (uint32_t)G;
    for (;;) {
        const uint32_t item = wave_grab0(ctl + CTL_FSDQ_HEAD, (uint32_t)NG) + group;   // NG walks per wavefront, G lanes each
        if (item - group >= n) break;
        uint32_t w = 0;
        bool have = item < n;
        if (have) {
            w = P.fsdq[qin][item];
            have = a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(empty)] == 0;   // (an empty record: the step ends before do_fsd, plt_path_detail.hpp:577-581)
        }
        // (the lanes of a group read the same words: one record per group; fields the sum does not use are never loaded)
        path_walk_t pw;
        memset(&pw, 0, sizeof(pw));
        vec3 interaction_wp{0.f, 0.f, 0.f};
        if (have) {
            soa_load(a.st.walks, a.st.walk_words, w, pw);
            const vec3 origin{__uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(origin.x)]), __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(origin.y)]),
                              __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(origin.z)])};
            const float dist = __uint_as_float(a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(dist)]);
            interaction_wp = origin + dist * pw.w.beam.env.d;
        }
        const float f = coop_do_fsd<G>(a.sc, pw.prev_beam.env, path_geo_prev(pw.w), interaction_wp, pw.ap, prev_pool + pw.ap.edge_offset, pw.w.beam.k, stack, &ctr, have);
        if (have && (threadIdx.x & (uint32_t)(G - 1)) == 0) P.fsd_f[w] = f;
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
    coop_set_dropped_counter(csh, a.st.counters + kDroppedSlot);
    coop_set_dropped_counter(sh, a.st.counters + kDroppedSlot);
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_GATHER_COUNT];
    for (;;) {
        const uint32_t item = wave_grab_item(ctl + CTL_GATHER_HEAD);
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
            off = wave_grab0(ctl + CTL_EPOOL_COUNT, n_edges);
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
WT_D void path_interact_body(const launch_args_t& a, const path_state_t& P, int in, int first_round, uint32_t round) {
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
#ifndef WTGPU_LB_PATH
#define WTGPU_LB_PATH 2
#endif
__global__ void __launch_bounds__(kBlock, WTGPU_LB_PATH) k_path_interact(launch_args_t a, const path_state_t* __restrict__ ps, int in, int first_round, uint32_t round) { path_interact_body<0>(a, *ps, in, first_round, round); }
__global__ void __launch_bounds__(kBlock, WTGPU_LB_PATH) k_path_interact_b(launch_args_t a, const path_state_t* __restrict__ ps, int in, uint32_t round) { path_interact_body<1>(a, *ps, in, 0, round); }

// plt_path, after the interaction step: next-event estimation towards the virtual sensor through the aperture the step just built (nee_forward,
// plt_path_detail.hpp:474-518) — one wavefront per walk: coherent UTD sum (coop_do_fsd), beam transform, integrate_beams, light-image splat.
__global__ void __launch_bounds__(64, 3) k_path_nee(launch_args_t a, const path_state_t* __restrict__ ps, uint32_t round) {
    const path_state_t& P = *ps;
    __shared__ stack_entry_t lds[kLdsStack * 64];
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_NEEQ_COUNT];
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack, 64u);
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    const utd_edge_rec_t* cur_pool = P.utd[round & 1u];
    constexpr int G = WTGPU_FSD_GROUP, NG = 64 / G;
    const uint32_t group = (threadIdx.x & 63u) / (uint32_t)G;
    for (;;) {
        const uint32_t item = wave_grab0(ctl + CTL_NEEQ_HEAD, (uint32_t)NG) + group;   // NG walks per wavefront, G lanes each
        if (item - group >= n) break;
        const bool have = item < n;
        const uint32_t w = have ? P.neeq[item] : 0u;
        // what the coherent sum reads of the walk's record (the whole record — two beams — only in the lane that splats)
        cone_t env;
        memset(&env, 0, sizeof(env));
        env.d = vec3{0.f, 0.f, 1.f};
        path_geo_t src_geo{vec3{0.f, 0.f, 0.f}, 0u, vec3{0.f, 0.f, 1.f}, kInvalid};
        vec3 dst{0.f, 0.f, 0.f};
        float k = 0.f;
        utd_aperture_t ap;
        memset(&ap, 0, sizeof(ap));
        if (have) {
            const path_nee_rec_t& r = P.nee_recs[w];
            env = r.beam.env;
            src_geo = path_geo_t{r.src_wp, r.src_kind, r.src_ng, r.src_tuid};
            dst = r.sd_beam.env.o;
            k = r.beam.k;
            soa_load(a.st.walks + offsetof(path_walk_t, ap) / 4, a.st.walk_words, w, ap);   // the aperture k_path_interact just stored
        }
        const float f = coop_do_fsd<G>(a.sc, env, src_geo, dst, ap, cur_pool + ap.edge_offset, k, stack, &ctr, have);
        if (have && (threadIdx.x & (uint32_t)(G - 1)) == 0 && f != 0.f) {
            const path_nee_rec_t r = P.nee_recs[w];
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

}   // namespace wtk
