// wave_tracer_amd — the wave-per-walk parts of a Fraunhofer interaction: region power sums (k_flux_*), rejection sampling (k_interact_c*) (see wtgpu_kernels.h for the list of kernel translation units).
#include "wtgpu_kernels.h"

namespace wtk {

// Intercepted power of interaction regions that overflowed the bounded list (find_closest_triangle's sum over ALL region triangles,
// plt_bdpt_detail.hpp:391-416) for the pass-C walks.  Such regions hold 10^3..10^5 triangles (a wide emitter beam over a finely
// tessellated mesh), 5000 on average in the headline workload: one wavefront per region would leave the round waiting for the
// largest one (measured: 27 ms for a 130,000-triangle region).  k_flux_split cuts the part of the tree that overlaps the region
// into subtrees of <= kFluxTaskTris (2048; swept 128 / 512 / 2048: 247 / 216 / 208 ms per pass) triangles, k_flux_tasks sums every subtree on whichever wavefront is free (f64 atomics).
__global__ void __launch_bounds__(64, 3) k_flux_split(launch_args_t a) {
    __shared__ coop_gather_shared_t sh;
    coop_set_dropped_counter(sh, a.st.counters + kDroppedSlot);
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_INTC_COUNT];
    const int lane = threadIdx.x & 63;
    // 64 queue items per grab: every lane looks at one walk's marker (most pass-C walks have a region that fitted its list and need no
    // split — bidir_room: 400,000 items a round, a few thousand to split; one item per grab was 11.5 ms of a 125-ms batch there), the
    // wavefront then cuts the regions of the flagged ones, one after the other
    for (;;) {
        const uint32_t base = wave_grab(ctl + CTL_FSPLIT_HEAD);
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
    coop_set_dropped_counter(sh, a.st.counters + kDroppedSlot);
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = min(ctl[CTL_FTASK_COUNT], a.st.ftask_cap);
    const size_t W2 = 2 * (size_t)a.st.cap;
    for (;;) {
        const uint32_t item = wave_grab_item(ctl + CTL_FTASK_HEAD);
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
template <int BLOCK>
WT_D void interact_c_body(const launch_args_t& a, int in) {
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
        uint32_t item = 0;
        if (HARD) {   // four wavefronts: through the shared word, between real barriers — the first one BEFORE the branch on the thread index
            __syncthreads();   // (wtgpu_kernels.h: wave_grab0 says why; it also keeps the last iteration's readers ahead of the write)
            if (tid == 0) s_item = atomicAdd(ctl + CTL_INTD_HEAD, 1u);
            __syncthreads();
            item = s_item;
        } else
            item = wave_grab_item(ctl + CTL_INTC_HEAD);
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

}   // namespace wtk
