// wave_tracer_amd — the body of the per-lane trace kernel with lane refill (k_trace_refill, kernels_trace.hip), in a header because the fused
// light-round kernel of kernels_walk.hip (k_light_rounds) runs it too.
#pragma once
#include "wtgpu_kernels.h"

namespace wtk {

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
WT_D bool policy_next_query(const scene_t& sc, const cone_t& env, bool rt, const stack_ref_t& stack, axis_walk_t& aw, cone_query_t& q, trav_result_t& r) {
    for (;;) {
        const int need = aw_next(sc, env, rt, stack, aw, q, r);
        if (need != AW_TEST) return need == AW_QUERY;
        aw_test_done(aw, cone_attempt_too_short_by(sc, env, aw.cand, aw.sr, aw.min_df_prog));
    }
}
// what the first thread of a round's trace kernel resets: the queues and pools this round's later kernels fill
WT_D void trace_round_begin(uint32_t* ctl, int in, uint32_t round, uint32_t n) {
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
#pragma unroll
    for (uint32_t c = 0; c < kNumWalkClasses; ++c) ctl[CTL_CLS_COUNT0 + c] = ctl[CTL_CLS_HEAD0 + c] = 0;   // the class queues of this round's pass A
    // plt_path: this round's wedge pool and the queue it fills for the next round's k_path_fsd; this round's k_path_fsd / k_path_nee heads
    ctl[CTL_UTD_COUNT0 + (round & 1u)] = 0;
    ctl[CTL_FSDQ_COUNT0 + ((round + 1u) & 1u)] = 0;
    ctl[CTL_FSDQ_HEAD] = 0;
    ctl[CTL_NEEQ_COUNT] = 0;
    ctl[CTL_NEEQ_HEAD] = 0;
    if (n > 0) ctl[CTL_ROUNDS] = round + 1;
}
WT_D void trace_refill_body(const launch_args_t& a, int in, int first_round, uint32_t round) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = queue_count(ctl, in);
    if (blockIdx.x == 0 && threadIdx.x == 0) trace_round_begin(ctl, in, round, n);
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
                tris = uint_list_t{slot, 1u, (a.collect_list & 1u) ? kMaxConeTris : 0u, reinterpret_cast<float*>(slot + kMaxConeTris)};
                env = walk_trace_envelope(a.sc, wk);
                ray_hit_t ah;
                // (The axis query in a kernel of its own was built twice: round 3 as a grid-stride kernel — 60 vs 56 ms per pass — and round 4 as a
                // lane-refill kernel like this one (k_trace_axis: 111 registers, 4 waves per SIMD, 2.2 G rays/s in the long rounds: 3.9 ms where this
                // section spends ~3): the two kernels together took 24.2 ms of the long rounds against 23.4 ms with the query in here, 22.4 vs 22.5
                // Msamples/s — the fetch section's rays overlap other wavefronts' cone queries, which a separate kernel gives up.  Not kept.)
                const bool axis_hit = ads_intersect_ray(a.sc, env.o, env.d, range_t{0.f, WT_INF}, stack, ah);
                aw_begin(aw, wavenum_to_wavelen_m(wk.k), WT_INF, axis_hit, ah, a.cone_budget, true, !(a.collect_list & 1u) || (a.collect_list & 2u), a.lane_cache ? wk.prev_offset_tuid : kInvalid);
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
                    if (a.collect_list & 1u) ctr.cone_tri_overflow += r.overflow;
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


}   // namespace wtk
