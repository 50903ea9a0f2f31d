// wave_tracer_amd — the traversal kernels: per-lane with lane refill (k_trace_refill), wave-cooperative (k_trace_heavy); per-query test kernels (see wtgpu_kernels.h for the list of kernel translation units).
#include "wtgpu_kernels.h"

namespace wtk {

}   // namespace wtk
#include "kernels_trace_refill.h"
namespace wtk {

__global__ void __launch_bounds__(kBlock, WTGPU_LB_TRACE) k_trace_refill(launch_args_t a, int in, int first_round, uint32_t round) { trace_refill_body(a, in, first_round, round); }

// ---- The per-lane trace kernel as a PHASE MACHINE (round 6; NOT the default: 1.7x slower than the staged kernels, kept as a parity-tested form —
// WTGPU_TRACE_SM=1 selects it; DESIGN.md §4 "The other forms").
// k_trace_refill above keeps a wavefront's lanes busy with different WALKS, but inside a step they still do different THINGS: the axis query of a
// freshly fetched walk is a whole ray traversal (a loop nest whose trip counts differ in every lane), a leaf step loops over one to four triangles
// per lane and, for the 3 % of them that pass the conservative filters, runs the exact cone-triangle intersection (intersect_cone_tri: ~10x a
// filter test) with one or two lanes while the others wait; the policy's tests of the remembered triangles — one or two exact tests per diffusive
// attempt, 9 per sample on the headline workload — run in the service section with whatever lanes happen to be there.  Measured
// (profiles/r05_pmc_SQ_lane_utilisation.csv): 9.6 of 64 lanes per vector instruction with the vector pipes 44 % busy — the kernel spends its issue
// slots on nearly empty instructions.  Here every lane is in exactly one PHASE and a step runs ONE phase for all lanes in it:
//   NODE    pops a stack entry: a leaf whose triangles it takes, or a node: the ray's eight slab tests or the cone's eight box tests and the sorted
//           push;
//   TRI     fetches ONE triangle — a triangle of the held leaf or a remembered triangle of the policy — and runs the ray-triangle test (axis
//           query) or the conservative filters (cone_tri_maybe) against the query's current slab / the attempt's search range; what passes the
//           filters is staged in the lane's LDS slot;
//   EXACT   the exact cone-triangle test of the staged triangle (leaf triangles and policy tests alike), taken only when WTGPU_SM_EXACT_MIN lanes
//           wait for it or nothing else can move: the expensive code runs with many lanes or not at all;
//   RAYDONE / SERVE / POLICY (the service section)  the axis query ended (aw_begin) / a cone query ended (cq_end, aw_query_done) / a policy test
//           was answered (aw_test_done), then aw_next for all of them together — one inlined copy of the policy instead of two.
// Per lane the sequence of node visits, triangle tests, list entries and budget charges is EXACTLY that of wt::traverse_axis (the filter is a pure
// pre-test of intersect_cone_tri's own first two rejections), so the records are bit-identical to k_trace_refill's
// (tests/test_gpu_traversal.py::test_trace_kernels_write_identical_records replays rounds through both: WTGPU_TRACE_AB).
#ifndef WTGPU_SM_EXACT_MIN
#define WTGPU_SM_EXACT_MIN 24
#endif
#ifndef WTGPU_SM_SPLIT_NODE
#define WTGPU_SM_SPLIT_NODE 0
#endif
#ifndef WTGPU_SM_BOTH
#define WTGPU_SM_BOTH 0   // 1: a step runs the NODE and the TRI phase (when each has lanes) instead of the larger of the two
#endif
enum : int { PH_IDLE = 0, PH_NODE = 1, PH_TRI = 2, PH_EXACT = 3, PH_SERVE = 4, PH_POLICY = 5, PH_RAYDONE = 6 };
#ifdef WTGPU_SM_PROF
// (diagnostic build, tools/build_variant.sh smprof -DWTGPU_SM_PROF: wave clocks, steps and lanes per phase into the profile slots 32.. of the counters)
#define SP_DECL() unsigned long long sp_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sp_l[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sp_n[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long sp_c = 0
#define SP_BEGIN() sp_c = clock64()
#define SP_END(i, mask) do { const long long d_ = clock64() - sp_c; sp_t[i] += (unsigned long long)d_; sp_l[i] += (unsigned long long)d_ * (unsigned long long)__popcll(mask); sp_n[i] += 1; } while (0)
#else
#define SP_DECL()
#define SP_BEGIN()
#define SP_END(i, mask)
#endif
__global__ void __launch_bounds__(kBlock, WTGPU_LB_TRACE) k_trace_sm(launch_args_t a, int in, int first_round, uint32_t round) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    __shared__ float lds_tri[12 * kBlock];   // the triangle a lane holds for its exact test: word k of lane l at [k * kBlock + l]
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = queue_count(ctl, in);
    if (blockIdx.x == 0 && threadIdx.x == 0) trace_round_begin(ctl, in, round, n);
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack);
    const lane_nodes_t ns = lane_nodes(a.sc);
    const bool rt = a.sc.sensor.ray_trace_only || a.sc.opts.force_ray_tracing;
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    float* my_tri = lds_tri + threadIdx.x;
    int ph = PH_IDLE;
    uint32_t w = 0;
    uint32_t t_next = 0, t_left = 0;   // TRI / EXACT: the triangle to test, how many of the leaf are left (including it)
    bool ray = false;                  // NODE / TRI: the axis query (its state: q.s, t_next / t_left, the hit so far in aw.ah)
    bool pol = false, pol_hit = false;   // TRI / EXACT: a remembered triangle of the policy (not a leaf triangle); POLICY: its answer
    cone_t env;
    axis_walk_t aw;
    cone_query_t q;
    memset(&env, 0, sizeof(env));
    memset(&aw, 0, sizeof(aw));
    memset(&q, 0, sizeof(q));
    bool exhausted = false;   // wave-uniform: the queue holds no more walks
    SP_DECL();
    for (;;) {
        // ======== service section
        bool fin = false, need_next = false;
        SP_BEGIN();
        const unsigned long long sp_m0 = __ballot(ph == PH_SERVE || ph == PH_POLICY || ph == PH_RAYDONE);
        trav_result_t r;
        // plt_bdpt: the bounded list (64 triangles + their cone-hit distances) of the interaction region in the walk's slot
        uint32_t* slot = a.st.tris + (size_t)w * kTriListWords;
        const uint_list_t tris{slot, 1u, (a.collect_list & 1u) ? kMaxConeTris : 0u, reinterpret_cast<float*>(slot + kMaxConeTris)};
        if (ph == PH_SERVE) {
            cq_end(env, tris, q);
            fin = aw_query_done(a.sc, env, aw, q.rec, r);
            need_next = !fin;
        } else if (ph == PH_POLICY) {
            aw_test_done(aw, pol_hit);
            need_next = true;
        } else if (ph == PH_RAYDONE) {
            // (ads_intersect_ray's epilogue: a hit beyond the range is none — the range is unbounded here)
            const ray_hit_t ah = aw.ah;
            const float lambda_m = aw.lambda_m;
            const uint32_t origin_tuid = aw.origin_tuid;
            aw_begin(aw, lambda_m, WT_INF, finitef(ah.dist), ah, a.cone_budget, true, !(a.collect_list & 1u) || (a.collect_list & 2u), origin_tuid);
            aw.use_cache = a.lane_cache;
            need_next = true;
        }
        SP_END(0, sp_m0);
        {
            const unsigned long long im = __ballot(ph == PH_IDLE);
            const bool any_work = __ballot(ph == PH_NODE || ph == PH_TRI || ph == PH_EXACT || need_next) != 0;
            if (!exhausted && im && (__popcll(im) >= WTGPU_REFILL_MIN || !any_work)) {
                const bool take = ph == PH_IDLE;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(ctl + CTL_HEAD_TRACE, (uint32_t)__popcll(im));
                base = (uint32_t)__shfl((int)base, 0, 64);
                if (base + (uint32_t)__popcll(im) >= n) exhausted = true;
                const uint32_t qi = base + (uint32_t)__popcll(im & below);
                SP_BEGIN();
                const unsigned long long sp_m1 = __ballot(take && qi < n);
                if (take && qi < n) {
                    w = queue_walk(a, ctl, in, qi, first_round);
                    const walk_trace_in_t wk = walk_load_trace_in(a.st.walks, a.st.walk_words, w);
                    env = walk_trace_envelope(a.sc, wk);
                    // the axis query (ads_intersect_ray over [0, inf)) begins: rq_begin
                    aw.lambda_m = wavenum_to_wavelen_m(wk.k);
                    aw.origin_tuid = a.lane_cache ? wk.prev_offset_tuid : kInvalid;
                    aw.ah.dist = WT_INF;
                    aw.ah.tuid = kInvalid;
                    aw.ah.bx = aw.ah.by = 0.f;
                    aw.ah.front_face = 0;
                    ray = true;
                    pol = false;
                    q.s = 0;
                    ph = PH_RAYDONE;
                    if (a.sc.n_nodes != 0) {
                        stack.set(0, stack_entry_t{0.f, 1});
                        q.s = 1;
                        ph = PH_NODE;
                    }
                }
                SP_END(1, sp_m1);
            }
        }
        SP_BEGIN();
        const unsigned long long sp_m2 = __ballot(need_next);
        if (need_next) {
            const int need = aw_next(a.sc, env, rt, stack, aw, q, r);
            ray = false;
            pol = need == AW_TEST;
            if (need == AW_FINAL)
                fin = true;
            else if (need == AW_TEST) {
                t_next = aw.cand;
                t_left = 1;
                ph = PH_TRI;
            } else
                ph = cq_running(q) ? PH_NODE : PH_SERVE;
        }
        SP_END(2, sp_m2);
        SP_BEGIN();
        const unsigned long long sp_m3 = __ballot(fin);
        {   // store the records of the walks that ended in this section (convergent: the queue append is a wave operation)
            const bool heavy = fin && r.aborted == 1;
            if (fin) {
                ph = PH_IDLE;
                if (heavy) {
                    // resume state for k_trace_heavy (aw_query_done: dist / ntris = distance / segment of the query, the axis hit, the last
                    // rejecting triangle in `overflow`)
                    a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(dist)] = __float_as_uint(r.dist);
                    a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(ntris)] = r.ntris;
                    a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(n_ray_queries)] = r.n_ray_queries;
                    a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(n_cone_queries)] = r.n_cone_queries;
                    a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(tuid)] = r.tuid;
                    a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(bx)] = __float_as_uint(r.bx);
                    a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(by)] = __float_as_uint(r.by);
                    a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(pdist)] = __float_as_uint(r.pdist);
                    a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(front_face)] = r.front_face;
                    a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(overflow)] = r.overflow;
                } else {
                    soa_store(a.st.trav, kTravWords, w, r);
                    ctr.segments += 1;
                    ctr.ray_queries += r.n_ray_queries;
                    ctr.cone_queries += r.n_cone_queries;
                    if (a.collect_list & 1u) ctr.cone_tri_overflow += r.overflow;
                }
            }
            wave_append(a.st.heavy_queue, ctl + CTL_HEAVY_COUNT, heavy, w);
        }
        SP_END(3, sp_m3);
        const unsigned long long m_work = __ballot(ph == PH_NODE || ph == PH_TRI || ph == PH_EXACT);
        if (!m_work) {
            if (__ballot(ph == PH_SERVE || ph == PH_POLICY || ph == PH_RAYDONE)) continue;   // (queries that ended at once, e.g. an empty scene)
            if (exhausted) break;
            continue;   // every lane is idle: fetch
        }
        // ======== traversal loop: until a quarter of the lanes that entered it (at most WTGPU_REFILL_MIN) wait to be served
        const int running = __popcll(m_work);
#ifdef WTGPU_SM_LEAVE
        const int leave_at = WTGPU_SM_LEAVE;   // (A/B: the service section only when this many lanes wait for it, or nothing else can move)
#else
        const int leave_at = running < 4 * WTGPU_REFILL_MIN ? (running + 3) / 4 : WTGPU_REFILL_MIN;
#endif
        for (;;) {
            const int c_node = __popcll(__ballot(ph == PH_NODE)), c_tri = __popcll(__ballot(ph == PH_TRI)), c_exact = __popcll(__ballot(ph == PH_EXACT));
            if (c_exact >= WTGPU_SM_EXACT_MIN || (c_exact > 0 && c_node + c_tri == 0)) {
                SP_BEGIN();
                const unsigned long long sp_m6 = __ballot(ph == PH_EXACT);
                if (ph == PH_EXACT) {
                    tri_geo_t tri;
                    tri.a = vec3{my_tri[0 * kBlock], my_tri[1 * kBlock], my_tri[2 * kBlock]};
                    tri.b = vec3{my_tri[3 * kBlock], my_tri[4 * kBlock], my_tri[5 * kBlock]};
                    tri.c = vec3{my_tri[6 * kBlock], my_tri[7 * kBlock], my_tri[8 * kBlock]};
                    tri.n = vec3{my_tri[9 * kBlock], my_tri[10 * kBlock], my_tri[11 * kBlock]};
                    const range_t rg = pol ? aw.sr : q.range;
                    cone_tri_hit_t h;
                    const bool hit = intersect_cone_tri(env, tri.a, tri.b, tri.c, tri.n, rg, h) && !(h.dist > rg.max);   // (numerics: bvh8w.cpp:162)
                    if (pol) {
                        pol_hit = hit && h.dist - rg.min < aw.min_df_prog;   // cone_attempt_too_short_by
                        ph = PH_POLICY;
                    } else {
                        if (hit) cq_apply_hit(env, stack, tris, q, t_next, h.dist, dot(tri.n, -env.d) > 0.f);
                        ++t_next;
                        --t_left;
                        if (q.rec.too_short)   // (cq_stop: nothing is left to visit)
                            ph = PH_SERVE;
                        else
                            ph = t_left ? PH_TRI : (q.s > 0 ? PH_NODE : PH_SERVE);
                    }
                }
                SP_END(6, sp_m6);
            } else {
                const bool do_tri = WTGPU_SM_BOTH ? c_tri > 0 : c_tri >= c_node;
                bool do_node = WTGPU_SM_BOTH ? c_node > 0 : !do_tri;
#if WTGPU_SM_SPLIT_NODE
                if (do_node) {   // one kind of query per NODE step: the kind with more lanes
                    const int c_ray = __popcll(__ballot(ph == PH_NODE && ray));
                    do_node = (2 * c_ray >= c_node) == ray;
                }
#endif
                SP_BEGIN();
                const unsigned long long sp_m4 = __ballot(do_node && ph == PH_NODE);
                (void)sp_m4;
                if (do_node && ph == PH_NODE) {
                    const stack_entry_t top = stack.get(q.s - 1);
                    --q.s;
                    if (top.ptr < 0) {   // a leaf: its triangles, one per TRI step
                        bvh8_leaf_t leaf = bvh_leaf_of(top.ptr);
                        if (!ray) {
                            q.leaf = top.ptr;
                            leaf = cq_take_leaf(q);   // (the budget charge; count 0: exceeded, the query stopped)
                        }
                        t_next = leaf.tris_ptr;
                        t_left = leaf.count;
                        ph = leaf.count ? PH_TRI : PH_SERVE;
                    } else {
                        if (ray) {   // (the exact nodes: wt/bvh.h, rq_node_step)
                            const wide_nodes_t ws{a.sc.nodes};
                            const bvh8_node_t nd = ws.fetch(top.ptr - 1);
                            ray_query_t rq;
                            rq.range = range_t{0.f, WT_INF};
                            rq.rec = aw.ah;
                            rq.s = q.s;
                            rq.lt0 = rq.lcnt = 0;
                            rq_node_children(ws, nd, env.o, env.d, stack, rq);
                            q.s = rq.s;
                            if (q.s == 0) ph = PH_RAYDONE;
                        } else {
                            const lane_nodes_t::node_t nd = ns.fetch(top.ptr - 1);
                            cq_node_children(ns, nd, env, stack, q);
                            if (q.s == 0) ph = PH_SERVE;   // ended (nothing left) or stopped (budget, full stack)
                        }
                    }
                }
                if (sp_m4) SP_END(4, sp_m4);
                SP_BEGIN();
                const unsigned long long sp_m5 = __ballot(do_tri && ph == PH_TRI);
                (void)sp_m5;
                if (do_tri && ph == PH_TRI) {
                    const tri_geo_t tri = a.sc.tri_geo[t_next];
                    if (ray) {
                        ray_tri_hit_t h;
                        if (intersect_ray_tri_wide(env.o, env.d, tri.a, tri.b, tri.c, range_t{0.f, WT_INF}, h) && h.dist < aw.ah.dist) {   // ray_gather_tris
                            aw.ah.dist = h.dist;
                            aw.ah.bx = h.bx;
                            aw.ah.by = h.by;
                            aw.ah.tuid = t_next;
                            aw.ah.front_face = dot(tri.n, env.d) <= 0.f;
                            int s = q.s;   // rq_leaf_step: nothing farther than the hit is left to visit
                            while (s > 0 && stack.get_t(s - 1) >= h.dist) --s;
                            q.s = s;
                        }
                        ++t_next;
                        --t_left;
                        ph = t_left ? PH_TRI : (q.s > 0 ? PH_NODE : PH_RAYDONE);
                    } else {
                        const range_t rg = pol ? aw.sr : q.range;
                        if (cone_tri_maybe(env, tri.a, tri.b, tri.c, rg)) {
                            my_tri[0 * kBlock] = tri.a.x; my_tri[1 * kBlock] = tri.a.y; my_tri[2 * kBlock] = tri.a.z;
                            my_tri[3 * kBlock] = tri.b.x; my_tri[4 * kBlock] = tri.b.y; my_tri[5 * kBlock] = tri.b.z;
                            my_tri[6 * kBlock] = tri.c.x; my_tri[7 * kBlock] = tri.c.y; my_tri[8 * kBlock] = tri.c.z;
                            my_tri[9 * kBlock] = tri.n.x; my_tri[10 * kBlock] = tri.n.y; my_tri[11 * kBlock] = tri.n.z;
                            ph = PH_EXACT;
                        } else if (pol) {
                            pol_hit = false;
                            ph = PH_POLICY;
                        } else {
                            ++t_next;
                            --t_left;
                            ph = t_left ? PH_TRI : (q.s > 0 ? PH_NODE : PH_SERVE);
                        }
                    }
                }
                if (sp_m5) SP_END(5, sp_m5);
            }
            const int waiting = __popcll(__ballot(ph == PH_SERVE || ph == PH_POLICY || ph == PH_RAYDONE)) + (exhausted ? 0 : __popcll(__ballot(ph == PH_IDLE)));
            if (waiting >= leave_at || !__ballot(ph == PH_NODE || ph == PH_TRI || ph == PH_EXACT)) break;
        }
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
#ifdef WTGPU_SM_PROF
    if (lane == 0)
        for (int i = 0; i < 8; ++i) {
            atomicAdd(a.st.counters + kNumCounters + 32 + i, sp_t[i]);
            atomicAdd(a.st.counters + kNumCounters + 40 + i, sp_l[i]);
            atomicAdd(a.st.counters + kNumCounters + 48 + i, sp_n[i]);
        }
#endif
}

// ---- The per-lane traversal in STAGES (round 6; WTGPU_TRACE_STAGED=1).
// k_trace_refill is one kernel that holds, per lane, everything a walk's traversal ever needs — the envelope, the policy's state, a cone query, the
// axis ray query's traversal, three inlined copies of the exact cone-triangle test — and the compiler gives it 168 registers and spills 56-107 more;
// what a lane does at any moment differs from its neighbours' (9.6 of 64 lanes per vector instruction).  Measured this round: neither half the node
// bytes, nor a cheaper box test, nor 2 or 4 waves per SIMD instead of 3 move it (profiles/r06_ab_experiments.log) — what it does per lane is
// cheap, how it is organised is not.  Here the traversal of a round's walks is cut where the policy (wt::traverse_axis's aw_* steps) already
// hands work out, into kernels that each do ONE thing with every lane and keep a walk's state in memory in between (trace_stage_t, 152 B):
//   k_tr_axis      lane = walk of the round's queue: the beam axis's closest hit (a plain ray traversal), aw_begin, then the policy up to its first
//                  cone query — the tests of the remembered triangles (one or two exact cone-triangle tests per attempt) run here with every
//                  lane of the wavefront doing the same;
//   k_tr_cone      lane = cone query, with lane refill from the stage's queue: node steps / leaf steps only (cq_*: envelope + query state, nothing
//                  else in registers), then aw_query_done: final record, hand-over to the wave-cooperative kernel, or back to the policy;
//   k_tr_policy    lane = walk whose attempt was rejected: the policy up to the next cone query (remembered-triangle tests);
//   k_tr_tail      after kTraceStages cone stages: the few walks still between attempts run policy and queries to their end in one lane each.
// Per walk the sequence of queries, visits and tests is that of wt::traverse_axis, as in k_trace_refill: the records are the same, word for word
// (tests/test_gpu_traversal.py::test_trace_kernels_write_identical_records).
WT_D void tr_finish(const launch_args_t& a, uint32_t* ctl, bdpt_counters_t& ctr, bool fin, uint32_t w, const trav_result_t& r) {   // (convergent: a wave operation)
    const bool heavy = fin && r.aborted == 1;
    if (fin) {
        if (heavy) {
            // resume state for k_trace_heavy (aw_query_done: dist / ntris = distance / segment of the query, the axis hit, the last
            // rejecting triangle in `overflow`)
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(dist)] = __float_as_uint(r.dist);
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(ntris)] = r.ntris;
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(n_ray_queries)] = r.n_ray_queries;
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(n_cone_queries)] = r.n_cone_queries;
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(tuid)] = r.tuid;
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(bx)] = __float_as_uint(r.bx);
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(by)] = __float_as_uint(r.by);
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(pdist)] = __float_as_uint(r.pdist);
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(front_face)] = r.front_face;
            a.st.trav[(size_t)w * kTravWords + WT_TRAV_WORD(overflow)] = r.overflow;
        } else {
            soa_store(a.st.trav, kTravWords, w, r);
            ctr.segments += 1;
            ctr.ray_queries += r.n_ray_queries;
            ctr.cone_queries += r.n_cone_queries;
            if (a.collect_list & 1u) ctr.cone_tri_overflow += r.overflow;
        }
    }
    wave_append(a.st.heavy_queue, ctl + CTL_HEAVY_COUNT, heavy, w);
}
// the policy from its current state up to the next cone query (TRUE: `aw` describes it) or its end (FALSE: `r` is final)
WT_D bool tr_policy_run(const scene_t& sc, const cone_t& env, bool rt, const stack_ref_t& stack, axis_walk_t& aw, trav_result_t& r) {
    cone_query_t q;   // (aw_next begins the query on `stack`; the cone stage begins it again on its own: cq_begin from aw.sr / aw.min_df_prog)
    for (;;) {
        const int need = aw_next(sc, env, rt, stack, aw, q, r);
        if (need != AW_TEST) return need == AW_QUERY;
        aw_test_done(aw, cone_attempt_too_short_by(sc, env, aw.cand, aw.sr, aw.min_df_prog));
    }
}
#ifndef WTGPU_LB_STAGE
#define WTGPU_LB_STAGE 3   // (4 waves per SIMD: 83-115 spilled registers, 5-8 % slower per round, profiles/r06_ab_experiments.log)
#endif
__global__ void __launch_bounds__(kBlock, WTGPU_LB_STAGE) k_tr_axis(launch_args_t a, int in, int first_round, uint32_t round) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = queue_count(ctl, in);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        trace_round_begin(ctl, in, round, n);
        ctl[CTL_TCONE_HEAD] = 0;      // this round's k_tr_cone(0) reads the queue this kernel fills (count 0) from its start
        ctl[CTL_TPOL_COUNT1] = 0;     // ... and fills this one
    }
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack);
    const bool rt = a.sc.sensor.ray_trace_only || a.sc.opts.force_ray_tracing;
    const int lane = threadIdx.x & 63;
    uint32_t* stage = trace_stage_words(a);
    uint32_t* coneq = trace_cone_queue(a);
    for (;;) {
        const uint32_t base = wave_grab(ctl + CTL_HEAD_TRACE);
        if (base >= n) break;
        const uint32_t qi = base + (uint32_t)lane;
        bool fin = false, query = false;
        uint32_t w = 0;
        trav_result_t r;
        if (qi < n) {
            w = queue_walk(a, ctl, in, qi, first_round);
            const walk_trace_in_t wk = walk_load_trace_in(a.st.walks, a.st.walk_words, w);
            trace_stage_t ts;
            ts.env = walk_trace_envelope(a.sc, wk);
            ray_hit_t ah;
            const bool axis_hit = ads_intersect_ray(a.sc, ts.env.o, ts.env.d, range_t{0.f, WT_INF}, stack, ah);
            aw_begin(ts.aw, wavenum_to_wavelen_m(wk.k), WT_INF, axis_hit, ah, a.cone_budget, true, !(a.collect_list & 1u) || (a.collect_list & 2u), a.lane_cache ? wk.prev_offset_tuid : kInvalid);
            ts.aw.use_cache = a.lane_cache;
            query = tr_policy_run(a.sc, ts.env, rt, stack, ts.aw, r);
            fin = !query;
            if (query) soa_store(stage, kStageWords, w, ts);
        }
        wave_append(coneq, ctl + CTL_TCONE_COUNT0, query, w);
        tr_finish(a, ctl, ctr, fin, w, r);
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}
// stage `it` of the cone queries: consumes the cone queue (count [it & 1]), fills the policy queue (count [(it + 1) & 1])
// WTGPU_CONE_DEFER = n > 0: the exact cone-triangle tests of the leaf steps are DEFERRED — a leaf step only fetches and filters (cone_tri_maybe);
// a triangle that passes waits in the lane's LDS slot until n lanes of the wavefront hold one (or nothing else can move), then all of them run
// intersect_cone_tri together.  Same tests in the same order per lane, so the same records.
#ifndef WTGPU_CONE_REFILL_MIN
#define WTGPU_CONE_REFILL_MIN 16   // idle lanes of a wavefront before it fetches queries again (k_trace_refill has its own: WTGPU_REFILL_MIN)
#endif
#ifndef WTGPU_CONE_DEFER
#define WTGPU_CONE_DEFER 0
#endif
__global__ void __launch_bounds__(kBlock, WTGPU_LB_STAGE) k_tr_cone(launch_args_t a, uint32_t it) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
#if WTGPU_CONE_DEFER
    __shared__ float lds_tri[12 * kBlock];   // the triangle a lane holds for its exact test: word k of lane l at [k * kBlock + l]
    float* my_tri = lds_tri + threadIdx.x;
    bool wait = false;
#endif
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_TCONE_COUNT0 + (it & 1u)];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl[CTL_TPOL_HEAD] = 0;                               // the next policy stage reads its queue from the start
        ctl[CTL_TCONE_COUNT0 + ((it + 1u) & 1u)] = 0;         // ... and fills this one
    }
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack);
    const lane_nodes_t ns = lane_nodes(a.sc);
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t* stage = trace_stage_words(a);
    const uint32_t* coneq = trace_cone_queue(a);
    uint32_t* polq = trace_pol_queue(a);
    uint32_t* pol_count = ctl + CTL_TPOL_COUNT0 + ((it + 1u) & 1u);
    // lane state: 0 = no query, 1 = cone query running, 2 = cone query ended (to be served)
    int st = 0;
    uint32_t w = 0;
    cone_t env;
    cone_query_t q;
    memset(&env, 0, sizeof(env));
    memset(&q, 0, sizeof(q));
    bool exhausted = false;   // wave-uniform: the queue holds no more queries
    SP_DECL();
    for (;;) {
        // ---- service section: ended queries go back to the policy (or are final), idle lanes fetch
        uint32_t* slot = a.st.tris + (size_t)w * kTriListWords;
        const uint_list_t tris{slot, 1u, (a.collect_list & 1u) ? kMaxConeTris : 0u, reinterpret_cast<float*>(slot + kMaxConeTris)};
        bool fin = false, again = false;
        trav_result_t r;
        SP_BEGIN();
        const unsigned long long sp_m0 = __ballot(st == 2);
        (void)sp_m0;
        if (st == 2) {
            cq_end(env, tris, q);
            axis_walk_t aw;
            soa_load(stage + offsetof(trace_stage_t, aw) / 4, kStageWords, w, aw);
            fin = aw_query_done(a.sc, env, aw, q.rec, r);
            again = !fin;
            if (again) {   // (aw_query_done changed the segment and the remembered rejecting triangle)
                stage[(size_t)w * kStageWords + (offsetof(trace_stage_t, aw) + offsetof(axis_walk_t, seg)) / 4] = aw.seg;
                stage[(size_t)w * kStageWords + (offsetof(trace_stage_t, aw) + offsetof(axis_walk_t, short_tuid)) / 4] = aw.short_tuid;
            }
            st = 0;
        }
        wave_append(polq, pol_count, again, w);
        tr_finish(a, ctl, ctr, fin, w, r);
        SP_END(0, sp_m0);
        {
            const unsigned long long im = __ballot(st == 0);
            const int n_run = __popcll(__ballot(st == 1));
            if (!exhausted && im && (__popcll(im) >= WTGPU_CONE_REFILL_MIN || n_run == 0)) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(ctl + CTL_TCONE_HEAD, (uint32_t)__popcll(im));
                base = (uint32_t)__shfl((int)base, 0, 64);
                if (base + (uint32_t)__popcll(im) >= n) exhausted = true;
                const uint32_t qi = base + (uint32_t)__popcll(im & below);
                SP_BEGIN();
                const unsigned long long sp_m1 = __ballot(st == 0 && qi < n);
                (void)sp_m1;
                if (st == 0 && qi < n) {
                    w = coneq[qi];
                    soa_load(stage, kStageWords, w, env);   // (the record begins with the envelope)
                    const size_t o = (size_t)w * kStageWords + offsetof(trace_stage_t, aw) / 4;
                    const range_t sr{__uint_as_float(stage[o + offsetof(axis_walk_t, sr) / 4]), __uint_as_float(stage[o + offsetof(axis_walk_t, sr) / 4 + 1])};
                    const float min_df_prog = __uint_as_float(stage[o + offsetof(axis_walk_t, min_df_prog) / 4]);
                    cq_begin(a.sc, env, sr, kMajorAxisToZScale, stack, a.cone_budget, min_df_prog, q);   // (aw.probe_first is set: wt::aw_next)
                    st = cq_running(q) ? 1 : 2;
                }
                SP_END(1, sp_m1);
            }
        }
        const int running = __popcll(__ballot(st == 1));
        if (running == 0) {
            if (__ballot(st == 2)) continue;
            if (exhausted) break;
            continue;
        }
        // ---- traversal loop: until a quarter of the lanes that entered it (at most WTGPU_CONE_REFILL_MIN) wait to be served
        uint32_t* slot2 = a.st.tris + (size_t)w * kTriListWords;
        const uint_list_t tris2{slot2, 1u, (a.collect_list & 1u) ? kMaxConeTris : 0u, reinterpret_cast<float*>(slot2 + kMaxConeTris)};
        const int leave_at = running < 4 * WTGPU_CONE_REFILL_MIN ? (running + 3) / 4 : WTGPU_CONE_REFILL_MIN;
#if WTGPU_CONE_DEFER
        for (;;) {
            for (;;) {   // nodes: every lane that holds no leaf descends, until the lanes with a leaf (or a staged triangle) are the majority
                const bool at_node = st == 1 && !wait && q.leaf == 0 && q.s > 0;
                if (!__ballot(at_node)) break;
                SP_BEGIN();
                if (at_node) {
                    cq_node_step(ns, env, stack, q);
                    if (q.leaf != 0) {   // a leaf: charged to the budget now, its triangles filtered below
                        const bvh8_leaf_t leaf = cq_take_leaf(q);
                        q.leaf = leaf.count ? -(int32_t)((leaf.tris_ptr << 3) | leaf.count) : 0;   // (count 0: over budget, the query stopped)
                    }
                }
                SP_END(4, __ballot(at_node));
                if (WTGPU_LEAF_DEN * __popcll(__ballot(st == 1 && (q.leaf != 0 || wait))) >= WTGPU_LEAF_NUM * running) break;
            }
            SP_BEGIN();
            const unsigned long long sp_m5 = __ballot(st == 1 && !wait && q.leaf != 0);
            (void)sp_m5;
            if (st == 1 && !wait && q.leaf != 0) {   // fetch + filter the leaf's remaining triangles up to the first that passes
                bvh8_leaf_t leaf = bvh_leaf_of(q.leaf);
                while (leaf.count) {
                    const tri_geo_t tri = a.sc.tri_geo[leaf.tris_ptr];
                    if (cone_tri_maybe(env, tri.a, tri.b, tri.c, q.range)) {
                        my_tri[0 * kBlock] = tri.a.x; my_tri[1 * kBlock] = tri.a.y; my_tri[2 * kBlock] = tri.a.z;
                        my_tri[3 * kBlock] = tri.b.x; my_tri[4 * kBlock] = tri.b.y; my_tri[5 * kBlock] = tri.b.z;
                        my_tri[6 * kBlock] = tri.c.x; my_tri[7 * kBlock] = tri.c.y; my_tri[8 * kBlock] = tri.c.z;
                        my_tri[9 * kBlock] = tri.n.x; my_tri[10 * kBlock] = tri.n.y; my_tri[11 * kBlock] = tri.n.z;
                        wait = true;
                        break;
                    }
                    ++leaf.tris_ptr;
                    --leaf.count;
                }
                q.leaf = leaf.count ? -(int32_t)((leaf.tris_ptr << 3) | leaf.count) : 0;
            }
            SP_END(5, sp_m5);
            {
                const int c_wait = __popcll(__ballot(wait));
                const bool others = __ballot(st == 1 && !wait && (q.leaf != 0 || q.s > 0)) != 0;
                if (c_wait >= WTGPU_CONE_DEFER || (c_wait > 0 && !others)) {
                    SP_BEGIN();
                    const unsigned long long sp_m6 = __ballot(wait);
                    (void)sp_m6;
                    if (wait) {
                        tri_geo_t tri;
                        tri.a = vec3{my_tri[0 * kBlock], my_tri[1 * kBlock], my_tri[2 * kBlock]};
                        tri.b = vec3{my_tri[3 * kBlock], my_tri[4 * kBlock], my_tri[5 * kBlock]};
                        tri.c = vec3{my_tri[6 * kBlock], my_tri[7 * kBlock], my_tri[8 * kBlock]};
                        tri.n = vec3{my_tri[9 * kBlock], my_tri[10 * kBlock], my_tri[11 * kBlock]};
                        bvh8_leaf_t leaf = bvh_leaf_of(q.leaf);
                        cq_exact_step_tri(env, stack, tris2, q, leaf.tris_ptr, tri);
                        wait = false;
                        if (q.rec.too_short)   // (cq_stop: nothing is left to visit)
                            q.leaf = 0;
                        else {
                            ++leaf.tris_ptr;
                            --leaf.count;
                            q.leaf = leaf.count ? -(int32_t)((leaf.tris_ptr << 3) | leaf.count) : 0;
                        }
                    }
                    SP_END(6, sp_m6);
                }
            }
            if (st == 1 && !wait && !cq_running(q)) st = 2;
            const int waiting = __popcll(__ballot(st == 2)) + (exhausted ? 0 : __popcll(__ballot(st == 0)));
            if (waiting >= leave_at || !__ballot(st == 1)) break;
        }
#else
        for (;;) {
            for (;;) {   // nodes: every lane that holds no leaf descends, until the lanes with a leaf are the majority
                const bool at_node = st == 1 && q.leaf == 0 && q.s > 0;
                if (!__ballot(at_node)) break;
                SP_BEGIN();
                if (at_node) cq_node_step(ns, env, stack, q);
                SP_END(4, __ballot(at_node));
                if (WTGPU_LEAF_DEN * __popcll(__ballot(st == 1 && q.leaf != 0)) >= WTGPU_LEAF_NUM * running) break;
            }
            SP_BEGIN();
            const unsigned long long sp_m5 = __ballot(st == 1 && q.leaf != 0);
            (void)sp_m5;
            if (st == 1 && q.leaf != 0) cq_leaf_step(a.sc, env, stack, tris2, q);
            SP_END(5, sp_m5);
            if (st == 1 && !cq_running(q)) st = 2;
            const int waiting = __popcll(__ballot(st == 2)) + (exhausted ? 0 : __popcll(__ballot(st == 0)));
            if (waiting >= leave_at || !__ballot(st == 1)) break;
        }
#endif
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
#ifdef WTGPU_SM_PROF
    if (lane == 0)
        for (int i = 0; i < 8; ++i) {
            atomicAdd(a.st.counters + kNumCounters + 32 + i, sp_t[i]);
            atomicAdd(a.st.counters + kNumCounters + 40 + i, sp_l[i]);
            atomicAdd(a.st.counters + kNumCounters + 48 + i, sp_n[i]);
        }
#endif
}
// stage `it` (>= 1) of the policy: consumes the policy queue (count [it & 1]), fills the cone queue (count [it & 1])
__global__ void __launch_bounds__(kBlock, WTGPU_LB_STAGE) k_tr_policy(launch_args_t a, uint32_t it) {
    __shared__ stack_entry_t lds[kBlock];   // (aw_next begins its query on a stack: one entry)
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_TPOL_COUNT0 + (it & 1u)];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl[CTL_TCONE_HEAD] = 0;
        ctl[CTL_TPOL_COUNT0 + ((it + 1u) & 1u)] = 0;
    }
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    const stack_ref_t stack = make_stack_ref(lds + threadIdx.x, kBlock, 1, 1, nullptr);
    const bool rt = a.sc.sensor.ray_trace_only || a.sc.opts.force_ray_tracing;
    const int lane = threadIdx.x & 63;
    uint32_t* stage = trace_stage_words(a);
    const uint32_t* polq = trace_pol_queue(a);
    uint32_t* coneq = trace_cone_queue(a);
    for (;;) {
        const uint32_t base = wave_grab(ctl + CTL_TPOL_HEAD);
        if (base >= n) break;
        const uint32_t qi = base + (uint32_t)lane;
        bool fin = false, query = false;
        uint32_t w = 0;
        trav_result_t r;
        if (qi < n) {
            w = polq[qi];
            trace_stage_t ts;
            soa_load(stage, kStageWords, w, ts);
            query = tr_policy_run(a.sc, ts.env, rt, stack, ts.aw, r);
            fin = !query;
            if (query) soa_store(stage + offsetof(trace_stage_t, aw) / 4, kStageWords, w, ts.aw);
        }
        wave_append(coneq, ctl + CTL_TCONE_COUNT0 + (it & 1u), query, w);
        tr_finish(a, ctl, ctr, fin, w, r);
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}
// what is left in the policy queue (count [it & 1]) after the last cone stage: policy and queries to the end, one lane per walk
__global__ void __launch_bounds__(kBlock, WTGPU_LB_TRACE) k_tr_tail(launch_args_t a, uint32_t it) {
    __shared__ stack_entry_t lds[kLdsStack * kBlock];
    uint32_t* ctl = a.st.ctl;
    const uint32_t n = ctl[CTL_TPOL_COUNT0 + (it & 1u)];
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl[CTL_TCONE_COUNT0] = ctl[CTL_TCONE_COUNT1] = 0;   // for the next round's k_tr_axis
    bdpt_counters_t ctr;
    memset(&ctr, 0, sizeof(ctr));
    stack_entry_t spill[kSpillStack];
    stack_ref_t stack;
    lds_stack(lds, spill, stack);
    const lane_nodes_t ns = lane_nodes(a.sc);
    const bool rt = a.sc.sensor.ray_trace_only || a.sc.opts.force_ray_tracing;
    const int lane = threadIdx.x & 63;
    uint32_t* stage = trace_stage_words(a);
    const uint32_t* polq = trace_pol_queue(a);
    for (;;) {
        const uint32_t base = wave_grab(ctl + CTL_TPOL_HEAD);
        if (base >= n) break;
        const uint32_t qi = base + (uint32_t)lane;
        const bool have = qi < n;
        uint32_t w = 0;
        trav_result_t r;
        if (have) {
            w = polq[qi];
            trace_stage_t ts;
            soa_load(stage, kStageWords, w, ts);
            uint32_t* slot = a.st.tris + (size_t)w * kTriListWords;
            const uint_list_t tris{slot, 1u, (a.collect_list & 1u) ? kMaxConeTris : 0u, reinterpret_cast<float*>(slot + kMaxConeTris)};
            cone_query_t q;
            for (;;) {
                const int need = aw_next(a.sc, ts.env, rt, stack, ts.aw, q, r);
                if (need == AW_FINAL) break;
                if (need == AW_TEST) {
                    aw_test_done(ts.aw, cone_attempt_too_short_by(a.sc, ts.env, ts.aw.cand, ts.aw.sr, ts.aw.min_df_prog));
                    continue;
                }
                while (cq_running(q)) {
                    while (q.s > 0 && q.leaf == 0) cq_node_step(ns, ts.env, stack, q);
                    if (q.leaf != 0) cq_leaf_step(a.sc, ts.env, stack, tris, q);
                }
                cq_end(ts.env, tris, q);
                if (aw_query_done(a.sc, ts.env, ts.aw, q.rec, r)) break;
            }
        }
        tr_finish(a, ctl, ctr, have, w, r);
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
}

// Heavy traversals: one wavefront (64-thread block) per walk, persistent blocks pulling from the heavy queue.
__global__ void __launch_bounds__(64, WTGPU_LB_HEAVY) k_trace_heavy(launch_args_t a) {
    __shared__ coop_shared_t sh;
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
        const uint32_t item = wave_grab_item(head);
        if (item >= n) break;
        const uint32_t w = hq[item];
        const walk_trace_in_t wk = walk_load_trace_in(a.st.walks, a.st.walk_words, w);   // uniform address: broadcast
        const uint_list_t tris{a.st.tris + (size_t)w * kTriListWords, 1u, (a.collect_list & 1u) ? kMaxConeTris : 0u};   // see k_trace
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
                                                &axis, !(a.collect_list & 1u) || (a.collect_list & 2u), a.heavy_probe != 0, a.heavy_cache ? short0 : kInvalid, a.heavy_cache ? wk.prev_offset_tuid : kInvalid, a.heavy_cache != 0);
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
            if (a.collect_list & 1u) ctr.cone_tri_overflow += tr2.overflow;
        }
    }
    if (a.count_stats) flush_counters(a.st.counters, ctr);
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

}   // namespace wtk
