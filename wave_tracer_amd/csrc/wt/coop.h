// wave_tracer_amd — wavefront-cooperative cone traversal for *heavy* beam queries (device only).
//
// Why: the cost of a cone query is wildly non-uniform.  In the cornell-box workload 1.2 % of the path segments account
// for ~70 % of all cone-triangle tests (a fat beam whose interaction slab covers a finely tessellated mesh visits
// 10^4..10^6 triangles, src/ads/bvh8w.cpp:134-136 "by far the slowest part of cone traversal / TODO: vectorize").  One
// lane doing that alone stalls its whole wavefront for seconds.  Here the 64 lanes of ONE wavefront serve ONE query:
//   * an internal node's 8 child boxes are tested by lanes 0..7 (cone x AABB, bvh8w.cpp:187-230) and pushed, far-first,
//     on a wave-shared LDS stack with a rank computed by cross-lane compares (the reference's insertion sort, 45-57);
//   * a subtree holding <= 64 triangles is treated as a leaf and its triangles are tested one per lane
//     (exact cone-triangle test, math/intersect/cone.hpp:550-626); closest distance by a wave min-reduce, hit
//     triangles appended with ballot + prefix popcount;
//   * the search slab shrinks after every batch with hits exactly like intersection_record_work_t::search_range.
// The policy loop of integrator::traverse (traversal.hpp:94-172) runs wave-uniformly around it; its ray segments are
// executed redundantly by all lanes (uniform control flow, same cost as one lane).
//
// Result semantics: identical closest distance / front-face flag; the triangle list is a superset-compatible variant
// of the sequential one (the reference's list is traversal-order dependent).
#pragma once
#if defined(__HIPCC__)
#include "bvh.h"
#include "gauss.h"

namespace wt {

constexpr int kCoopStack = 512;
// Children a full cooperative stack could not hold (never seen; a dropped child would be a silently wrong region) are COUNTED, in the scene's own
// counter block: every kernel that declares one of the shared structs below points its `dropped` member at the scene's slot first
// (coop_set_dropped_counter); wtgpu_get_counters reports the slot as traversal_stack_dropped and the full-size GPU tests assert it zero.
#ifndef WTGPU_COOP_LEAF_TRIS
#define WTGPU_COOP_LEAF_TRIS 256
#endif
// Subtrees of at most this many triangles are tested whole, 64 triangles per step, instead of being descended into (measured,
// k_trace_heavy stream-summed per pass: 16 / 32 / 64 / 128 / 256 -> 293 / 201 / 193 / 177 / 169 ms: the wide beams this kernel serves
// meet most of such a subtree anyway and a batch of cheap filter tests costs less than the node steps it replaces).  The region walks
// (coop_gather) keep 64: with `edges_only` they prune by edge_mask at node level, which a larger threshold would bypass (k_edges
// 86 -> 102 ms at 128).
constexpr uint32_t kCoopLeafTris = WTGPU_COOP_LEAF_TRIS;
constexpr uint32_t kCoopGatherLeafTris = 64;
constexpr uint32_t kCoopTriBuf = 64 + 8 * kCoopLeafTris;   // buffered triangle ids: < 64 pending + 8 entries x <= 64 triangles

#ifndef WTGPU_COOP_FLUSH_AT
#define WTGPU_COOP_FLUSH_AT 12
#endif
constexpr uint32_t kCoopFlushAt = WTGPU_COOP_FLUSH_AT;   // survivors worth an exact-test pass before the stack is empty
constexpr uint32_t kCoopSurvCap = 128;   // candidates that passed the cheap filter and await the exact cone-triangle test

#ifndef WT_COOP_SPHERES
#define WT_COOP_SPHERES 1
#endif
#ifndef WT_COOP_SPHERE_PREFETCH
#define WT_COOP_SPHERE_PREFETCH 1
#endif
constexpr uint32_t kCoopSpherePrefetch = WT_COOP_SPHERE_PREFETCH;   // batches of 64 bounding spheres fetched ahead of their tests (coop_cone_query, phase B1a)
// The triangles' bounding spheres (centre, radius: tri_bounding_sphere, wt/cone.h) are uploaded right behind the scene's triangles, in the
// same allocation (wtgpu_scene_upload) — reached through tri_geo, so that the kernels' launch block does not grow by another pointer.
WT_D const float4* coop_tri_spheres(const scene_t& sc) { return reinterpret_cast<const float4*>(sc.tri_geo + sc.n_tris); }
struct coop_shared_t {
    unsigned long long* dropped;   // the scene's counter of children a full stack could not hold
    stack_entry_t stack[kCoopStack];
    uint32_t tri_buf[kCoopTriBuf];
    uint32_t surv[kCoopSurvCap];
    float hit_dist[64];   // cone-hit distance of every listed triangle (list capacity kMaxConeTris = 64)
};
// the region walks' (coop_gather, coop_split) LDS: the same with the smaller candidate buffer their leaf threshold needs
struct coop_gather_shared_t {
    unsigned long long* dropped;
    stack_entry_t stack[kCoopStack];
    uint32_t tri_buf[64 + 8 * 64];
    uint32_t surv[kCoopSurvCap];
};
// LDS of the edge-collecting gather only (k_edges): kept out of coop_shared_t so that the traversal kernels, whose wavefronts wait on
// memory most of the time, fit twice as many wavefronts per CU (LDS is what bounds their occupancy).
struct coop_edges_t {
    uint32_t edge_ids[96];   // coop_gather: sorted classified-edge set of an interaction region (scenes with more than kCoopEdgeBits edges)
    uint32_t edge_bits[1024];   // coop_gather: the same set as a bitmap over the scene's edge ids (unbounded, sorted and de-duplicated for free)
};
constexpr uint32_t kCoopEdgeBits = 32768;

template <class SH>
WT_D void coop_set_dropped_counter(SH& sh, unsigned long long* slot) {
    if (threadIdx.x == 0) sh.dropped = slot;
    __syncthreads();
}
WT_D float wave_min(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}

// cone x AABB of child i (bvh8w.cpp:187-230); returns hit + tmin
WT_D bool cone_child_test(const bvh8_node_t& n, int i, vec3 ro, vec3 rd, vec3 rinvd, bool sx, bool sy, bool sz, float ta, float ix,
                                       const range_t& range, float& tmin_out) {
    float ominx = n.minx[i] - ro.x, ominy = n.miny[i] - ro.y, ominz = n.minz[i] - ro.z;
    float omaxx = n.maxx[i] - ro.x, omaxy = n.maxy[i] - ro.y, omaxz = n.maxz[i] - ro.z;
    const bool outside = cone_box_outside(ominx, ominy, ominz, omaxx, omaxy, omaxz, rd, ta, ix, range);
    const float bx = sx ? ominx : omaxx, by = sy ? ominy : omaxy, bz = sz ? ominz : omaxz;
    const float dot_d_b = rd.x * bx + rd.y * by + rd.z * bz;
    const float maxz = clampf(dot_d_b, 0.f, range.max);
    const float enlr = fmaf(maxz, ta, ix);
    ominx -= enlr;
    ominy -= enlr;
    ominz -= enlr;
    omaxx += enlr;
    omaxy += enlr;
    omaxz += enlr;
    const float dminx = (sx ? omaxx : ominx) * rinvd.x, dmaxx = (sx ? ominx : omaxx) * rinvd.x;
    const float dminy = (sy ? omaxy : ominy) * rinvd.y, dmaxy = (sy ? ominy : omaxy) * rinvd.y;
    const float dminz = (sz ? omaxz : ominz) * rinvd.z, dmaxz = (sz ? ominz : omaxz) * rinvd.z;
    float tmin = 0.f, tmax = dmaxx;
    tmin = fmaxf_(tmin, dminx);
    tmax = fminf_(tmax, dmaxy);
    tmin = fmaxf_(tmin, dminy);
    tmax = fminf_(tmax, dmaxz);
    tmin = fmaxf_(tmin, dminz);
    tmin_out = tmin;
    return !outside && tmin <= tmax && tmax >= range.min && tmin <= range.max && !(tmin >= range.max);
}

// One wavefront, one cone query.  Must be called by all 64 lanes of a 64-thread block with identical arguments.
//   * up to EIGHT stack entries are popped per step: lane l serves entry l/8, child l%8, so all 64 lanes run the cone x AABB
//     test (bvh8w.cpp:187-230) at once; hits are pushed with ballot/popcount, the children of the nearest popped node on top,
//     far-first within a node (the reference's order, bvh8w.cpp:45-57, is kept per node);
//   * leaves (and subtrees with <= 64 triangles) are not tested one by one: their triangle ranges are buffered in LDS and
//     tested 64 triangles per step, whatever leaf they came from (leaves hold ~4 triangles: testing them leaf by leaf would
//     leave 60 lanes idle);
//   * any_hit = false: closest distance by a wave min-reduce, hit triangles appended with ballot + prefix popcount, the
//     search slab shrinks after every batch with hits exactly like intersection_record_work_t::search_range;
//     any_hit = true : the any-hit probe (bvh_cone_any_hit): returns at the first batch with a hit.
template <bool any_hit>
WT_D bool coop_cone_query(const scene_t& sc, const cone_t& cone, const range_t& searchrange, float z_scale, coop_shared_t& sh,
                                       const uint_list_t& tris, cone_hit_t& rec, unsigned long long* prof = nullptr, float min_progress = -WT_INF) {
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, sub = lane & 7;
    rec.dist = WT_INF;
    rec.front_face = 0;
    rec.ntris = 0;
    rec.overflow = 0;
    rec.aborted = 0;
    rec.too_short = 0;
    rec.short_tuid = kInvalid;
    if (sc.n_nodes == 0) return false;
    const vec3 ro = cone.o, rd = cone.d;
    const vec3 rinvd{1.f / rd.x, 1.f / rd.y, 1.f / rd.z};
    const bool sx = __builtin_signbit(rinvd.x), sy = __builtin_signbit(rinvd.y), sz = __builtin_signbit(rinvd.z);
    const float ta = cone.tan_alpha, ix = cone.x0;
    range_t range = any_hit ? searchrange : cone_search_range(cone, searchrange, rec.dist, z_scale);
    float slab_max = range.max;   // far end of the current interaction slab (list membership); range.max may be pruned below it
    int s = 1;
    uint32_t leaf_total = 0, nsurv = 0;
    if (lane == 0) sh.stack[0] = stack_entry_t{0.f, 1};
    __syncthreads();
    // Drops the listed triangles whose cone-hit distance lies beyond zmax.  The sequential traversal tests near triangles first
    // and shrinks its slab at once, so its list holds (almost) only triangles inside the final slab [closest, closest + z_scale *
    // axis]; a 64-wide batch is tested against the slab as it was before the batch.  Compacting whenever the slab shrinks keeps
    // the list = the triangles that meet the cone inside the current interaction region (the sequential list may keep a few more
    // that it saw before its slab shrank) and keeps the bounded list for the triangles that matter.
    auto compact = [&](float zmax) __attribute__((always_inline)) {
        if (rec.ntris == 0) return;
        __syncthreads();
        const bool mine = (uint32_t)lane < rec.ntris;
        const uint32_t val = mine ? tris[lane] : 0u;
        const float dv = mine ? sh.hit_dist[lane] : 0.f;
        const bool keep = mine && !(dv > zmax);
        const unsigned long long km = __ballot(keep);
        __syncthreads();
        if (keep) {
            const uint32_t np_ = (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
            tris[np_] = val;
            sh.hit_dist[np_] = dv;
        }
        rec.ntris = (uint32_t)__popcll(km);
        __syncthreads();
    };
    // phase B2: exact cone-triangle test of the buffered survivors, a full wave at a time; TRUE = any-hit probe satisfied
    auto flush = [&]() __attribute__((always_inline)) -> bool {
        __syncthreads();
        bool any = false;
        for (uint32_t b2 = 0; b2 < nsurv && !any; b2 += 64) {
#ifdef WTGPU_COOP_PROF
            if (prof) prof[9] += 1;
#endif
            const uint32_t k2 = b2 + lane;
            bool hit = false, ff = false;
            float d = WT_INF;
            uint32_t t2 = 0;
            if (k2 < nsurv) {
                t2 = sh.surv[k2];
                const tri_geo_t tri = sc.tri_geo[t2];
                ff = dot(tri.n, -rd) > 0.f;
                cone_tri_hit_t ht;
                if (intersect_cone_tri<any_hit>(cone, tri.a, tri.b, tri.c, tri.n, range, ht) && !(ht.dist > range.max)) {
                    hit = true;
                    d = ht.dist;
                }
            }
            const unsigned long long mask = __ballot(hit);
            if (!mask) continue;
            if (any_hit) {
                rec.short_tuid = (uint32_t)__shfl((int)t2, __ffsll((long long)mask) - 1, 64);
                any = true;
                break;
            }
            const float dm = wave_min(d);
            if (dm < rec.dist) {
                const unsigned long long m2 = __ballot(hit && d == dm);
                const int src = __ffsll((long long)m2) - 1;
                rec.front_face = (uint32_t)__shfl((int)ff, src, 64);
                rec.dist = dm;
                if (rec.dist - searchrange.min < min_progress) {   // decided: too near to be accepted (see bvh_traverse_cone)
                    rec.short_tuid = (uint32_t)__shfl((int)t2, src, 64);
                    rec.too_short = 1;
                    any = true;
                    break;
                }
                range = cone_search_range(cone, searchrange, rec.dist, z_scale);
                slab_max = range.max;
                compact(slab_max);   // the slab shrank: listed triangles beyond it leave (and make room)
            }
            const bool hit2 = hit && !(d > slab_max);   // this batch was tested against the slab as it was before it
            const unsigned long long mask2 = __ballot(hit2);
            const uint32_t pos = rec.ntris + (uint32_t)__popcll(mask2 & ((1ull << lane) - 1ull));
            if (hit2 && pos < tris.cap) {
                tris[pos] = t2;
                sh.hit_dist[pos & 63u] = d;
            }
            const uint32_t total = rec.ntris + (uint32_t)__popcll(mask2);
            const uint32_t newn = total < tris.cap ? total : tris.cap;
            rec.overflow += total - newn;
            rec.ntris = newn;
            if (rec.overflow > 0) range.max = fminf_(range.max, rec.dist);   // bounded-list regime (traversal pruning only), see bvh.h
        }
        nsurv = 0;
        __syncthreads();
        return any;
    };
#ifdef WTGPU_COOP_PROF
    long long cp_t = clock64();
#define CP(k)                                              \
    do {                                                   \
        const long long n_ = clock64();                    \
        if (prof) prof[k] += (unsigned long long)(n_ - cp_t); \
        cp_t = n_;                                         \
    } while (0)
#else
#define CP(k) ((void)0)
#endif
    for (;;) {
        // ---- phase A: expand up to 8 stack entries per step until >= 64 triangles are buffered (or the stack is empty)
        const long long ta0 = prof ? clock64() : 0;
        while (s > 0 && leaf_total < 64u) {
            const int np = s < 8 ? s : 8;
#ifndef WTGPU_COOP_PROF
            if (prof) prof[7] += 1ull + ((unsigned long long)np << 32);
#endif
            stack_entry_t e{0.f, 0};
            if (grp < np) e = sh.stack[s - 1 - grp];
            s -= np;
            __syncthreads();   // everyone has read its entry before the slots are overwritten
            CP(0);
            const bool live = grp < np && (any_hit || e.t < range.max);
            bool leafish = false, h = false;
            uint32_t t0 = 0, cnt = 0;
            float tmin = 0.f;
            int32_t cp = 0;
            if (live) {
                if (e.ptr < 0) {
                    const bvh8_leaf_t leaf = bvh_leaf_of(e.ptr);
                    t0 = leaf.tris_ptr;
                    cnt = leaf.count;
                    leafish = true;
                } else {
                    // all loads of the node are issued before anything depends on them (one memory latency per step, not two)
                    const bvh8_node_t& node = sc.nodes[e.ptr - 1];
                    const uint32_t ntc = node.tris_count, nts = node.tris_start;
                    cp = node.child[sub];
                    const bool hc = cone_child_test(node, sub, ro, rd, rinvd, sx, sy, sz, ta, ix, range, tmin);
                    if (ntc <= kCoopLeafTris) {
                        t0 = nts;
                        cnt = ntc;
                        leafish = true;
                    } else {
                        h = hc && cp != 0;
                    }
                }
            }
            CP(1);
            // push the child hits: group 0 served the top (nearest) entry, its children go on top
            const unsigned long long hm = __ballot(h);
            if (hm) {
                const uint32_t gbits = (uint32_t)(hm >> (grp * 8)) & 0xffu;
                int rank = 0;   // far-first within the node
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float tj = __shfl(tmin, (lane & ~7) + j, 64);
                    if (((gbits >> j) & 1u) && (tj > tmin || (tj == tmin && j < sub))) ++rank;
                }
                const int above = grp < 7 ? __popcll(hm >> ((grp + 1) * 8)) : 0;   // hits of the groups serving deeper entries
                const int pos = s + above + rank;
                if (h && pos < kCoopStack) sh.stack[pos] = stack_entry_t{tmin, cp};
                else if (h) atomicAdd(sh.dropped, 1ull);   // reported: wtgpu_counters::traversal_stack_dropped
                const int total = s + __popcll(hm);
                s = total < kCoopStack ? total : kCoopStack;
            }
            CP(2);
            // buffer the leaf ranges
            const unsigned long long lm = __ballot(leafish && sub == 0 && cnt > 0);
            unsigned long long m = lm;
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                uint32_t c = (uint32_t)__shfl((int)cnt, src, 64);
                const uint32_t t = (uint32_t)__shfl((int)t0, src, 64);
                if (c > kCoopLeafTris) c = kCoopLeafTris;   // cannot happen with this builder (leaves hold <= MAX_LEAF triangles)
                for (uint32_t q = (uint32_t)lane; q < c; q += 64u) sh.tri_buf[leaf_total + q] = t + q;
                leaf_total += c;
            }
            __syncthreads();
            CP(3);
        }
        if (prof) prof[5] += (unsigned long long)(clock64() - ta0);
        const long long tb0 = prof ? clock64() : 0;
        // ---- phase B1: cheap conservative filter (slab + lateral rejection, cone_tri_maybe) over the buffered candidates.  97 % of
        // them fail it; testing them with the exact routine would make every 64-wide batch pay its ~20x more expensive
        // plane/edge path for the 2-3 lanes that need it.  Survivors are compacted into an LDS list instead.
        bool found_any = false;
#ifdef WTGPU_COOP_PROF
        if (prof) { prof[10] += 1; prof[11] += leaf_total; prof[8] += (leaf_total + 63u) / 64u; }
#endif
#if WT_COOP_SPHERES
        // B1a (round 4): the candidates' BOUNDING SPHERES first (cone_sphere_maybe: 16 B and ~20 operations per triangle; the scene's spheres lie
        // behind its triangles, coop_tri_spheres); what passes is compacted IN PLACE at the front of sh.tri_buf (a survivor's slot lies at or
        // below a slot that was already read) and only that goes through the filter below (36 B, ~150 operations per triangle, whichever lane
        // needs them): k_trace_heavy 139 -> 127 ms per three steps, exclusive (run r4q).  An item averages 3.8 entries into this phase with
        // 1,150 candidates in 19 batches, and 2.3 exact-test batches (WTGPU_COOP_PROF).  The spheres of kCoopSpherePrefetch batches can be
        // fetched before the first of them is tested (the empty asm pins the loaded values: the compiler would otherwise sink every load into
        // the block that tests it; the loop holds no calls, so it unrolls without pushing the kernel over the inliner's budget): measured 1 / 4 /
        // 8 batches ahead -> 127 / - / 130 ms, i.e. the batches' round trips are not what this phase waits for.  Default 1.
        {
            const float4* spheres = coop_tri_spheres(sc);
            uint32_t n2 = 0;
            for (uint32_t base0 = 0; base0 < leaf_total; base0 += 64u * kCoopSpherePrefetch) {
                float4 bs[kCoopSpherePrefetch];
                uint32_t tu[kCoopSpherePrefetch];
#pragma unroll
                for (uint32_t j = 0; j < kCoopSpherePrefetch; ++j) {
                    const uint32_t k = base0 + 64u * j + lane;
                    tu[j] = 0;
                    bs[j] = float4{0.f, 0.f, 0.f, 0.f};
                    if (k < leaf_total) {
                        tu[j] = sh.tri_buf[k];
                        bs[j] = spheres[tu[j]];
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < kCoopSpherePrefetch; ++j) asm volatile("" : "+v"(bs[j].x), "+v"(bs[j].y), "+v"(bs[j].z), "+v"(bs[j].w));
                __syncthreads();   // every slot of this chunk has been read before any is overwritten
#pragma unroll
                for (uint32_t j = 0; j < kCoopSpherePrefetch; ++j) {
                    const uint32_t k = base0 + 64u * j + lane;
                    const bool p1 = k < leaf_total && cone_sphere_maybe(cone, vec3{bs[j].x, bs[j].y, bs[j].z}, bs[j].w, range);
                    const unsigned long long pm1 = __ballot(p1);
                    if (p1) sh.tri_buf[n2 + (uint32_t)__popcll(pm1 & ((1ull << lane) - 1ull))] = tu[j];
                    n2 += (uint32_t)__popcll(pm1);
                }
            }
            leaf_total = n2;
            __syncthreads();
            CP(7);
        }
#endif
        for (uint32_t base = 0; base < leaf_total && !found_any; base += 64) {
            const uint32_t k = base + lane;
            bool pass = false;
            uint32_t tuid = 0;
            if (k < leaf_total) {
                tuid = sh.tri_buf[k];
                const tri_geo_t tri = sc.tri_geo[tuid];
                pass = cone_tri_maybe(cone, tri.a, tri.b, tri.c, range);
            }
            const unsigned long long pm = __ballot(pass);
            if (pm) {
                const uint32_t pos = nsurv + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
                if (pass && pos < kCoopSurvCap) sh.surv[pos] = tuid;
                nsurv += (uint32_t)__popcll(pm);   // < 64 + 64 <= kCoopSurvCap: a full wave is flushed right away
            }
            CP(7);
            if (nsurv >= 64u) found_any = flush();
            CP(6);
        }
        leaf_total = 0;
        // ---- phase B2 at the end of a round when the query is about to finish (stack empty) or enough survivors wait: one more
        // round of expansion + filtering costs ~1/5 of an exact-test pass, so a handful of survivors is worth waiting for.
        CP(7);
        if (!found_any && nsurv > 0 && (s == 0 || nsurv >= kCoopFlushAt)) found_any = flush();
        CP(6);
        __syncthreads();
        if (prof) prof[6] += (unsigned long long)(clock64() - tb0);
        if (found_any && (any_hit || rec.too_short)) return true;
        if (s == 0) break;
    }
    return rec.ntris > 0;
}

WT_D void coop_cone(const scene_t& sc, const cone_t& cone, const range_t& searchrange, float z_scale, coop_shared_t& sh,
                                 const uint_list_t& tris, cone_hit_t& rec, unsigned long long* prof = nullptr, float min_progress = -WT_INF) {
    coop_cone_query<false>(sc, cone, searchrange, z_scale, sh, tris, rec, prof, min_progress);
}
// Wave-cooperative any-hit probe (see bvh_cone_any_hit).
WT_D bool coop_cone_any(const scene_t& sc, const cone_t& cone, const range_t& range, coop_shared_t& sh, unsigned long long* prof = nullptr,
                                     uint32_t* hit_tuid = nullptr) {
    cone_hit_t rec;
    const uint_list_t none{nullptr, 0, 0};
    const bool hit = coop_cone_query<true>(sc, cone, range, 0.f, sh, none, rec, prof);
    if (hit_tuid) *hit_tuid = rec.short_tuid;
    return hit;
}

// ---- interaction-region gather for beams whose footprint holds more triangles than the bounded list (kMaxConeTris) -----------
// The interaction step needs three things from the triangle list of a diffusive hit (bdpt_walk_step): the triangle under the beam
// axis (a BVH ray query when the list overflowed), the fraction of the beam's power the front-facing triangles intercept, and the
// set of classified (silhouette) edges.  The last two are sums / unions over the list, so for a region that does not fit the list
// one wavefront walks the region once more — every triangle that meets the traced cone inside the final slab — and accumulates
// them directly: the result is that of an unbounded list (the reference's std::vector), whatever the triangle count.
struct gather_out_t {
    double flux;   // f64 sum (see bdpt_walk_step)
    uint32_t n_edges, edge_overflow;
    uint32_t n_tris;   // triangles of the region (met by the cone inside the slab)
};
template <class SH>
WT_D gather_out_t coop_gather(const scene_t& sc, const cone_t& tcone, const range_t& slab, const cone_t& envelope, const frame_t& beam_frame,
                                           const range_t& izr, vec2 sigma, bool want_front, SH& sh, bool do_flux, bool do_edges,
                                           unsigned long long* stats = nullptr, int32_t root = 1, coop_edges_t* eg = nullptr) {
    uint32_t* edges = eg ? eg->edge_ids : nullptr;   // eg: required when do_edges
    const uint32_t edge_cap = 96;
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, sub = lane & 7;
    gather_out_t out{0.0, 0u, 0u, 0u};
    const bool edges_only = do_edges && !do_flux;
    // edge set as an LDS bitmap whenever the scene's edge ids fit (n_edges <= 32768): no capacity limit, ids come out sorted; the
    // caller reads eg->edge_bits (coop_edge_count / coop_edge_write).  Larger scenes: the sorted 96-entry list (overflow counted).
    const bool bitmap = do_edges && sc.n_edges <= kCoopEdgeBits;
    if (bitmap) {
        for (uint32_t j = threadIdx.x & 63; j < (sc.n_edges + 31u) / 32u; j += 64) eg->edge_bits[j] = 0u;
        __syncthreads();
    }
    if (sc.n_nodes == 0) return out;
    const vec3 ro = tcone.o, rd = tcone.d;
    const vec3 rinvd{1.f / rd.x, 1.f / rd.y, 1.f / rd.z};
    const bool sx = __builtin_signbit(rinvd.x), sy = __builtin_signbit(rinvd.y), sz = __builtin_signbit(rinvd.z);
    const float ta = tcone.tan_alpha, ix = tcone.x0;
    const float csz = centre(izr);
    int s = 1;
    uint32_t leaf_total = 0, nsurv = 0;
    if (lane == 0) sh.stack[0] = stack_entry_t{0.f, root};   // (root: a subtree of the region, k_flux_tasks)
    __syncthreads();
    auto flush = [&]() __attribute__((always_inline)) {
        __syncthreads();
        if (stats) stats[1] += nsurv;
        for (uint32_t b2 = 0; b2 < nsurv; b2 += 64) {
            const uint32_t k2 = b2 + lane;
            float contrib = 0.f;
            uint32_t eid[3] = {kInvalid, kInvalid, kInvalid};
            bool member = false;
            if (k2 < nsurv) {
                const uint32_t t2 = sh.surv[k2];
                const tri_geo_t tri = sc.tri_geo[t2];
                cone_tri_hit_t ht;
                // membership = "meets the cone inside the slab": the any-hit form decides at the first contained vertex (most triangles
                // of a large region lie inside the beam)
                if (intersect_cone_tri<true>(tcone, tri.a, tri.b, tri.c, tri.n, slab, ht) && !(ht.dist > slab.max)) {
                    member = true;
                    if (do_edges) {
                        const tri_meta_t m = sc.tri_meta[t2];
                        eid[0] = m.edge[0];
                        eid[1] = m.edge[1];
                        eid[2] = m.edge[2];
                    }
                    if (do_flux && (dot(tri.n, -rd) > 0.f) == want_front) {   // find_closest_triangle's footprint integral (bdpt.h)
                        contrib = region_local_triangle_flux(envelope, izr, csz, sigma, to_local(beam_frame, tri.a - envelope.o),
                                                             to_local(beam_frame, tri.b - envelope.o), to_local(beam_frame, tri.c - envelope.o));
                    }
                }
            }
            out.n_tris += (uint32_t)__popcll(__ballot(member));
            // f64 butterfly sum (order-independent to ~1e-16)
            double csum = (double)contrib;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) csum += __shfl_xor(csum, off, 64);
            out.flux += csum;
            // classified edges are rare (a few hundred in a 170K-triangle scene): insert them one by one into the sorted LDS
            // list, all lanes cooperating on the search
            if (bitmap) {
#pragma unroll
                for (int e = 0; e < 3; ++e)
                    if (eid[e] != kInvalid) atomicOr(&eg->edge_bits[eid[e] >> 5], 1u << (eid[e] & 31u));
            }
            unsigned long long em = bitmap ? 0ull : __ballot(eid[0] != kInvalid || eid[1] != kInvalid || eid[2] != kInvalid);
            while (em) {
                const int src = __ffsll((long long)em) - 1;
                em &= em - 1;
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const uint32_t id = (uint32_t)__shfl((int)eid[e], src, 64);
                    if (id == kInvalid) continue;
                    // position = number of entries smaller than id; present = some entry equals id   (entries lane, lane+64, ...)
                    uint32_t less = 0;
                    bool present = false;
                    for (uint32_t j = (uint32_t)lane; j < out.n_edges; j += 64) {
                        const uint32_t v = edges[j];
                        less += v < id ? 1u : 0u;
                        present = present || v == id;
                    }
                    if (__ballot(present)) continue;
                    if (out.n_edges >= edge_cap) {
                        out.edge_overflow++;
                        continue;
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) less += (uint32_t)__shfl_xor((int)less, off, 64);
                    // shift the tail up by one (read everything first), then insert
                    uint32_t mv[2];
                    int nm = 0;
                    for (uint32_t j = less + (uint32_t)lane; j < out.n_edges && nm < 2; j += 64) mv[nm++] = edges[j];
                    __syncthreads();
                    nm = 0;
                    for (uint32_t j = less + (uint32_t)lane; j < out.n_edges && nm < 2; j += 64) edges[j + 1] = mv[nm++];
                    if (lane == 0) edges[less] = id;
                    out.n_edges++;
                    __syncthreads();
                }
            }
        }
        nsurv = 0;
        __syncthreads();
    };
    for (;;) {
        while (s > 0 && leaf_total < 64u) {
            const int np = s < 8 ? s : 8;
            stack_entry_t e{0.f, 0};
            if (grp < np) e = sh.stack[s - 1 - grp];
            s -= np;
            __syncthreads();
            const bool live = grp < np;
            bool leafish = false, h = false;
            uint32_t t0 = 0, cnt = 0;
            float tmin = 0.f;
            int32_t cp = 0;
            if (live) {
                if (e.ptr < 0) {
                    const bvh8_leaf_t leaf = bvh_leaf_of(e.ptr);
                    t0 = leaf.tris_ptr;
                    cnt = leaf.count;
                    leafish = true;
                } else {
                    const bvh8_node_t& node = sc.nodes[e.ptr - 1];
                    const uint32_t ntc = node.tris_count, nts = node.tris_start;
                    cp = node.child[sub];
                    // edges only: descend only into subtrees that hold classified edges (bvh8_node_t::edge_mask)
                    const bool hc = (!edges_only || ((node.edge_mask >> sub) & 1u)) && cone_child_test(node, sub, ro, rd, rinvd, sx, sy, sz, ta, ix, slab, tmin);
                    if (ntc <= kCoopGatherLeafTris) {
                        t0 = nts;
                        cnt = ntc;
                        leafish = true;
                    } else {
                        h = hc && cp != 0;
                    }
                }
            }
            const unsigned long long hm = __ballot(h);
            if (hm) {
                const int pos = s + __popcll(hm & ((1ull << lane) - 1ull));   // order is irrelevant here
                if (h && pos < kCoopStack) sh.stack[pos] = stack_entry_t{tmin, cp};
                else if (h) atomicAdd(sh.dropped, 1ull);   // reported: wtgpu_counters::traversal_stack_dropped
                const int total = s + __popcll(hm);
                s = total < kCoopStack ? total : kCoopStack;
            }
            unsigned long long m = __ballot(leafish && sub == 0 && cnt > 0);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                uint32_t c = (uint32_t)__shfl((int)cnt, src, 64);
                const uint32_t t = (uint32_t)__shfl((int)t0, src, 64);
                if (c > kCoopGatherLeafTris) c = kCoopGatherLeafTris;
                for (uint32_t q = (uint32_t)lane; q < c; q += 64u) sh.tri_buf[leaf_total + q] = t + q;
                leaf_total += c;
            }
            __syncthreads();
        }
        if (stats) stats[0] += leaf_total;
        for (uint32_t base = 0; base < leaf_total; base += 64) {
            const uint32_t k = base + lane;
            bool pass = false;
            uint32_t tuid = 0;
            if (k < leaf_total) {
                tuid = sh.tri_buf[k];
                bool want = true;
                if (edges_only) {   // ... and test only the triangles that carry one
                    const tri_meta_t m = sc.tri_meta[tuid];
                    want = m.edge[0] != kInvalid || m.edge[1] != kInvalid || m.edge[2] != kInvalid;
                }
                if (want) {
                    const tri_geo_t tri = sc.tri_geo[tuid];
                    pass = cone_tri_maybe(tcone, tri.a, tri.b, tri.c, slab);
                }
            }
            const unsigned long long pm = __ballot(pass);
            if (pm) {
                const uint32_t pos = nsurv + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
                if (pass && pos < kCoopSurvCap) sh.surv[pos] = tuid;
                nsurv += (uint32_t)__popcll(pm);
            }
            if (nsurv >= 64u) flush();
        }
        leaf_total = 0;
        if (nsurv > 0 && s == 0) flush();
        __syncthreads();
        if (s == 0) break;
    }
    return out;
}

// The bitmap edge set left in sh.edge_bits by coop_gather(do_edges): number of ids / the first `cap` ids in ascending order -> dst.
WT_D uint32_t coop_edge_count(const scene_t& sc, coop_edges_t& sh) {
    __syncthreads();
    uint32_t c = 0;
    for (uint32_t j = threadIdx.x & 63; j < (sc.n_edges + 31u) / 32u; j += 64) c += (uint32_t)__popc(sh.edge_bits[j]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off, 64);
    return c;
}
WT_D void coop_edge_write(const scene_t& sc, coop_edges_t& sh, uint32_t* dst, uint32_t cap) {
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    const uint32_t nw = (sc.n_edges + 31u) / 32u;
    for (uint32_t j0 = 0; j0 < nw && base < cap; j0 += 64) {
        const uint32_t j = j0 + (uint32_t)lane;
        uint32_t bits = j < nw ? sh.edge_bits[j] : 0u;
        uint32_t c = (uint32_t)__popc(bits), pre = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)pre, off, 64);
            if (lane >= off) pre += o;
        }
        uint32_t pos = base + pre - c;
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1;
            if (pos < cap) dst[pos] = j * 32u + (uint32_t)b;
            ++pos;
        }
        base += (uint32_t)__shfl((int)pre, 63, 64);
    }
}

// Cuts the part of the tree that overlaps (cone ∩ slab) into subtrees of at most `max_tris` triangles and hands each to emit(ptr)
// (ptr: child reference as in bvh8_node_t::child).  One wavefront; emit is called by ONE lane per subtree, possibly several lanes at
// once.  Used to spread the region sums of interaction regions with 10^3..10^5 triangles over many wavefronts (k_flux_split).
template <class SH, class Emit>
WT_D void coop_split(const scene_t& sc, const cone_t& tcone, const range_t& slab, SH& sh, uint32_t max_tris, Emit&& emit) {
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, sub = lane & 7;
    if (sc.n_nodes == 0) return;
    const vec3 ro = tcone.o, rd = tcone.d;
    const vec3 rinvd{1.f / rd.x, 1.f / rd.y, 1.f / rd.z};
    const bool sx = __builtin_signbit(rinvd.x), sy = __builtin_signbit(rinvd.y), sz = __builtin_signbit(rinvd.z);
    const float ta = tcone.tan_alpha, ix = tcone.x0;
    int s = 1;
    if (lane == 0) sh.stack[0] = stack_entry_t{0.f, 1};
    __syncthreads();
    if (sc.nodes[0].tris_count <= max_tris) {
        if (lane == 0) emit(1);
        return;
    }
    while (s > 0) {
        const int np = s < 8 ? s : 8;
        stack_entry_t e{0.f, 0};
        if (grp < np) e = sh.stack[s - 1 - grp];
        s -= np;
        __syncthreads();
        bool push = false;
        int32_t cp = 0;
        float tmin = 0.f;
        if (grp < np) {   // every stack entry is an internal node with more than max_tris triangles
            const bvh8_node_t& node = sc.nodes[e.ptr - 1];
            cp = node.child[sub];
            if (cp != 0 && cone_child_test(node, sub, ro, rd, rinvd, sx, sy, sz, ta, ix, slab, tmin)) {
                if (cp < 0 || sc.nodes[cp - 1].tris_count <= max_tris)
                    emit(cp);
                else
                    push = true;
            }
        }
        const unsigned long long hm = __ballot(push);
        if (hm) {
            const int pos = s + __popcll(hm & ((1ull << lane) - 1ull));
            if (push && pos < kCoopStack) sh.stack[pos] = stack_entry_t{tmin, cp};
            else if (push) emit(cp);   // stack full (cannot happen at these depths): the subtree goes out as one task
            const int total = s + __popcll(hm);
            s = total < kCoopStack ? total : kCoopStack;
        }
        __syncthreads();
    }
}

// One wavefront, one closest-hit ray query (bvh_traverse_ray<false> + ads_intersect_ray), same scheme as coop_cone_query:
// 8 stack entries x 8 children per step, buffered leaves tested 64 triangles per step.  A serial per-lane traversal is a
// chain of ~30 dependent loads (~1 us each at this occupancy); this one is ~10 steps.  Equal-distance ties (a ray through
// a shared edge) are resolved towards the lowest buffered triangle instead of the first visited one.
WT_D bool coop_ray_query(const scene_t& sc, vec3 ro, vec3 rd, const range_t& range, coop_shared_t& sh, ray_hit_t& rec) {
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, sub = lane & 7;
    rec.dist = WT_INF;
    rec.tuid = kInvalid;
    rec.bx = rec.by = 0.f;
    rec.front_face = 0;
    if (sc.n_nodes == 0) return false;
    const vec3 rinvd{1.f / rd.x, 1.f / rd.y, 1.f / rd.z};
    const bool sx = __builtin_signbit(rinvd.x), sy = __builtin_signbit(rinvd.y), sz = __builtin_signbit(rinvd.z);
    int s = 1;
    uint32_t leaf_total = 0;
    if (lane == 0) sh.stack[0] = stack_entry_t{0.f, 1};
    __syncthreads();
    for (;;) {
        while (s > 0 && leaf_total < 64u) {
            const int np = s < 8 ? s : 8;
            stack_entry_t e{0.f, 0};
            if (grp < np) e = sh.stack[s - 1 - grp];
            s -= np;
            __syncthreads();
            const bool live = grp < np && e.t < rec.dist;
            bool leafish = false, h = false;
            uint32_t t0 = 0, cnt = 0;
            float tmin = 0.f;
            int32_t cp = 0;
            if (live) {
                if (e.ptr < 0) {
                    const bvh8_leaf_t leaf = bvh_leaf_of(e.ptr);
                    t0 = leaf.tris_ptr;
                    cnt = leaf.count;
                    leafish = true;
                } else {
                    const bvh8_node_t& n = sc.nodes[e.ptr - 1];
                    const uint32_t ntc = n.tris_count, nts = n.tris_start;
                    cp = n.child[sub];
                    if (ntc <= kCoopLeafTris) {
                        t0 = nts;
                        cnt = ntc;
                        leafish = true;
                    } else {
                        {
                            const int i = sub;
                            const float tfar = fminf_(rec.dist, range.max);
                            const float bminx = sx ? n.maxx[i] : n.minx[i], bmaxx = sx ? n.minx[i] : n.maxx[i];
                            const float bminy = sy ? n.maxy[i] : n.miny[i], bmaxy = sy ? n.miny[i] : n.maxy[i];
                            const float bminz = sz ? n.maxz[i] : n.minz[i], bmaxz = sz ? n.minz[i] : n.maxz[i];
                            const float t1x = (bminx - ro.x) * rinvd.x, t2x = (bmaxx - ro.x) * rinvd.x;
                            const float t1y = (bminy - ro.y) * rinvd.y, t2y = (bmaxy - ro.y) * rinvd.y;
                            const float t1z = (bminz - ro.z) * rinvd.z, t2z = (bmaxz - ro.z) * rinvd.z;
                            const float rmin = fmaxf_(fmaxf_(t1x, t1y), fmaxf_(t1z, range.min));
                            const float rmax = fminf_(fminf_(t2x, t2y), fminf_(t2z, tfar));
                            h = rmin <= rmax && cp != 0;
                            tmin = rmin;
                        }
                    }
                }
            }
            const unsigned long long hm = __ballot(h);
            if (hm) {
                const uint32_t gbits = (uint32_t)(hm >> (grp * 8)) & 0xffu;
                int rank = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float tj = __shfl(tmin, (lane & ~7) + j, 64);
                    if (((gbits >> j) & 1u) && (tj > tmin || (tj == tmin && j < sub))) ++rank;
                }
                const int above = grp < 7 ? __popcll(hm >> ((grp + 1) * 8)) : 0;
                const int pos = s + above + rank;
                if (h && pos < kCoopStack) sh.stack[pos] = stack_entry_t{tmin, cp};
                else if (h) atomicAdd(sh.dropped, 1ull);   // reported: wtgpu_counters::traversal_stack_dropped
                const int total = s + __popcll(hm);
                s = total < kCoopStack ? total : kCoopStack;
            }
            const unsigned long long lm = __ballot(leafish && sub == 0 && cnt > 0);
            unsigned long long m = lm;
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                uint32_t c = (uint32_t)__shfl((int)cnt, src, 64);
                const uint32_t t = (uint32_t)__shfl((int)t0, src, 64);
                if (c > kCoopLeafTris) c = kCoopLeafTris;   // cannot happen with this builder (leaves hold <= MAX_LEAF triangles)
                for (uint32_t q = (uint32_t)lane; q < c; q += 64u) sh.tri_buf[leaf_total + q] = t + q;
                leaf_total += c;
            }
            __syncthreads();
        }
        if (leaf_total == 0) {
            if (s == 0) break;
            continue;
        }
        for (uint32_t base = 0; base < leaf_total; base += 64) {
            const uint32_t k = base + lane;
            bool hit = false;
            float d = WT_INF;
            uint32_t tuid = 0;
            ray_tri_hit_t ht{WT_INF, 0.f, 0.f};
            bool ff = false;
            if (k < leaf_total) {
                tuid = sh.tri_buf[k];
                const tri_geo_t tri = sc.tri_geo[tuid];
                if (intersect_ray_tri_wide(ro, rd, tri.a, tri.b, tri.c, range, ht) && ht.dist < rec.dist) {
                    hit = true;
                    d = ht.dist;
                    ff = dot(tri.n, rd) <= 0.f;
                }
            }
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                const float dm = wave_min(d);
                const unsigned long long m2 = __ballot(hit && d == dm);
                const int src = __ffsll((long long)m2) - 1;
                rec.dist = dm;
                rec.tuid = (uint32_t)__shfl((int)tuid, src, 64);
                rec.bx = __shfl(ht.bx, src, 64);
                rec.by = __shfl(ht.by, src, 64);
                rec.front_face = (uint32_t)__shfl((int)ff, src, 64);
            }
        }
        leaf_total = 0;
        __syncthreads();
        if (s == 0) break;
    }
    if (!finitef(rec.dist) || rec.dist > range.max) {
        rec.dist = WT_INF;
        return false;
    }
    return true;
}

// integrator::traverse (traversal.hpp:94-172), wave-uniform — the device form of wt::traverse_axis (bvh.h): the closest hit of the beam
// axis (`axis`, from the per-lane kernel's hand-over; computed here when nullptr) stands in for the per-segment ray queries, bounds
// the cone queries and names the triangle under the axis of an overflowed region.  resume: continue with the cone query of segment
// seg0 at distance dist0 (everything before is settled).
WT_D trav_result_t coop_traverse(const scene_t& sc, const cone_t& envelope, float lambda_m, float distance, bool force_ray_tracing,
                                              coop_shared_t& sh, const uint_list_t& tris, unsigned long long* prof = nullptr, bool resume = false,
                                              uint32_t seg0 = 0, float dist0 = 0.f, uint32_t nray0 = 0, uint32_t ncone0 = 0, const ray_hit_t* axis = nullptr,
                                              bool primary_always = false, bool probe_resumed = true, uint32_t short_tuid = kInvalid,
                                              uint32_t origin_tuid = kInvalid, bool use_cache = true) {
#ifdef WTGPU_COOP_PROF
#define WT_COOP_PROF(i, t0_)
#else
#define WT_COOP_PROF(i, t0_) \
    if (prof) prof[i] += (unsigned long long)(clock64() - (t0_));
#endif
    trav_result_t r;
    r.aborted = 0;
    r.origin = envelope.o;
    r.empty = 1;
    r.ballistic = 1;
    r.dist = -WT_INF;
    r.region_depth = 0.f;
    r.front_face = 0;
    r.tuid = kInvalid;
    r.bx = r.by = 0.f;
    r.pdist = 0.f;
    r.ntris = 0;
    r.overflow = 0;
    r.n_ray_queries = resume ? nray0 : 0u;
    r.n_cone_queries = resume ? ncone0 : 0u;
    const vec3 ro = envelope.o, rd = envelope.d;
    ray_hit_t ah;
    bool axis_hit;
    if (axis) {
        ah = *axis;
        axis_hit = ah.tuid != kInvalid;
    } else {
        r.n_ray_queries++;
        const long long tq0 = prof ? clock64() : 0;
        axis_hit = coop_ray_query(sc, ro, rd, range_t{0.f, distance}, sh, ah);
        WT_COOP_PROF(0, tq0)
    }
    auto ballistic_hit = [&]() __attribute__((always_inline)) {
        r.empty = 0;
        r.dist = ah.dist;
        r.tuid = ah.tuid;
        r.bx = ah.bx;
        r.by = ah.by;
        r.front_face = ah.front_face;
        r.ntris = 1;
    };
    if (force_ray_tracing || cone_is_ray(envelope)) {
        if (axis_hit) ballistic_hit();
        return r;
    }
    float dist = resume ? dist0 : 0.f;
    for (uint32_t seg = resume ? seg0 : 0u;; ++seg) {
        const float ballistic_dist = max_ballistic_distance(lambda_m, seg, 0.f);
        if (!(resume && seg == seg0)) {   // (that segment is settled already; `dist` is past it)
            if (axis_hit && ah.dist <= fminf_(distance, dist + ballistic_dist * kBallisticScale)) {
                ballistic_hit();
                return r;
            }
            dist += ballistic_dist;
            if (ballistic_dist == WT_INF || dist >= distance) return r;
        }
        const float min_df_prog = cone_axes(envelope, dist).x / 2.f;
        cone_hit_t ch;
        r.n_cone_queries++;
        // A thin-slab any-hit probe first: for wide beams it is far cheaper than letting the full (near-first, 8-wide) query find
        // a too-near hit, which expands the whole cone's top levels before its first triangle batch (measured: 1.6x slower).
        const long long tp0 = prof ? clock64() : 0;
        // (probe_resumed = false: not for the query a per-lane attempt handed over — that attempt spent its budget near-first without
        // meeting a too-near hit, so the full query, which also stops at the first too-near hit, rarely finds one)
        // ... and before the probe the two remembered triangles (see wt::traverse_axis): the one that satisfied the previous probe — or made
        // the per-lane attempt before the hand-over too short — and the one the beam started from.  Same test, same slab as the probe's.
        const range_t thin{dist, fminf_(distance, dist + min_df_prog)};
        bool near_hit = false;
        for (int c = 0; c < 2 && !near_hit; ++c) {
            const uint32_t cand = c == 0 ? short_tuid : origin_tuid;
            if (!use_cache || cand == kInvalid || (c == 1 && cand == short_tuid)) continue;
            const tri_geo_t tri = sc.tri_geo[cand];
            cone_tri_hit_t ht;
            near_hit = intersect_cone_tri<true>(envelope, tri.a, tri.b, tri.c, tri.n, thin, ht) && !(ht.dist > thin.max);
        }
        if (!near_hit) {
            uint32_t hit_tuid = kInvalid;
            near_hit = (probe_resumed || !(resume && seg == seg0)) && coop_cone_any(sc, envelope, thin, sh, prof, &hit_tuid);
            if (near_hit) short_tuid = hit_tuid;
        }
        WT_COOP_PROF(1, tp0)
        if (near_hit) continue;   // too short (see bvh_cone_any_hit)
        const long long tc0 = prof ? clock64() : 0;
        const float cone_max = axis_hit ? fminf_(distance, cone_axis_bound(envelope, ah.dist)) : distance;
        coop_cone(sc, envelope, range_t{dist, cone_max}, kMajorAxisToZScale, sh, tris, ch, prof, min_df_prog);
        WT_COOP_PROF(2, tc0)
        if (ch.too_short) {   // (boundary case of the probe's inclusive slab)
            short_tuid = ch.short_tuid;
            continue;
        }
        const bool df_empty = ch.ntris == 0 && ch.overflow == 0;
        if (df_empty || ch.dist - dist >= min_df_prog) {
            r.ballistic = 0;
            r.empty = df_empty;
            r.dist = df_empty ? -WT_INF : ch.dist;
            r.front_face = ch.front_face;
            r.ntris = ch.ntris;
            r.overflow = ch.overflow;
            r.region_depth = df_empty ? 0.f : kMajorAxisToZScale * cone_axes(envelope, ch.dist).x;
            if (!df_empty && (primary_always || ch.overflow > 0)) primary_from_axis(sc, envelope, axis_hit, ah, r);
            return r;
        }
    }
}

}   // namespace wt
#endif
