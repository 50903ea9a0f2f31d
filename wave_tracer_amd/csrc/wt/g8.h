// wave_tracer_amd — 8-lane-group BVH traversal (device only): ONE QUERY PER 8 LANES, eight queries per wavefront.
//
// Why.  A lane that walks the 8-wide BVH on its own fetches every node by value (256 B = 16 dwordx4 loads per lane, 1024
// scattered cache-line requests per wavefront and node step), holds the node in 64 VGPRs, keeps a private stack that spills to
// scratch, and idles while the slowest of its 63 neighbours finishes.  Here the 8 lanes of a group serve one query:
//   * lane i of the group loads and tests child i of the popped node (28 B per lane, 8 x 32-B contiguous runs per group): a node
//     step costs a wavefront 8 x 256 B of coalesced traffic instead of 64 x 256 B of scattered traffic;
//   * the child hits are pushed far-first on a group-owned LDS stack with ranks computed by 8 cross-lane compares — the
//     reference's stable insertion sort (src/ads/bvh8w.cpp:45-57), so the visiting order is that of the sequential traversal;
//   * leaf triangles are tested one per lane (leaves hold <= 4, ray shortcut subtrees <= 16 triangles, bvh8w.cpp:29,512-526);
//     the sequential loop updates its search range only after a whole leaf (bvh8w.cpp:134-179), so testing a leaf's triangles
//     side by side against the same range is the same computation: closest hit, tie-breaks and (cone queries) the order of the
//     triangle list are IDENTICAL to the per-lane / CPU traversal (wt/bvh.h), unlike the 64-wide variant in coop.h.
// All 8 lanes of a group must call with identical arguments; groups of a wavefront run independent queries (divergent control
// flow between groups is fine: every cross-lane operation stays inside a group).
#pragma once
#if defined(__HIPCC__)
#include "coop.h"

namespace wt {

constexpr int kG8Stack = 64;   // entries per group (LDS, contiguous): the per-lane traversal's stack size (bvh8w.cpp: 64)

struct g8_stack_t {
    stack_entry_t* p;   // this group's kG8Stack entries
};

__device__ inline int g8_sub() { return (int)(threadIdx.x & 7u); }
__device__ inline uint32_t g8_bits(unsigned long long m) { return (uint32_t)(m >> (threadIdx.x & 56u)) & 0xffu; }   // my group's 8 ballot bits
template <class T>
__device__ inline T g8_bcast(T v, int sub) {
    return __shfl(v, sub, 8);
}
__device__ inline float g8_min(float v) {
    v = fminf(v, __shfl_xor(v, 1, 8));
    v = fminf(v, __shfl_xor(v, 2, 8));
    v = fminf(v, __shfl_xor(v, 4, 8));
    return v;
}
// LDS accesses of one wavefront execute in program order; this only keeps the compiler from moving them across
__device__ inline void g8_fence() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }

// far-first push of the group's child hits (stable for equal tmin: the lower child index stays deeper), returns the new stack size
__device__ inline int g8_push_sorted(const g8_stack_t& st, int s, bool h, float tmin, int32_t cp, bool& overflow) {
    const uint32_t gb = g8_bits(__ballot(h));
    if (gb) {
        const int sub = g8_sub();
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float tj = g8_bcast(tmin, j);
            if (((gb >> j) & 1u) && (tj > tmin || (tj == tmin && j < sub))) ++rank;
        }
        const int pos = s + rank;
        if (h && pos < kG8Stack) st.p[pos] = stack_entry_t{tmin, cp};
        const int total = s + __popc(gb);
        if (total > kG8Stack) overflow = true;
        s = total < kG8Stack ? total : kG8Stack;
        g8_fence();
    }
    return s;
}

// Closest-hit (shadow = false) or any-hit (shadow = true) ray query: bvh_traverse_ray (wt/bvh.h; src/ads/bvh8w.cpp:469-554).
template <bool shadow>
__device__ inline bool g8_ray_query(const scene_t& sc, vec3 ro, vec3 rd, const range_t& range, const g8_stack_t& st, ray_hit_t& rec) {
    const int sub = g8_sub();
    rec.dist = WT_INF;
    rec.tuid = kInvalid;
    rec.bx = rec.by = 0.f;
    rec.front_face = 0;
    if (sc.n_nodes == 0) return false;
    const vec3 rinvd{1.f / rd.x, 1.f / rd.y, 1.f / rd.z};
    const bool sx = __builtin_signbit(rinvd.x), sy = __builtin_signbit(rinvd.y), sz = __builtin_signbit(rinvd.z);
    int s = 1;
    if (sub == 0) st.p[0] = stack_entry_t{0.f, 1};
    g8_fence();
    bool ovf = false;
    while (s > 0) {
        const stack_entry_t top = st.p[s - 1];   // same address for the 8 lanes: LDS broadcast
        --s;
        g8_fence();
        uint32_t t0 = 0, cnt = 0;
        if (top.ptr < 0) {
            const bvh8_leaf_t leaf = sc.leaves[-top.ptr - 1];
            t0 = leaf.tris_ptr;
            cnt = leaf.count;
        } else {
            const bvh8_node_t& n = sc.nodes[top.ptr - 1];
            const uint32_t ntc = n.tris_count;
            if ((int)ntc <= kRayLeafShortcut) {
                t0 = n.tris_start;
                cnt = ntc;
            } else {
                const int32_t cp = n.child[sub];
                const float tfar = fminf_(rec.dist, range.max);
                const float bminx = sx ? n.maxx[sub] : n.minx[sub], bmaxx = sx ? n.minx[sub] : n.maxx[sub];
                const float bminy = sy ? n.maxy[sub] : n.miny[sub], bmaxy = sy ? n.miny[sub] : n.maxy[sub];
                const float bminz = sz ? n.maxz[sub] : n.minz[sub], bmaxz = sz ? n.minz[sub] : n.maxz[sub];
                const float t1x = (bminx - ro.x) * rinvd.x, t2x = (bmaxx - ro.x) * rinvd.x;
                const float t1y = (bminy - ro.y) * rinvd.y, t2y = (bmaxy - ro.y) * rinvd.y;
                const float t1z = (bminz - ro.z) * rinvd.z, t2z = (bmaxz - ro.z) * rinvd.z;
                const float rmin = fmaxf_(fmaxf_(t1x, t1y), fmaxf_(t1z, range.min));
                const float rmax = fminf_(fminf_(t2x, t2y), fminf_(t2z, tfar));
                s = g8_push_sorted(st, s, cp != 0 && rmin <= rmax, rmin, cp, ovf);
                continue;
            }
        }
        // ---- triangles t0 .. t0+cnt, one per lane
        bool any = false;
        for (uint32_t b = 0; b < cnt; b += 8) {
            const uint32_t t = b + (uint32_t)sub;
            bool hit = false;
            ray_tri_hit_t h{WT_INF, 0.f, 0.f};
            bool ff = false;
            if (t < cnt) {
                const tri_geo_t tri = sc.tri_geo[t0 + t];
                if (shadow) {
                    hit = test_ray_tri_wide(ro, rd, tri.a, tri.b, tri.c, range);
                } else {
                    hit = intersect_ray_tri_wide(ro, rd, tri.a, tri.b, tri.c, range, h) && h.dist < rec.dist;
                    ff = dot(tri.n, rd) <= 0.f;
                }
            }
            const uint32_t hb = g8_bits(__ballot(hit));
            if (!hb) continue;
            if (shadow) {
                rec.dist = range.min;
                return true;
            }
            const float dm = g8_min(hit ? h.dist : WT_INF);
            const uint32_t mb = g8_bits(__ballot(hit && h.dist == dm));
            const int src = __ffs((int)mb) - 1;   // ties: the lowest triangle index, like the sequential loop's strict '<'
            rec.dist = dm;
            rec.tuid = t0 + b + (uint32_t)src;
            rec.bx = g8_bcast(h.bx, src);
            rec.by = g8_bcast(h.by, src);
            rec.front_face = (uint32_t)g8_bcast((int)ff, src);
            any = true;
        }
        if (any)
            while (s > 0 && st.p[s - 1].t >= rec.dist) --s;
    }
    (void)ovf;
    return rec.dist < WT_INF;
}

__device__ inline bool g8_intersect_ray(const scene_t& sc, vec3 ro, vec3 rd, const range_t& range, const g8_stack_t& st, ray_hit_t& hit) {
    g8_ray_query<false>(sc, ro, rd, range, st, hit);
    if (!finitef(hit.dist) || hit.dist > range.max) {
        hit.dist = WT_INF;
        return false;
    }
    return true;
}
__device__ inline bool g8_shadow_ray(const scene_t& sc, vec3 ro, vec3 rd, const range_t& range, const g8_stack_t& st) {
    ray_hit_t h;
    g8_ray_query<true>(sc, ro, rd, range, st, h);
    return h.dist < WT_INF;
}

// Cone query (bvh_traverse_cone, wt/bvh.h; src/ads/bvh8w.cpp:232-318): closest distance + every triangle hit inside the shrinking
// z-slab, in the sequential traversal's order.  `budget` in units of 1 per triangle test / kNodeBudgetCost per node, as in bvh.h.
__device__ inline bool g8_cone_query(const scene_t& sc, const cone_t& cone, const range_t& searchrange, float z_scale, const g8_stack_t& st,
                                     const uint_list_t& tris, cone_hit_t& rec, uint32_t budget = 0xFFFFFFFFu, float min_progress = -WT_INF) {
    const int sub = g8_sub();
    rec.dist = WT_INF;
    rec.front_face = 0;
    rec.ntris = 0;
    rec.overflow = 0;
    rec.aborted = 0;
    rec.too_short = 0;
    uint32_t tests = 0;
    if (sc.n_nodes == 0) return false;
    const vec3 ro = cone.o, rd = cone.d;
    const vec3 rinvd{1.f / rd.x, 1.f / rd.y, 1.f / rd.z};
    const bool sx = __builtin_signbit(rinvd.x), sy = __builtin_signbit(rinvd.y), sz = __builtin_signbit(rinvd.z);
    const float ta = cone.tan_alpha, ix = cone.x0;
    range_t range = cone_search_range(cone, searchrange, rec.dist, z_scale);
    int s = 1;
    if (sub == 0) st.p[0] = stack_entry_t{0.f, 1};
    g8_fence();
    while (s > 0) {
        const stack_entry_t top = st.p[s - 1];
        --s;
        g8_fence();
        if (top.ptr < 0) {
            const bvh8_leaf_t leaf = sc.leaves[-top.ptr - 1];
            tests += leaf.count;
            if (tests > budget) {
                rec.aborted = 1;
                return false;
            }
            bool found = false;
            for (uint32_t b = 0; b < leaf.count; b += 8) {
                const uint32_t t = b + (uint32_t)sub;
                bool hit = false, ff = false;
                float d = WT_INF;
                if (t < leaf.count) {
                    const tri_geo_t tri = sc.tri_geo[leaf.tris_ptr + t];
                    ff = dot(tri.n, -rd) > 0.f;
                    cone_tri_hit_t h;
                    if (intersect_cone_tri(cone, tri.a, tri.b, tri.c, tri.n, range, h) && !(h.dist > range.max)) {   // numerics (bvh8w.cpp:162)
                        hit = true;
                        d = h.dist;
                    }
                }
                const uint32_t hb = g8_bits(__ballot(hit));
                if (!hb) continue;
                found = true;
                const float dm = g8_min(d);
                if (dm < rec.dist) {
                    const uint32_t mb = g8_bits(__ballot(hit && d == dm));
                    rec.dist = dm;
                    rec.front_face = (uint32_t)g8_bcast((int)ff, __ffs((int)mb) - 1);
                }
                const uint32_t pos = rec.ntris + (uint32_t)__popc(hb & ((1u << sub) - 1u));
                if (hit && pos < tris.cap) tris[pos] = leaf.tris_ptr + t;
                const uint32_t total = rec.ntris + (uint32_t)__popc(hb);
                const uint32_t newn = total < tris.cap ? total : tris.cap;
                rec.overflow += total - newn;
                rec.ntris = newn;
            }
            if (found) {
                if (rec.dist - searchrange.min < min_progress) {   // traversal.hpp:146,157: decided, see bvh_traverse_cone
                    rec.too_short = 1;
                    return true;
                }
                range = cone_search_range(cone, searchrange, rec.dist, z_scale);
                if (rec.overflow > 0) range.max = fminf_(range.max, rec.dist);   // bounded-list regime, see bvh_traverse_cone
                while (s > 0 && st.p[s - 1].t >= range.max) --s;
            }
            continue;
        }
        const bvh8_node_t& n = sc.nodes[top.ptr - 1];
        tests += kNodeBudgetCost;
        if (tests > budget) {
            rec.aborted = 1;
            return false;
        }
        const int32_t cp = n.child[sub];
        float tmin = 0.f;
        const bool hc = cone_child_test(n, sub, ro, rd, rinvd, sx, sy, sz, ta, ix, range, tmin);
        bool ovf = false;
        s = g8_push_sorted(st, s, hc && cp != 0, tmin, cp, ovf);
        if (ovf && budget != 0xFFFFFFFFu) {
            rec.aborted = 1;   // the 64-entry stack is full: the wave-cooperative query (512 entries) takes over
            return false;
        }
    }
    return rec.ntris > 0;
}

// integrator::traverse (traversal.hpp:94-172) for one group — the policy loop of wt::traverse (bvh.h) around the group queries.
// Extra (device only): when an accepted diffusive hit overflowed the bounded triangle list, the triangle under the beam axis
// (find_closest_triangle, plt_bdpt_detail.hpp:362-389: the closest axis hit among the region's triangles) is resolved right
// here with one ray query over the region's z-slab, so that the interaction kernels need no BVH stack: r.aborted = 2 marks
// "primary in r.tuid / r.bx / r.by / r.pdist (kInvalid: the axis misses the region)".
// resume: an earlier kernel settled every query before the cone query of segment seg0 (rays missed, diffusive attempts were too short;
// wt::traverse hand-over state): continue with that cone query at distance dist0.
__device__ inline trav_result_t g8_traverse(const scene_t& sc, const cone_t& envelope, float lambda_m, float distance, bool force_ray_tracing,
                                            const g8_stack_t& st, const uint_list_t& tris, uint32_t cone_budget, bool resume = false, uint32_t seg0 = 0,
                                            float dist0 = 0.f, uint32_t nray0 = 0, uint32_t ncone0 = 0) {
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
    r.n_ray_queries = r.n_cone_queries = 0;
    const vec3 ro = envelope.o, rd = envelope.d;
    ray_hit_t rh;
    if (force_ray_tracing || cone_is_ray(envelope)) {
        r.n_ray_queries++;
        if (g8_intersect_ray(sc, ro, rd, range_t{0.f, distance}, st, rh)) {
            r.empty = 0;
            r.dist = rh.dist;
            r.tuid = rh.tuid;
            r.bx = rh.bx;
            r.by = rh.by;
            r.front_face = rh.front_face;
            r.ntris = 1;
        }
        return r;
    }
    float dist = resume ? dist0 : 0.f;
    if (resume) {
        r.n_ray_queries = nray0;
        r.n_cone_queries = ncone0;
    }
    for (uint32_t seg = resume ? seg0 : 0u;; ++seg) {
        const float ballistic_dist = max_ballistic_distance(lambda_m, seg, 0.f);
        if (!(resume && seg == seg0)) {   // that segment's ray query missed already; `dist` is past it
            r.n_ray_queries++;
            if (g8_intersect_ray(sc, ro, rd, range_t{dist, fminf_(distance, dist + ballistic_dist * kBallisticScale)}, st, rh)) {
                r.empty = 0;
                r.dist = rh.dist;
                r.tuid = rh.tuid;
                r.bx = rh.bx;
                r.by = rh.by;
                r.front_face = rh.front_face;
                r.ntris = 1;
                return r;
            }
            dist += ballistic_dist;
            if (ballistic_dist == WT_INF || dist >= distance) return r;
        }
        const float min_df_prog = cone_axes(envelope, dist).x / 2.f;
        cone_hit_t ch;
        r.n_cone_queries++;
        g8_cone_query(sc, envelope, range_t{dist, distance}, kMajorAxisToZScale, st, tris, ch, cone_budget, min_df_prog);
        if (ch.aborted) {   // hand-over state for the cooperative kernel, as in wt::traverse
            r.aborted = 1;
            r.dist = dist;
            r.ntris = seg;
            r.n_cone_queries--;
            return r;
        }
        if (ch.too_short) continue;
        const bool df_empty = ch.ntris == 0 && ch.overflow == 0;
        if (df_empty || ch.dist - dist >= min_df_prog) {
            r.ballistic = 0;
            r.empty = df_empty;
            r.dist = df_empty ? -WT_INF : ch.dist;
            r.front_face = ch.front_face;
            r.ntris = ch.ntris;
            r.overflow = ch.overflow;
            r.region_depth = df_empty ? 0.f : kMajorAxisToZScale * cone_axes(envelope, ch.dist).x;
            return r;
        }
    }
}

// the primary triangle of an overflowed interaction region (see g8_traverse); `r` is an accepted diffusive result
__device__ inline void g8_resolve_primary(const scene_t& sc, const cone_t& envelope, const g8_stack_t& st, trav_result_t& r) {
    const range_t izr{r.dist, r.dist + r.region_depth};
    const float wtol = cone_intersection_tolerance(envelope.o, sc.world_min, sc.world_max, sc.world_max);
    r.aborted = 2;
    r.tuid = kInvalid;
    ray_hit_t rh;
    if (g8_intersect_ray(sc, envelope.o, envelope.d, grow(izr, wtol), st, rh)) {
        const tri_geo_t g = sc.tri_geo[rh.tuid];
        const float fptol = cone_intersection_tolerance(envelope.o, g.a, g.b, g.c);
        if (contains(grow(izr, fptol), rh.dist)) {
            r.tuid = rh.tuid;
            r.bx = rh.bx;
            r.by = rh.by;
            r.pdist = rh.dist;
        }
    }
}

}   // namespace wt
#endif
