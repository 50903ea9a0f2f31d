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

namespace wt {

constexpr int kCoopStack = 192;
constexpr uint32_t kCoopLeafTris = 64;

struct coop_shared_t {
    stack_entry_t stack[kCoopStack];
};

__device__ inline float wave_min(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}

// cone x AABB of child i (bvh8w.cpp:187-230); returns hit + tmin
__device__ inline bool cone_child_test(const bvh8_node_t& n, int i, vec3 ro, vec3 rd, vec3 rinvd, bool sx, bool sy, bool sz, float ta, float ix,
                                       const range_t& range, float& tmin_out) {
    float ominx = n.minx[i] - ro.x, ominy = n.miny[i] - ro.y, ominz = n.minz[i] - ro.z;
    float omaxx = n.maxx[i] - ro.x, omaxy = n.maxy[i] - ro.y, omaxz = n.maxz[i] - ro.z;
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
    return tmin <= tmax && tmax >= range.min && tmin <= range.max && !(tmin >= range.max);
}

// One wavefront, one cone query.  Must be called by all 64 lanes of a 64-thread block with identical arguments.
__device__ inline void coop_cone(const scene_t& sc, const cone_t& cone, const range_t& searchrange, float z_scale, coop_shared_t& sh,
                                 const uint_list_t& tris, cone_hit_t& rec) {
    const int lane = threadIdx.x & 63;
    rec.dist = WT_INF;
    rec.front_face = 0;
    rec.ntris = 0;
    rec.overflow = 0;
    rec.aborted = 0;
    if (sc.n_nodes == 0) return;
    const vec3 ro = cone.o, rd = cone.d;
    const vec3 rinvd{1.f / rd.x, 1.f / rd.y, 1.f / rd.z};
    const bool sx = __builtin_signbit(rinvd.x), sy = __builtin_signbit(rinvd.y), sz = __builtin_signbit(rinvd.z);
    const float ta = cone.tan_alpha, ix = cone.x0;
    range_t range = cone_search_range(cone, searchrange, rec.dist, z_scale);
    int s = 1;
    if (lane == 0) sh.stack[0] = stack_entry_t{0.f, 1};
    __syncthreads();
    while (s > 0) {
        const stack_entry_t top = sh.stack[s - 1];
        --s;
        __syncthreads();   // everyone has read the top before it may be overwritten
        if (top.t >= range.max) continue;
        uint32_t t0 = 0, cnt = 0;
        bool brute = false;
        const bvh8_node_t* node = nullptr;
        if (top.ptr < 0) {
            const bvh8_leaf_t leaf = sc.leaves[-top.ptr - 1];
            t0 = leaf.tris_ptr;
            cnt = leaf.count;
            brute = true;
        } else {
            node = &sc.nodes[top.ptr - 1];
            if (node->tris_count <= kCoopLeafTris) {
                t0 = node->tris_start;
                cnt = node->tris_count;
                brute = true;
            }
        }
        if (brute) {
            for (uint32_t base = 0; base < cnt; base += 64) {
                bool hit = false;
                float d = WT_INF;
                bool ff = false;
                const uint32_t ti = base + lane;
                if (ti < cnt) {
                    const tri_geo_t tri = sc.tri_geo[t0 + ti];
                    ff = dot(tri.n, -rd) > 0.f;
                    cone_tri_hit_t h;
                    if (intersect_cone_tri(cone, tri.a, tri.b, tri.c, tri.n, range, h) && !(h.dist > range.max)) {
                        hit = true;
                        d = h.dist;
                    }
                }
                const unsigned long long mask = __ballot(hit);
                if (mask) {
                    const float dm = wave_min(d);
                    const unsigned long long m2 = __ballot(hit && d == dm);
                    const int src = __ffsll((long long)m2) - 1;
                    const int ffmin = __shfl((int)ff, src, 64);
                    if (dm < rec.dist) {
                        rec.dist = dm;
                        rec.front_face = (uint32_t)ffmin;
                    }
                    const uint32_t pos = rec.ntris + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                    if (hit && pos < tris.cap) tris[pos] = t0 + ti;
                    const uint32_t total = rec.ntris + (uint32_t)__popcll(mask);
                    const uint32_t newn = total < tris.cap ? total : tris.cap;
                    rec.overflow += total - newn;
                    rec.ntris = newn;
                    range = cone_search_range(cone, searchrange, rec.dist, z_scale);
                    if (rec.overflow > 0) range.max = fminf_(range.max, rec.dist);   // bounded-list regime, see bvh.h
                }
            }
        } else {
            bool h = false;
            float tmin = 0.f;
            int32_t cp = 0;
            if (lane < 8) {
                cp = node->child[lane];
                if (cp != 0) h = cone_child_test(*node, lane, ro, rd, rinvd, sx, sy, sz, ta, ix, range, tmin);
            }
            const unsigned mask = (unsigned)(__ballot(h) & 0xffull);
            const int n = __popc(mask);
            // rank for a far-first (descending tmin) stable order
            int rank = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float tj = __shfl(tmin, j, 64);
                const bool hj = (mask >> j) & 1u;
                if (hj && (tj > tmin || (tj == tmin && j < lane))) ++rank;
            }
            if (h && s + rank < kCoopStack) sh.stack[s + rank] = stack_entry_t{tmin, cp};
            s = (s + n < kCoopStack) ? s + n : kCoopStack;
        }
        __syncthreads();
    }
}

// Wave-cooperative any-hit probe (see bvh_cone_any_hit).
__device__ inline bool coop_cone_any(const scene_t& sc, const cone_t& cone, const range_t& range, coop_shared_t& sh) {
    const int lane = threadIdx.x & 63;
    if (sc.n_nodes == 0) return false;
    const vec3 ro = cone.o, rd = cone.d;
    const vec3 rinvd{1.f / rd.x, 1.f / rd.y, 1.f / rd.z};
    const bool sx = __builtin_signbit(rinvd.x), sy = __builtin_signbit(rinvd.y), sz = __builtin_signbit(rinvd.z);
    const float ta = cone.tan_alpha, ix = cone.x0;
    int s = 1;
    if (lane == 0) sh.stack[0] = stack_entry_t{0.f, 1};
    __syncthreads();
    bool found = false;
    while (s > 0 && !found) {
        const stack_entry_t top = sh.stack[s - 1];
        --s;
        __syncthreads();
        uint32_t t0 = 0, cnt = 0;
        bool brute = false;
        const bvh8_node_t* node = nullptr;
        if (top.ptr < 0) {
            const bvh8_leaf_t leaf = sc.leaves[-top.ptr - 1];
            t0 = leaf.tris_ptr;
            cnt = leaf.count;
            brute = true;
        } else {
            node = &sc.nodes[top.ptr - 1];
            if (node->tris_count <= kCoopLeafTris) {
                t0 = node->tris_start;
                cnt = node->tris_count;
                brute = true;
            }
        }
        if (brute) {
            bool hit = false;
            if ((uint32_t)lane < cnt) {
                const tri_geo_t tri = sc.tri_geo[t0 + lane];
                cone_tri_hit_t h;
                hit = intersect_cone_tri(cone, tri.a, tri.b, tri.c, tri.n, range, h) && !(h.dist > range.max);
            }
            if (__ballot(hit)) found = true;
        } else {
            bool h = false;
            float tmin = 0.f;
            int32_t cp = 0;
            if (lane < 8) {
                cp = node->child[lane];
                if (cp != 0) h = cone_child_test(*node, lane, ro, rd, rinvd, sx, sy, sz, ta, ix, range, tmin);
            }
            const unsigned mask = (unsigned)(__ballot(h) & 0xffull);
            const int n = __popc(mask);
            int rank = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float tj = __shfl(tmin, j, 64);
                const bool hj = (mask >> j) & 1u;
                if (hj && (tj > tmin || (tj == tmin && j < lane))) ++rank;
            }
            if (h && s + rank < kCoopStack) sh.stack[s + rank] = stack_entry_t{tmin, cp};
            s = (s + n < kCoopStack) ? s + n : kCoopStack;
        }
        __syncthreads();
    }
    return found;
}

// integrator::traverse (traversal.hpp:94-172), wave-uniform.  `stack` is a per-lane stack for the (redundant) ray queries.
__device__ inline trav_result_t coop_traverse(const scene_t& sc, const cone_t& envelope, float lambda_m, float distance, bool force_ray_tracing,
                                              const stack_ref_t& stack, coop_shared_t& sh, const uint_list_t& tris) {
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
    r.ntris = 0;
    r.overflow = 0;
    r.n_ray_queries = r.n_cone_queries = 0;
    const vec3 ro = envelope.o, rd = envelope.d;
    ray_hit_t rh;
    if (force_ray_tracing || cone_is_ray(envelope)) {
        r.n_ray_queries++;
        if (ads_intersect_ray(sc, ro, rd, range_t{0.f, distance}, stack, rh)) {
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
    float dist = 0.f;
    for (uint32_t seg = 0;; ++seg) {
        const float ballistic_dist = max_ballistic_distance(lambda_m, seg, 0.f);
        r.n_ray_queries++;
        if (ads_intersect_ray(sc, ro, rd, range_t{dist, fminf_(distance, dist + ballistic_dist * kBallisticScale)}, stack, rh)) {
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
        const float min_df_prog = cone_axes(envelope, dist).x / 2.f;
        cone_hit_t ch;
        r.n_cone_queries++;
        if (coop_cone_any(sc, envelope, range_t{dist, fminf_(distance, dist + min_df_prog)}, sh)) continue;   // too short (see bvh_cone_any_hit)
        coop_cone(sc, envelope, range_t{dist, distance}, kMajorAxisToZScale, sh, tris, ch);
        const bool df_empty = ch.ntris == 0;
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

}   // namespace wt
#endif
