// wave_tracer_amd — 8-lane-group BVH traversal (device only): ONE QUERY PER 8 LANES, eight queries per wavefront.
//
// Why.  A lane that walks the 8-wide BVH on its own fetches every node by value (256 B = 16 dwordx4 loads per lane, 1024
// scattered cache-line requests per wavefront and node step), holds the node in 64 VGPRs, keeps a private stack that spills to
// scratch, and idles while the slowest of its 63 neighbours finishes.  Here the 8 lanes of a group serve one query:
//   * lane i of the group loads and tests child i of the popped node (28 B per lane, 8 x 32-B contiguous runs per group): a node
//     step costs a wavefront 8 x 256 B of coalesced traffic instead of 64 x 256 B of scattered traffic;
//   * the child hits are pushed far-first on a group-owned LDS stack with ranks computed by 8 cross-lane compares — the
//     reference's stable insertion sort (src/ads/bvh8w.cpp:45-57), so the visiting order is that of the sequential traversal;
//   * leaf triangles are tested one per lane (leaves hold <= 4, ray shortcut subtrees <= 16 triangles, bvh8w.cpp:29,512-526):
//     closest hit and tie-breaks are IDENTICAL to the per-lane / CPU traversal (wt/bvh.h).
// Measured: 1.2x the per-lane kernel on plain ray queries; the same scheme for cone queries (node test per lane, leaf
// triangles 8 at a time) was 3x SLOWER than lane-per-walk in the pipeline (groups of a wavefront diverge and serialise) and is not kept.
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
            const bvh8_leaf_t leaf = bvh_leaf_of(top.ptr);
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

}   // namespace wt
#endif
