// wave_tracer_amd — 8-wide BVH ray / cone traversal and the ballistic<->diffusive traversal policy
// (SURVEY.md §8 rows a6, a7).
//
// Reference: src/ads/bvh8w.cpp:45-57 (child sort), 123-185 (cone leaf test), 187-230 (cone x 8 AABB),
//            232-318 (cone traversal), 394-452 (ray leaf test), 454-467 (ray x 8 AABB),
//            469-554 (ray traversal), 556-603 (closest / any hit);
//            include/wt/ads/traversal_common.hpp:62-88 (search_range), 116-149 (record assembly);
//            include/wt/integrator/traversal.hpp:26-57, 94-172, 319-333.
//
// The traversal stack is addressed through a (pointer, stride) pair so that a HIP kernel can keep it
// in LDS, interleaved across the lanes of a wavefront (entry i of lane l at lds[i*blockDim + l]: every
// lane owns its own bank column, ds_read_b64 / ds_write_b64 are conflict-free), while the CPU checker
// passes a plain local array with stride 1.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "cone.h"
#include "scene.h"

namespace wt {

struct alignas(8) stack_entry_t {   // (moved as one 64-bit word: stack_ref_t)
    float t;
    int32_t ptr;
};
// Two-segment stack: entries [0,n_fast) live at p[i*stride] (LDS, lane-interleaved, on the device), entries
// [n_fast,cap) in a private spill array q (scratch).  The CPU checker uses n_fast = cap, q = nullptr.
//
// An entry travels as ONE 64-bit scalar and, in device code, the two segment pointers carry their address spaces in their TYPES
// (LDS = 3, scratch = 5).  Both matter: with generic pointers — or with struct-typed loads, which bind a generic reference — the compiler
// folds "load from p or from q" into a FLAT access through a selected pointer (and sinks half of an entry's store behind the merge as a
// flat store): every push and pop then went through the texture path and waited on vmcnt together with the node fetches (round 3's
// kernel: 10 flat loads, 21 flat stores, no ds_read at all).  Typed like this the fast segment is ds_read_b64 / ds_write_b64.
#if defined(__HIP_DEVICE_COMPILE__)
#define WT_AS_LDS __attribute__((address_space(3)))
#define WT_AS_PRIVATE __attribute__((address_space(5)))
#else
#define WT_AS_LDS
#define WT_AS_PRIVATE
#endif
typedef WT_AS_LDS unsigned long long* stack_fast_ptr_t;
typedef WT_AS_PRIVATE unsigned long long* stack_spill_ptr_t;
WT_HD unsigned long long stack_pack(stack_entry_t e) {
    uint32_t tb;
    __builtin_memcpy(&tb, &e.t, 4);
    return ((unsigned long long)(uint32_t)e.ptr << 32) | tb;
}
WT_HD stack_entry_t stack_unpack(unsigned long long v) {
    const uint32_t tb = (uint32_t)v;
    stack_entry_t e;
    __builtin_memcpy(&e.t, &tb, 4);
    e.ptr = (int32_t)(uint32_t)(v >> 32);
    return e;
}
struct stack_ref_t {
    stack_fast_ptr_t p;
    uint32_t stride;
    uint32_t cap;
    uint32_t n_fast;
    stack_spill_ptr_t q;
    WT_HD stack_entry_t get(int i) const {
        unsigned long long v;
        if ((uint32_t)i < n_fast)
            v = p[(size_t)i * stride];
        else
            v = q[(uint32_t)i - n_fast];
        return stack_unpack(v);
    }
    WT_HD void set(int i, stack_entry_t e) const {
        const unsigned long long v = stack_pack(e);
        if ((uint32_t)i < n_fast)
            p[(size_t)i * stride] = v;
        else
            q[(uint32_t)i - n_fast] = v;
    }
    WT_HD float get_t(int i) const { return get(i).t; }
};
// (stack storage is declared as stack_entry_t arrays: 8-byte entries, 8-byte aligned)
static_assert(sizeof(stack_entry_t) == 8 && alignof(stack_entry_t) == 8, "stack entries are moved as 64-bit words");
WT_HD stack_ref_t make_stack_ref(stack_entry_t* fast, uint32_t stride, uint32_t cap, uint32_t n_fast, stack_entry_t* spill) {
    stack_ref_t s;
    s.p = (stack_fast_ptr_t)fast;
    s.stride = stride;
    s.cap = cap;
    s.n_fast = n_fast;
    s.q = (stack_spill_ptr_t)spill;
    return s;
}
WT_HD bvh8_leaf_t bvh_leaf_of(int32_t child) {   // child < 0 (wt/scene.h: bvh8_leaf_t)
    const uint32_t v = (uint32_t)(-child);
    return bvh8_leaf_t{v >> 3, v & 7u};
}
WT_HD stack_ref_t make_flat_stack(stack_entry_t* p, uint32_t cap) { return make_stack_ref(p, 1, cap, cap, nullptr); }

struct uint_list_t {   // bounded output list with the same (pointer,stride) addressing
    uint32_t* p;
    uint32_t stride;
    uint32_t cap;
    // optional: the cone-hit distance of every listed triangle (same addressing).  With it a cone query ends by dropping the
    // triangles beyond its FINAL slab — `if (wt.dist > range.max) continue;`, include/wt/ads/traversal_common.hpp:131-135.  (In the
    // reference that filter never fires: src/ads/bvh8w.cpp:175 records the triangle without its distance, so `wt.dist` is 0 and the
    // record keeps whatever the traversal met while its slab was still wider — a set that depends on the visiting order, i.e. on
    // the BVH builder.  d == nullptr reproduces that; everything here passes d: the record as traversal_common.hpp states it, which is
    // also what the device's whole-region walks compute.)
    float* d = nullptr;
    WT_HD uint32_t& operator[](uint32_t i) const { return p[(size_t)i * stride]; }
};

// ---- node sources of the traversals -----------------------------------------------------------------------------------------------
// The traversal steps below read a node through a NODE SOURCE: the reference-layout node (bvh8_node_t, 256 B, exact float boxes: the CPU checker,
// the wave-cooperative kernels, which read a node with 64 lanes) or — the device's per-lane traversals — a 128-BYTE node whose child boxes are
// 16-bit coordinates on ONE grid over the scene's bounding box, rounded OUTWARDS (bvh8_qnode_t, built at upload from the same tree: same indices, same
// child references).  Why: a lane fetches a whole node by itself, so 256 B are two cache lines per lane and 64 registers in flight; the 5.7 MB of
// nodes of the headline scene do not fit the 4 MB L2 of an XCD, 2.8 MB do.  What it may change: nothing but the amount of work.  A decoded box
// CONTAINS the exact one, so a query visits a superset of the children the exact boxes admit; every triangle test is the exact one; a closest
// hit, an any-hit answer and the set of triangles inside a query's final slab do not depend on which boxes were opened.  One grid
// for the whole scene instead of one per node (the round-1 experiment: 8 bits relative to the node, many false positives): a cell is 1 / 65533 of
// the scene's extent — 0.1 mm in a 6 m room whose leaf boxes measure millimetres — and a node needs no origin / scale of its own: 96 B of boxes +
// 32 B of child references, no change of the tree's layout.  (The coordinate is decoded with ONE fused multiply-add, which rounds once, on the
// host exactly as on the device: the builder checks every decoded bound against the float it must enclose and steps outwards until it does.)
struct alignas(16) bvh8_qchild_t {   // 16 B: one 128-bit load per child
    uint16_t lo[3];   // box minimum (x, y, z), rounded down by at least one cell
    uint16_t hi[3];   // maximum, rounded up by at least one cell; an empty child: lo = 65535, hi = 0
    int32_t child;    // as bvh8_node_t::child
};
struct alignas(16) bvh8_qnode_t {
    bvh8_qchild_t c[8];
};
static_assert(sizeof(bvh8_qnode_t) == 128, "one cache line per node");
struct qgrid_t {
    vec3 origin, cell;   // coordinate = fma(q, cell, origin)
};
WT_HD float qgrid_decode(float origin, float cell, uint32_t q) { return fmaf((float)q, cell, origin); }
// the grid over [mn, mx] (two cells of slack on either side; a flat axis gets cells of its own)
WT_HD qgrid_t qgrid_make(vec3 mn, vec3 mx) {
    qgrid_t g;
    const float ex = mx.x - mn.x, ey = mx.y - mn.y, ez = mx.z - mn.z;
    const float fx = fmaxf_(fabsf(mn.x), fabsf(mx.x)) * 1e-6f + 1e-30f, fy = fmaxf_(fabsf(mn.y), fabsf(mx.y)) * 1e-6f + 1e-30f,
                fz = fmaxf_(fabsf(mn.z), fabsf(mx.z)) * 1e-6f + 1e-30f;
    g.cell = vec3{fmaxf_(ex / 65530.f, fx), fmaxf_(ey / 65530.f, fy), fmaxf_(ez / 65530.f, fz)};
    g.origin = vec3{mn.x - 2.f * g.cell.x, mn.y - 2.f * g.cell.y, mn.z - 2.f * g.cell.z};
    return g;
}
// host: the 16-bit coordinate ONE CELL BEYOND the largest whose decoded value is <= v (down) / the smallest whose decoded value is >= v (up): the
// decoded box encloses the exact one with at least a cell to spare on every side, which is what lets a traversal fold the decoding into its own
// arithmetic (t = fma(q, cell / d, (origin - o) / d) rounds differently from ((fma(q, cell, origin) - o) / d) by a few ulp, a cell is 2^-16 of the scene)
inline uint16_t qgrid_encode(float origin, float cell, float v, bool up) {
    const double t = ((double)v - (double)origin) / (double)cell;
    long q = (long)(up ? std::ceil(t) : std::floor(t));
    if (q < 0) q = 0;
    if (q > 65535) q = 65535;
    if (up) {
        while (q < 65535 && qgrid_decode(origin, cell, (uint32_t)q) < v) ++q;
        while (q > 0 && qgrid_decode(origin, cell, (uint32_t)(q - 1)) >= v) --q;
        if (q < 65535) ++q;
    } else {
        while (q > 0 && qgrid_decode(origin, cell, (uint32_t)q) > v) --q;
        while (q < 65535 && qgrid_decode(origin, cell, (uint32_t)(q + 1)) <= v) ++q;
        if (q > 0) --q;
    }
    return (uint16_t)q;
}
// FALSE: the grid cannot enclose this node (a box outside the grid's range) — such a scene is refused at upload
inline bool qnode_make(const bvh8_node_t& n, const qgrid_t& g, bvh8_qnode_t& out) {
    bool ok = true;
    const float* mn[3] = {n.minx, n.miny, n.minz};
    const float* mx[3] = {n.maxx, n.maxy, n.maxz};
    const float o[3] = {g.origin.x, g.origin.y, g.origin.z}, c[3] = {g.cell.x, g.cell.y, g.cell.z};
    for (int i = 0; i < 8; ++i) {
        out.c[i].child = n.child[i];
        for (int ax = 0; ax < 3; ++ax) {
            if (n.child[i] == 0) {
                out.c[i].lo[ax] = 65535;
                out.c[i].hi[ax] = 0;
                continue;
            }
            out.c[i].lo[ax] = qgrid_encode(o[ax], c[ax], mn[ax][i], false);
            out.c[i].hi[ax] = qgrid_encode(o[ax], c[ax], mx[ax][i], true);
            // (strictly beyond: a cell to spare)
            ok = ok && qgrid_decode(o[ax], c[ax], out.c[i].lo[ax]) < mn[ax][i] && qgrid_decode(o[ax], c[ax], out.c[i].hi[ax]) > mx[ax][i];
        }
    }
    return ok;
}
struct node_box_t {
    float x0, y0, z0, x1, y1, z1;
};
struct wide_nodes_t {   // the reference-layout nodes
    typedef bvh8_node_t node_t;
    const bvh8_node_t* p;
    WT_HD node_t fetch(int32_t idx) const { return p[idx]; }
    WT_HD int32_t child(const node_t& n, int i) const { return n.child[i]; }
    WT_HD node_box_t box(const node_t& n, int i) const { return node_box_t{n.minx[i], n.miny[i], n.minz[i], n.maxx[i], n.maxy[i], n.maxz[i]}; }
    // the box of child i relative to a query's origin `o`; rel(o) is computed once per step
    WT_HD vec3 rel(vec3 o) const { return o; }
    WT_HD node_box_t box_rel(const node_t& n, int i, vec3 r) const {
        return node_box_t{n.minx[i] - r.x, n.miny[i] - r.y, n.minz[i] - r.z, n.maxx[i] - r.x, n.maxy[i] - r.y, n.maxz[i] - r.z};
    }
};
struct grid_nodes_t {   // the 128-byte nodes on the scene grid
    typedef bvh8_qnode_t node_t;
    const bvh8_qnode_t* p;
    qgrid_t g;
    WT_HD node_t fetch(int32_t idx) const { return p[idx]; }
    WT_HD int32_t child(const node_t& n, int i) const { return n.c[i].child; }
    WT_HD node_box_t box(const node_t& n, int i) const {
        const bvh8_qchild_t& c = n.c[i];
        return node_box_t{qgrid_decode(g.origin.x, g.cell.x, c.lo[0]), qgrid_decode(g.origin.y, g.cell.y, c.lo[1]), qgrid_decode(g.origin.z, g.cell.z, c.lo[2]),
                          qgrid_decode(g.origin.x, g.cell.x, c.hi[0]), qgrid_decode(g.origin.y, g.cell.y, c.hi[1]), qgrid_decode(g.origin.z, g.cell.z, c.hi[2])};
    }
    // ... relative to a query's origin: the decoding folded into the subtraction (one fused multiply-add per coordinate; qgrid_encode's spare cell
    // covers the different rounding)
    WT_HD vec3 rel(vec3 o) const { return vec3{g.origin.x - o.x, g.origin.y - o.y, g.origin.z - o.z}; }
    WT_HD node_box_t box_rel(const node_t& n, int i, vec3 r) const {
        const bvh8_qchild_t& c = n.c[i];
        return node_box_t{qgrid_decode(r.x, g.cell.x, c.lo[0]), qgrid_decode(r.y, g.cell.y, c.lo[1]), qgrid_decode(r.z, g.cell.z, c.lo[2]),
                          qgrid_decode(r.x, g.cell.x, c.hi[0]), qgrid_decode(r.y, g.cell.y, c.hi[1]), qgrid_decode(r.z, g.cell.z, c.hi[2])};
    }
};
// What a traversal that is not handed a node source reads: on the device the 128-byte nodes — stored BEHIND the scene's nodes in the same allocation,
// followed by the grid (wtgpu.hip: upload_impl; a scene whose boxes the grid cannot enclose gets cell.x = 0 there, and its kernels the exact
// nodes: lane_nodes_usable) —, on the host the exact ones.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WT_NO_QNODES)
typedef grid_nodes_t lane_nodes_t;
WT_HD lane_nodes_t lane_nodes(const scene_t& sc) {
    const bvh8_qnode_t* q = reinterpret_cast<const bvh8_qnode_t*>(sc.nodes + sc.n_nodes);
    const float* g = reinterpret_cast<const float*>(q + sc.n_nodes);
    return lane_nodes_t{q, qgrid_t{vec3{g[0], g[1], g[2]}, vec3{g[3], g[4], g[5]}}};
}
#else
typedef wide_nodes_t lane_nodes_t;
WT_HD lane_nodes_t lane_nodes(const scene_t& sc) { return lane_nodes_t{sc.nodes}; }
#endif

#if defined(WTGPU_FSD_WATCH) && defined(__HIPCC__)
// (bring-up aid, tools/watch_path.py: a host-mapped buffer the kernels of a hanging launch write their progress into — device printf never
// flushes from a kernel that does not end)
__device__ volatile unsigned int* g_watch = nullptr;
#endif
#if defined(WTGPU_FSD_WATCH) && defined(__HIP_DEVICE_COMPILE__)
#define WT_WATCH(slot, value) do { if (g_watch) g_watch[(blockIdx.x & 63u) * 16u + (slot)] = (unsigned int)(value); } while (0)
#define WT_WATCH_ADD(slot) do { if (g_watch) atomicAdd((unsigned int*)g_watch + (blockIdx.x & 63u) * 16u + (slot), 1u); } while (0)
#else
#define WT_WATCH(slot, value) ((void)0)
#define WT_WATCH_ADD(slot) ((void)0)
#endif

// The children of a node that passed their box test go on the stack far-first (descending tmin; equal tmin: in child order) — the result
// of the reference's insertion sort of the freshly pushed entries (bvh8w.cpp:45-57).  Here every entry is written ONCE, at its final
// position: rank = number of accepted children that sort before it, from 28 register compares, no loop over memory and no data-dependent
// trip count (on the device the insertion sort was a per-lane loop of stack loads and stores whose length differed in every lane).
// t[i] / c[i]: tmin and child reference of child i, ok bit i: accepted.  Returns the new stack size.
WT_HD int stack_push_sorted(const stack_ref_t& s, int begin, const float t[8], const int32_t c[8], uint32_t ok) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j == i) continue;
            const bool before = j < i ? t[j] >= t[i] : t[j] > t[i];   // (j sorts before i: farther, or as far and earlier in child order)
            rank += (((ok >> j) & 1u) && before) ? 1 : 0;
        }
        if ((ok >> i) & 1u) {
            s.set(begin + rank, stack_entry_t{t[i], c[i]});
            ++n;
        }
    }
    return begin + n;
}

constexpr uint32_t kNodeBudgetCost = 2;   // budget units charged per node visit of a cone query (1 unit = 1 triangle test)

struct bvh_counters_t {
    uint32_t nodes, leaves, tri_tests;                  // ray queries
    uint32_t cone_nodes, cone_leaves, cone_tri_tests;   // full cone queries
    uint32_t probe_nodes, probe_tri_tests;              // any-hit cone probes
};

// ---- ray ---------------------------------------------------------------------------------------
struct ray_hit_t {
    float dist;   // +inf: none
    uint32_t tuid;
    float bx, by;
    uint32_t front_face;
};

// The triangles of a leaf can be FETCHED IN BATCHES before the first of them is tested (WT_LEAF_BATCH): a loop that loads triangle t inside
// iteration t is a chain of dependent memory round trips (the loads cannot be hoisted over the early exits), and the device's traversal kernels
// are bound by such round trips, not by arithmetic.  Measured in round 4 (leaves hold <= 4 triangles): batches of 2 / 4 cost 24 / 48 more
// registers where the kernels already spill — 22.3 / 21.3 against 22.4 Msamples/s with the plain loop (batch 1, the default; 22.6 with batches
// of 2 and 256 registers per lane, i.e. two wavefronts per SIMD instead of three).  The tests run in the reference's order either way.
#ifndef WT_LEAF_BATCH
#define WT_LEAF_BATCH 1
#endif
constexpr uint32_t kLeafBatch = WT_LEAF_BATCH;
template <bool shadow>
WT_HD bool ray_gather_tris(const scene_t& sc, vec3 ro, vec3 rd, uint32_t t0, uint32_t count, const range_t& range, ray_hit_t& rec,
                           bvh_counters_t* ctr) {
    bool intersects = false;
    for (uint32_t b = 0; b < count; b += kLeafBatch) {
        tri_geo_t T[kLeafBatch];
        const uint32_t m = count - b < kLeafBatch ? count - b : kLeafBatch;
#pragma unroll
        for (uint32_t i = 0; i < kLeafBatch; ++i)
            if (i < m) T[i] = sc.tri_geo[t0 + b + i];
#pragma unroll
        for (uint32_t i = 0; i < kLeafBatch; ++i) {
            if (i >= m) break;
            const tri_geo_t& tri = T[i];
            if (ctr) ctr->tri_tests++;
            if (shadow) {
                if (test_ray_tri_wide(ro, rd, tri.a, tri.b, tri.c, range)) {
                    rec.dist = range.min;
                    return true;
                }
                continue;
            }
            ray_tri_hit_t h;
            if (intersect_ray_tri_wide(ro, rd, tri.a, tri.b, tri.c, range, h) && h.dist < rec.dist) {
                rec.dist = h.dist;
                rec.bx = h.bx;
                rec.by = h.by;
                rec.tuid = t0 + b + i;
                rec.front_face = dot(tri.n, rd) <= 0.f;
                intersects = true;
            }
        }
    }
    return intersects;
}

// Closest-hit (shadow=false) or any-hit (shadow=true) ray traversal (bvh8w.cpp:469-554).
// Box culling uses [range.min, min(range.max, closest)] instead of the reference's [0, closest]; the set of
// accepted triangles is identical because the triangle test itself enforces `range`.
//
// Written as resumable steps over an explicit state (ray_query_t) like the cone query further down: rq_begin, then rq_node_step while
// the query holds no leaf and rq_leaf_step when it does, until rq_running() is false.  bvh_traverse_ray is the sequential driver;
// the device's trace kernel runs the axis queries of its lanes through the same steps, interleaved with their cone queries.
struct ray_query_t {
    range_t range;
    ray_hit_t rec;
    int s;                  // stack entries
    uint32_t lt0, lcnt;     // triangles to test next (lcnt = 0: none)
};
WT_HD bool rq_running(const ray_query_t& q) { return q.s > 0 || q.lcnt != 0; }
WT_HD void rq_begin(const scene_t& sc, const range_t& range, const stack_ref_t& stack, ray_query_t& q) {
    q.range = range;
    q.rec.dist = WT_INF;
    q.rec.tuid = kInvalid;
    q.rec.bx = q.rec.by = 0.f;
    q.rec.front_face = 0;
    q.lt0 = q.lcnt = 0;
    q.s = 0;
    if (sc.n_nodes == 0) return;
    q.s = 1;
    stack.set(0, stack_entry_t{0.f, 1});
}
// the children of a fetched node `n`, tested and pushed far-first (the entry that named it has been popped: q.s is the stack size without it)
template <class NS>
WT_HD void rq_node_children(const NS& ns, const typename NS::node_t& n, vec3 ro, vec3 rd, const stack_ref_t& stack, ray_query_t& q, bvh_counters_t* ctr = nullptr) {
    const vec3 rinvd{1.f / rd.x, 1.f / rd.y, 1.f / rd.z};
    const bool sx = __builtin_signbit(rinvd.x), sy = __builtin_signbit(rinvd.y), sz = __builtin_signbit(rinvd.z);
    const int s = q.s;
    const vec3 rel = ns.rel(ro);
    if (ctr) ctr->nodes++;
    const float tfar = fminf_(q.rec.dist, q.range.max);
    // all eight slab tests without a branch, then one write per accepted child at its sorted position (stack_push_sorted)
    float tm[8];
    int32_t cps[8];
    uint32_t ok = 0;
    int room = (int)stack.cap - s;   // (a full stack drops the children that do not fit, in child order)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int32_t cp = ns.child(n, i);
        const node_box_t b = ns.box_rel(n, i, rel);
        const float bminx = sx ? b.x1 : b.x0, bmaxx = sx ? b.x0 : b.x1;
        const float bminy = sy ? b.y1 : b.y0, bmaxy = sy ? b.y0 : b.y1;
        const float bminz = sz ? b.z1 : b.z0, bmaxz = sz ? b.z0 : b.z1;
        const float t1x = bminx * rinvd.x, t2x = bmaxx * rinvd.x;
        const float t1y = bminy * rinvd.y, t2y = bmaxy * rinvd.y;
        const float t1z = bminz * rinvd.z, t2z = bmaxz * rinvd.z;
        const float rmin = fmaxf_(fmaxf_(t1x, t1y), fmaxf_(t1z, q.range.min));
        const float rmax = fminf_(fminf_(t2x, t2y), fminf_(t2z, tfar));
        const bool acc = cp != 0 && rmin <= rmax && room > 0;
        room -= acc ? 1 : 0;
        ok |= acc ? 1u << i : 0u;
        tm[i] = rmin;
        cps[i] = cp;
    }
    q.s = stack_push_sorted(stack, s, tm, cps, ok);
}
// pops one entry: a leaf is kept for rq_leaf_step, a node is fetched and its children tested (requires q.s > 0, q.lcnt == 0)
template <class NS>
WT_HD void rq_node_step(const NS& ns, vec3 ro, vec3 rd, const stack_ref_t& stack, ray_query_t& q, bvh_counters_t* ctr = nullptr) {
    int s = q.s;
    const stack_entry_t top = stack.get(s - 1);
    --s;
    q.s = s;
    if (top.ptr < 0) {
        const bvh8_leaf_t leaf = bvh_leaf_of(top.ptr);
        if (ctr) ctr->leaves++;
        q.lt0 = leaf.tris_ptr;
        q.lcnt = leaf.count;
        return;
    }
    // by value: the whole node is fetched with wide loads issued back to back (one memory latency per node instead of
    // one per child field); the unrolled child loop then runs from registers
    // (The reference tests the triangles of a subtree of <= 16 one after the other instead of descending, src/ads/bvh8w.cpp:29,512: a shortcut for
    // its 8-wide SIMD box test, not a different answer — the closest hit is the closest hit.  It is not taken here, by the checker or by the device:
    // lanes of a wavefront with 1 to 16 triangles each run the longest loop, and the 128-byte nodes do not carry a subtree's triangle count.)
    const typename NS::node_t n = ns.fetch(top.ptr - 1);
    rq_node_children(ns, n, ro, rd, stack, q, ctr);
}
// RAY queries always read the exact nodes.  A ray that lies IN the plane of an axis-aligned wall grazes that wall's flat box: (box - origin) / d is
// 0 x inf for that axis, the slab test fails, and the coplanar triangles are never tested — as in the reference, whose boxes are exact floats too.
// A box rounded outwards is entered, and the tolerant ray-triangle test then reports a hit at the wall's rim: on the city-block scene (etoile: rays
// leaving diffraction points along the walls) 0.1 % of such rays found an occluder the exact boxes do not show, 2.5e-3 of the film's energy
// (profiles/r06_ab_experiments.log).  Cone queries grow every box by the beam's radius first: no such degenerate case, and both node sources give
// the same answers (tests/test_oracle.py::test_grid_nodes_answer_like_exact_nodes).
WT_HD void rq_node_step(const scene_t& sc, vec3 ro, vec3 rd, const stack_ref_t& stack, ray_query_t& q, bvh_counters_t* ctr = nullptr) {
    rq_node_step(wide_nodes_t{sc.nodes}, ro, rd, stack, q, ctr);
}
// tests the held triangles (requires q.lcnt != 0); TRUE: an any-hit (shadow) query is decided
template <bool shadow>
WT_HD bool rq_leaf_step(const scene_t& sc, vec3 ro, vec3 rd, const stack_ref_t& stack, ray_query_t& q, bvh_counters_t* ctr = nullptr) {
    const uint32_t lt0 = q.lt0, lcnt = q.lcnt;
    q.lcnt = 0;
    const bool intr = ray_gather_tris<shadow>(sc, ro, rd, lt0, lcnt, q.range, q.rec, ctr);
    if (intr) {
        if (shadow) {
            q.s = 0;
            return true;
        }
        int s = q.s;
        while (s > 0 && stack.get_t(s - 1) >= q.rec.dist) --s;
        q.s = s;
    }
    return false;
}
template <bool shadow, class NS>
WT_HD bool bvh_traverse_ray_ns(const NS& ns, const scene_t& sc, vec3 ro, vec3 rd, const range_t& range, const stack_ref_t& stack, ray_hit_t& rec,
                               bvh_counters_t* ctr = nullptr) {
    // "while-while" form (Aila & Laine): every lane of a wavefront first descends through nodes until it holds a leaf (lanes that
    // found theirs wait at the inner loop's exit), then all of them test their leaf's triangles together — instead of paying the node
    // path AND the leaf path in every iteration because some lane needs each.  Per lane the order of visits is unchanged.
    ray_query_t q;
    rq_begin(sc, range, stack, q);
    while (rq_running(q)) {
        while (q.s > 0 && q.lcnt == 0) {
            WT_WATCH_ADD(8);
            WT_WATCH(9, q.s);
            rq_node_step(ns, ro, rd, stack, q, ctr);
        }
        WT_WATCH_ADD(10);
        WT_WATCH(11, q.lcnt);
        if (q.lcnt != 0 && rq_leaf_step<shadow>(sc, ro, rd, stack, q, ctr)) break;
    }
    rec = q.rec;
    return rec.dist < WT_INF;
}
template <bool shadow>
WT_HD bool bvh_traverse_ray(const scene_t& sc, vec3 ro, vec3 rd, const range_t& range, const stack_ref_t& stack, ray_hit_t& rec,
                            bvh_counters_t* ctr = nullptr) {
    return bvh_traverse_ray_ns<shadow>(wide_nodes_t{sc.nodes}, sc, ro, rd, range, stack, rec, ctr);   // (exact nodes: see rq_node_step)
}

// ads_t::intersect(ray) + ray_work_to_intersection_record (traversal_common.hpp:94-113)
WT_HD bool ads_intersect_ray(const scene_t& sc, vec3 ro, vec3 rd, const range_t& range, const stack_ref_t& stack, ray_hit_t& hit,
                             bvh_counters_t* ctr = nullptr) {
    bvh_traverse_ray<false>(sc, ro, rd, range, stack, hit, ctr);
    if (!finitef(hit.dist) || hit.dist > range.max) {
        hit.dist = WT_INF;
        return false;
    }
    return true;
}
// ads_t::shadow(ray): TRUE if occluded (bvh8w.cpp:579-603)
WT_HD bool ads_shadow_ray(const scene_t& sc, vec3 ro, vec3 rd, const range_t& range, const stack_ref_t& stack,
                          bvh_counters_t* ctr = nullptr) {
    ray_hit_t h;
    bvh_traverse_ray<true>(sc, ro, rd, range, stack, h, ctr);
    return h.dist < WT_INF;
}

// ---- cone --------------------------------------------------------------------------------------
// Extra conservative culling of a child box against (cone ∩ z-slab), not in the reference (result-preserving: a culled box
// cannot contain a point of the cone inside `range`).  The reference's test below grows the box by the cone radius at its far
// z and slab-tests the AXIS against it, which for wide beams lets through everything beside a thin slab; these two tests bound
// the box itself: (1) its extent along the axis must overlap the slab, (2) the lateral distance of its bounding sphere from
// the axis must not exceed the cone's radius at the largest admissible z (containment x^2+(e y)^2 <= r(z)^2 with e >= 1
// implies Euclidean lateral distance <= r(z) <= r(z_hi)).  b0/b1: box corners relative to the cone origin.
WT_HD bool cone_box_outside(float b0x, float b0y, float b0z, float b1x, float b1y, float b1z, vec3 rd, float ta, float ix, const range_t& range) {
#if WT_SS_ACTIVE
    return false;   // second source of the cone x box culls: none — every child is visited, the query's result cannot depend on a box test
#endif
    const float cx = 0.5f * (b0x + b1x), cy = 0.5f * (b0y + b1y), cz = 0.5f * (b0z + b1z);
    const float hx = 0.5f * (b1x - b0x), hy = 0.5f * (b1y - b0y), hz = 0.5f * (b1z - b0z);
    const float zc = cx * rd.x + cy * rd.y + cz * rd.z;
    const float he = fabsf(rd.x) * hx + fabsf(rd.y) * hy + fabsf(rd.z) * hz;
    const float slack = 1e-5f * (fabsf(zc) + he) + 1e-12f;
    if (zc - he > range.max + slack || zc + he < range.min - slack) return true;
#if defined(WT_CONE_BOX_NO_LATERAL)
    return false;   // (A/B: the axial test alone)
#endif
    const float rho2 = fmaxf_(0.f, cx * cx + cy * cy + cz * cz - zc * zc);
    const float rbox = cull_sqrtf(hx * hx + hy * hy + hz * hz);   // (the test below carries a 5e-4 margin)
    const float zhi = fminf_(range.max, zc + he);
    const float rcone = fmaxf_(0.f, fmaf(zhi, ta, ix));
    const float lim = (rcone + rbox) * 1.0005f + slack;
    return finitef(lim) && rho2 > lim * lim;
}


struct cone_hit_t {
    float dist;   // closest intersection distance (+inf: none)
    uint32_t front_face;
    uint32_t ntris;      // triangles written to the list
    uint32_t overflow;   // triangles dropped because the list was full
    uint32_t aborted;    // work budget exceeded
    uint32_t too_short;  // early exit: the closest hit is already known to lie within `min_progress` of the search start
    uint32_t short_tuid; // ... and the triangle that decided it (kInvalid otherwise): the next attempt tests it first (traverse_axis)
};

// intersection_record_work_t::search_range (traversal_common.hpp:76-83)
WT_HD range_t cone_search_range(const cone_t& cone, const range_t& searchrange, float intr_dist, float z_scale) {
    const float dist = fmaxf_(searchrange.min, intr_dist);
    const float z_dist = cone_axes(cone, dist).x * z_scale;
    range_t r{searchrange.min, fminf_(searchrange.max, dist + z_dist)};
    return rand_(r, range_positive());
}

// Cone traversal (bvh8w.cpp:232-318): closest distance + every triangle hit inside the (shrinking) z-slab.
// `budget`: maximum number of cone-triangle tests; when exceeded the query stops and rec.aborted is set (the device
// hands such heavy queries to the wavefront-cooperative traversal, wtgpu.hip: coop_cone).
//
// The query is written as RESUMABLE STEPS over an explicit state (cone_query_t): cq_begin, then cq_node_step while the query holds no
// leaf, cq_leaf_step when it does, until cq_running() is false, then cq_end.  bvh_traverse_cone below is the sequential driver (one
// query from start to end: the CPU checker, the plt_path kernels); the device's per-lane trace kernel (wtgpu.hip: k_trace) drives the
// same steps for 64 independent queries in lock step — all lanes that hold a node descend together, all lanes that hold a leaf test
// triangles together, and a lane whose query ends is handed its next query (or a new walk) while the others carry on.  One
// implementation, one order of visits per query, whoever drives it.
struct cone_query_t {
    range_t sr;           // search range of the query
    float z_scale;
    float min_progress;   // early exit: a closest hit nearer than this to sr.min ends the query ("too short")
    uint32_t budget;
    range_t range;        // current (shrinking) slab
    cone_hit_t rec;
    uint32_t tests;
    int s;                // stack entries
    int32_t leaf;         // child reference of the leaf to test next (0: none)
};
WT_HD bool cq_running(const cone_query_t& q) { return q.s > 0 || q.leaf != 0; }
WT_HD void cq_begin(const scene_t& sc, const cone_t& cone, const range_t& searchrange, float z_scale, const stack_ref_t& stack, uint32_t budget, float min_progress,
                    cone_query_t& q) {
    q.sr = searchrange;
    q.z_scale = z_scale;
    q.min_progress = min_progress;
    q.budget = budget;
    q.rec.dist = WT_INF;
    q.rec.front_face = 0;
    q.rec.ntris = 0;
    q.rec.overflow = 0;
    q.rec.aborted = 0;
    q.rec.too_short = 0;
    q.rec.short_tuid = kInvalid;
    q.tests = 0;
    q.leaf = 0;
    q.s = 0;
    q.range = cone_search_range(cone, searchrange, q.rec.dist, z_scale);
    if (sc.n_nodes == 0) return;
    q.s = 1;
    stack.set(0, stack_entry_t{0.f, 1});
}
// ends the query at once (budget exceeded / stack full / too short): nothing is left to visit
WT_HD void cq_stop(cone_query_t& q) {
    q.s = 0;
    q.leaf = 0;
}
// the children of a fetched node `n`: budget charge, box tests, sorted push (the entry that named it has been popped: q.s is the stack size without it)
template <class NS>
WT_HD void cq_node_children(const NS& ns, const typename NS::node_t& n, const cone_t& cone, const stack_ref_t& stack, cone_query_t& q, bvh_counters_t* ctr = nullptr) {
    const vec3 ro = cone.o, rd = cone.d;
    const vec3 rinvd{1.f / rd.x, 1.f / rd.y, 1.f / rd.z};
    const bool sx = __builtin_signbit(rinvd.x), sy = __builtin_signbit(rinvd.y), sz = __builtin_signbit(rinvd.z);
    const float ta = cone.tan_alpha, ix = cone.x0;
    const range_t range = q.range;
    const vec3 rel = ns.rel(ro);
    int s = q.s;
    if (ctr) ctr->cone_nodes++;
    q.tests += kNodeBudgetCost;   // an 8-wide node visit costs a lane about as much as a few triangle tests
    if (q.tests > q.budget) {
        q.rec.aborted = 1;
        cq_stop(q);
        return;
    }
    // all eight box tests without a branch, then one write per accepted child at its sorted position (stack_push_sorted)
    float tm[8];
    int32_t cps[8];
    uint32_t ok = 0;
    int room = (int)stack.cap - s;
    bool full = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int32_t cp = ns.child(n, i);
        // cone_cluster_intersect (bvh8w.cpp:187-230): grow the box by the cone radius at its far z
        const node_box_t nb = ns.box_rel(n, i, rel);
        float ominx = nb.x0, ominy = nb.y0, ominz = nb.z0;
        float omaxx = nb.x1, omaxy = nb.y1, omaxz = nb.z1;
        const float b0x = ominx, b0y = ominy, b0z = ominz, b1x = omaxx, b1y = omaxy, b1z = omaxz;
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
#if WT_SS_ACTIVE
        const bool hit = cp != 0;
        tmin = 0.f;
#else
        const bool hit = cp != 0 && tmin <= tmax && tmax >= range.min && tmin <= range.max && !(tmin >= range.max) &&
                         !cone_box_outside(b0x, b0y, b0z, b1x, b1y, b1z, rd, ta, ix, range);
#endif
        const bool acc = hit && room > 0;
        full = full || (hit && !acc);   // the stack cannot hold this child
        room -= acc ? 1 : 0;
        ok |= acc ? 1u << i : 0u;
        tm[i] = tmin;
        cps[i] = cp;
    }
    if (full) {
        if (q.budget != 0xFFFFFFFFu) {   // device: the 64-entry per-lane stack is full -> the wave-cooperative query (512 entries) takes over
            q.rec.aborted = 1;
            cq_stop(q);
            return;
        }
#if !defined(__HIP_DEVICE_COMPILE__)
        // CPU checker (unbudgeted): a dropped child would be a silently wrong answer
        fprintf(stderr, "wt::cq_node_step: traversal stack of %u entries is full\n", stack.cap);
        abort();
#endif
    }
    s = stack_push_sorted(stack, s, tm, cps, ok);
    q.s = s;
}
// pops one entry: a leaf is kept for cq_leaf_step, a node's children are tested and pushed far-first (requires q.s > 0, q.leaf == 0)
template <class NS>
WT_HD void cq_node_step(const NS& ns, const cone_t& cone, const stack_ref_t& stack, cone_query_t& q, bvh_counters_t* ctr = nullptr) {
    int s = q.s;
    const stack_entry_t top = stack.get(s - 1);
    --s;
    if (top.ptr < 0) {
        q.leaf = top.ptr;
        q.s = s;
        return;
    }
    // by value: the whole node is fetched with wide loads issued back to back (one memory latency per node instead of
    // one per child field); the unrolled child loop then runs from registers
    const typename NS::node_t n = ns.fetch(top.ptr - 1);
    q.s = s;
    cq_node_children(ns, n, cone, stack, q, ctr);
}
WT_HD void cq_node_step(const scene_t& sc, const cone_t& cone, const stack_ref_t& stack, cone_query_t& q, bvh_counters_t* ctr = nullptr) {
    cq_node_step(lane_nodes(sc), cone, stack, q, ctr);
}
// What follows from a hit of the running query at distance `dist` on triangle `tuid`: closest distance, list entry, early exit, slab
// shrink and stack pruning (the body of the reference's leaf loop, bvh8w.cpp:134-185).
WT_HD void cq_apply_hit(const cone_t& cone, const stack_ref_t& stack, const uint_list_t& tris, cone_query_t& q, uint32_t tuid, float dist, bool front_face) {
    cone_hit_t& rec = q.rec;
    const bool moved = dist < rec.dist;
    if (moved) {
        rec.dist = dist;
        rec.front_face = front_face;
    }
    if (rec.ntris < tris.cap) {
        if (tris.d) tris.d[(size_t)rec.ntris * tris.stride] = dist;
        tris[rec.ntris++] = tuid;
    } else
        rec.overflow++;
    // integrator::traverse discards a diffusive hit closer than `min_progress` to the start of the search
    // (traversal.hpp:146,157); the closest distance only ever decreases, so the outcome is decided right here
    if (rec.dist - q.sr.min < q.min_progress) {
        rec.too_short = 1;
        rec.short_tuid = moved ? tuid : rec.short_tuid;
        cq_stop(q);
        return;
    }
    if (moved) rec.short_tuid = tuid;   // (the triangle of the closest hit so far: names the rejecting triangle if the query ends too short)
    q.range = cone_search_range(cone, q.sr, rec.dist, q.z_scale);
    // Bounded-list regime (device only; the CPU checker's list never saturates): once the triangle list is full
    // nothing further can be recorded, so only triangles that can still lower the closest distance matter.
    if (rec.overflow > 0) q.range.max = fminf_(q.range.max, rec.dist);
    int s = q.s;
    while (s > 0 && stack.get_t(s - 1) >= q.range.max) --s;
    q.s = s;
}
// One exact cone-triangle test of the running query.
WT_HD void cq_exact_step_tri(const cone_t& cone, const stack_ref_t& stack, const uint_list_t& tris, cone_query_t& q, uint32_t tuid, const tri_geo_t& tri,
                             bvh_counters_t* ctr = nullptr) {
    if (ctr) ctr->cone_tri_tests++;
    cone_tri_hit_t h;
    if (!intersect_cone_tri(cone, tri.a, tri.b, tri.c, tri.n, q.range, h)) return;
    if (h.dist > q.range.max) return;   // numerics (bvh8w.cpp:162)
    cq_apply_hit(cone, stack, tris, q, tuid, h.dist, dot(tri.n, -cone.d) > 0.f);
}
WT_HD void cq_exact_step(const scene_t& sc, const cone_t& cone, const stack_ref_t& stack, const uint_list_t& tris, cone_query_t& q, uint32_t tuid,
                         bvh_counters_t* ctr = nullptr) {
    cq_exact_step_tri(cone, stack, tris, q, tuid, sc.tri_geo[tuid], ctr);
}
// Takes the held leaf (requires q.leaf != 0): charges its triangles to the budget and returns them as (first, count); count 0 when the
// budget is exceeded (the query is stopped).
WT_HD bvh8_leaf_t cq_take_leaf(cone_query_t& q, bvh_counters_t* ctr = nullptr) {
    const bvh8_leaf_t leaf = bvh_leaf_of(q.leaf);
    q.leaf = 0;
    if (ctr) ctr->cone_leaves++;
    q.tests += leaf.count;
    if (q.tests > q.budget) {
        q.rec.aborted = 1;
        cq_stop(q);
        return bvh8_leaf_t{0u, 0u};
    }
    return leaf;
}
// tests the triangles of the held leaf one after the other, fetched kLeafBatch at a time (see ray_gather_tris; requires q.leaf != 0)
// Device: the triangles' BOUNDING SPHERES first (16 B each, stored behind the scene's triangles at upload: wt/coop.h) — all of a leaf's at once, one
// memory round trip — and the 48-byte triangle only of those whose sphere meets the cone inside the current slab (cone_sphere_maybe: conservative,
// tests/test_gpu_traversal.py::test_bounding_sphere_filter_is_conservative; a slab that shrinks while the leaf is tested only rejects more).  The
// plain loop fetches triangle after triangle, each fetch behind the previous test: up to four dependent round trips per leaf visit in a kernel that
// waits for memory 60 % of its time, for triangles of which three in four fail the slab test at once.  Same hits, same order, same budget charge.
#ifndef WT_LEAF_SPHERES
#define WT_LEAF_SPHERES 0   // (A/B: tools/build_variant.sh lsph -DWT_LEAF_SPHERES=1)
#endif
#ifndef WT_LEAF_SPHERE_GROUP
#define WT_LEAF_SPHERE_GROUP 4   // spheres fetched together
#endif
WT_HD void cq_leaf_step(const scene_t& sc, const cone_t& cone, const stack_ref_t& stack, const uint_list_t& tris, cone_query_t& q,
                        bvh_counters_t* ctr = nullptr) {
    const bvh8_leaf_t leaf = cq_take_leaf(q, ctr);
#if defined(__HIP_DEVICE_COMPILE__) && WT_LEAF_SPHERES
    {
        const float4* sph = reinterpret_cast<const float4*>(sc.tri_geo + sc.n_tris) + leaf.tris_ptr;
        constexpr uint32_t G = WT_LEAF_SPHERE_GROUP;
        for (uint32_t b = 0; b < leaf.count; b += G) {
            const uint32_t m = leaf.count - b < G ? leaf.count - b : G;
            float4 bs[G];
#pragma unroll
            for (uint32_t i = 0; i < G; ++i)
                if (i < m) bs[i] = sph[b + i];
            uint32_t pass = 0;
#pragma unroll
            for (uint32_t i = 0; i < G; ++i)
                if (i < m && cone_sphere_maybe(cone, vec3{bs[i].x, bs[i].y, bs[i].z}, bs[i].w, q.range)) pass |= 1u << i;
            while (pass) {
                const uint32_t i = (uint32_t)__builtin_ctz(pass);
                pass &= pass - 1u;
                const uint32_t tuid = leaf.tris_ptr + b + i;
                cq_exact_step_tri(cone, stack, tris, q, tuid, sc.tri_geo[tuid], ctr);
                if (q.rec.too_short) return;
            }
        }
        return;
    }
#endif
    for (uint32_t b = 0; b < leaf.count; b += kLeafBatch) {
        tri_geo_t T[kLeafBatch];
        const uint32_t m = leaf.count - b < kLeafBatch ? leaf.count - b : kLeafBatch;
#pragma unroll
        for (uint32_t i = 0; i < kLeafBatch; ++i)
            if (i < m) T[i] = sc.tri_geo[leaf.tris_ptr + b + i];
#pragma unroll
        for (uint32_t i = 0; i < kLeafBatch; ++i) {
            if (i >= m) break;
            cq_exact_step_tri(cone, stack, tris, q, leaf.tris_ptr + b + i, T[i], ctr);
            if (q.rec.too_short) return;
        }
    }
}
// after the last step: the triangles beyond the final slab leave the list (cone_work_to_intersection_record)
WT_HD bool cq_end(const cone_t& cone, const uint_list_t& tris, cone_query_t& q) {
    cone_hit_t& rec = q.rec;
    if (rec.aborted) return false;
    if (rec.too_short) return true;
    if (tris.d && rec.ntris > 0) {
        const float zmax = cone_search_range(cone, q.sr, rec.dist, q.z_scale).max;
        uint32_t m = 0;
        for (uint32_t j = 0; j < rec.ntris; ++j) {
            const float dj = tris.d[(size_t)j * tris.stride];
            if (dj > zmax) continue;
            const uint32_t tj = tris[j];
            tris[m] = tj;
            tris.d[(size_t)m * tris.stride] = dj;
            ++m;
        }
        rec.ntris = m;
    }
    return rec.ntris + rec.overflow > 0;
}
template <class NS>
WT_HD bool bvh_traverse_cone_ns(const NS& ns, const scene_t& sc, const cone_t& cone, const range_t& searchrange, float z_scale, const stack_ref_t& stack,
                                const uint_list_t& tris, cone_hit_t& rec, bvh_counters_t* ctr = nullptr, uint32_t budget = 0xFFFFFFFFu,
                                float min_progress = -WT_INF) {
    cone_query_t q;
    cq_begin(sc, cone, searchrange, z_scale, stack, budget, min_progress, q);
    while (cq_running(q)) {   // "while-while" form, see bvh_traverse_ray
        while (q.s > 0 && q.leaf == 0) cq_node_step(ns, cone, stack, q, ctr);
        if (q.leaf != 0) cq_leaf_step(sc, cone, stack, tris, q, ctr);
    }
    const bool any = cq_end(cone, tris, q);
    rec = q.rec;
    return any;
}
WT_HD bool bvh_traverse_cone(const scene_t& sc, const cone_t& cone, const range_t& searchrange, float z_scale, const stack_ref_t& stack,
                             const uint_list_t& tris, cone_hit_t& rec, bvh_counters_t* ctr = nullptr, uint32_t budget = 0xFFFFFFFFu,
                             float min_progress = -WT_INF) {
    return bvh_traverse_cone_ns(lane_nodes(sc), sc, cone, searchrange, z_scale, stack, tris, rec, ctr, budget, min_progress);
}

// Any-hit cone probe: TRUE if some triangle intersects the cone inside `range` (first hit terminates).
// Used by the device's traverse(): a diffusive attempt is rejected whenever the closest cone hit lies within
// [dist, dist + major_axis/2) (traversal.hpp:146,157), i.e. exactly when this probe over that thin slab succeeds;
// probing first avoids the full (closest + triangle list) query whose result the reference computes and discards.
WT_HD bool bvh_cone_any_hit(const scene_t& sc, const cone_t& cone, const range_t& range, const stack_ref_t& stack, uint32_t budget, bool& aborted,
                             bvh_counters_t* ctr = nullptr) {
    aborted = false;
    if (sc.n_nodes == 0) return false;
    const vec3 ro = cone.o, rd = cone.d;
    const vec3 rinvd{1.f / rd.x, 1.f / rd.y, 1.f / rd.z};
    const bool sx = __builtin_signbit(rinvd.x), sy = __builtin_signbit(rinvd.y), sz = __builtin_signbit(rinvd.z);
    const float ta = cone.tan_alpha, ix = cone.x0;
    uint32_t tests = 0;
    int s = 1;
    stack.set(0, stack_entry_t{0.f, 1});
    const lane_nodes_t ns = lane_nodes(sc);
    const vec3 rel = ns.rel(ro);
    while (s > 0) {
        const stack_entry_t top = stack.get(s - 1);
        --s;
        if (top.ptr < 0) {
            const bvh8_leaf_t leaf = bvh_leaf_of(top.ptr);
            tests += leaf.count;
            if (ctr) ctr->probe_tri_tests += leaf.count;
            if (tests > budget) {
                aborted = true;
                return false;
            }
            for (uint32_t b = 0; b < leaf.count; b += kLeafBatch) {   // (fetched kLeafBatch at a time: ray_gather_tris)
                tri_geo_t T[kLeafBatch];
                const uint32_t m = leaf.count - b < kLeafBatch ? leaf.count - b : kLeafBatch;
#pragma unroll
                for (uint32_t i = 0; i < kLeafBatch; ++i)
                    if (i < m) T[i] = sc.tri_geo[leaf.tris_ptr + b + i];
#pragma unroll
                for (uint32_t i = 0; i < kLeafBatch; ++i) {
                    if (i >= m) break;
                    cone_tri_hit_t h;
                    if (intersect_cone_tri<true>(cone, T[i].a, T[i].b, T[i].c, T[i].n, range, h) && !(h.dist > range.max)) return true;
                }
            }
            continue;
        }
        // by value: the whole node is fetched with wide loads issued back to back (one memory latency per node instead of
        // one per child field); the unrolled child loop then runs from registers
        const typename lane_nodes_t::node_t n = ns.fetch(top.ptr - 1);
        if (ctr) ctr->probe_nodes++;
        tests += kNodeBudgetCost;
        if (tests > budget) {
            aborted = true;
            return false;
        }
        float tm[8];
        int32_t cps[8];
        uint32_t ok = 0;
        int room = (int)stack.cap - s;
        bool full = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int32_t cp = ns.child(n, i);
            const node_box_t nb = ns.box_rel(n, i, rel);
            float ominx = nb.x0, ominy = nb.y0, ominz = nb.z0;
            float omaxx = nb.x1, omaxy = nb.y1, omaxz = nb.z1;
            const float b0x = ominx, b0y = ominy, b0z = ominz, b1x = omaxx, b1y = omaxy, b1z = omaxz;
            const float bx = sx ? ominx : omaxx, by = sy ? ominy : omaxy, bz = sz ? ominz : omaxz;
            const float maxz = clampf(rd.x * bx + rd.y * by + rd.z * bz, 0.f, range.max);
            const float enlr = fmaf(maxz, ta, ix);
            ominx -= enlr; ominy -= enlr; ominz -= enlr;
            omaxx += enlr; omaxy += enlr; omaxz += enlr;
            const float dminx = (sx ? omaxx : ominx) * rinvd.x, dmaxx = (sx ? ominx : omaxx) * rinvd.x;
            const float dminy = (sy ? omaxy : ominy) * rinvd.y, dmaxy = (sy ? ominy : omaxy) * rinvd.y;
            const float dminz = (sz ? omaxz : ominz) * rinvd.z, dmaxz = (sz ? ominz : omaxz) * rinvd.z;
            float tmin = 0.f, tmax = dmaxx;
            tmin = fmaxf_(tmin, dminx);
            tmax = fminf_(tmax, dmaxy);
            tmin = fmaxf_(tmin, dminy);
            tmax = fminf_(tmax, dmaxz);
            tmin = fmaxf_(tmin, dminz);
#if WT_SS_ACTIVE
            const bool hit = cp != 0;
            tmin = 0.f;
#else
            const bool hit = cp != 0 && tmin <= tmax && tmax >= range.min && tmin <= range.max && !cone_box_outside(b0x, b0y, b0z, b1x, b1y, b1z, rd, ta, ix, range);
#endif
            const bool acc = hit && room > 0;
            full = full || (hit && !acc);
            room -= acc ? 1 : 0;
            ok |= acc ? 1u << i : 0u;
            tm[i] = tmin;
            cps[i] = cp;
        }
        if (full && budget != 0xFFFFFFFFu) {
            aborted = true;
            return false;
        }
        s = stack_push_sorted(stack, s, tm, cps, ok);
    }
    return false;
}

// ---- traversal policy (include/wt/integrator/traversal.hpp) -------------------------------------
constexpr float kBallisticScale = 1.001f;      // traversal.hpp:26

// traversal.hpp:39-57 (min_ballistic_distance is always 0 at the call sites: ray.o == envelope.o)
WT_HD float max_ballistic_distance(float lambda_m, uint32_t segment, float min_ballistic_distance) {
    const uint64_t max_segments = 16, seg_lambdas = 8, max_seg_lambdas = 1u << 16;
    const float min_dist = min_ballistic_distance * 1.05f;
    if (segment >= max_segments) return WT_INF;
    const uint64_t sh = seg_lambdas << (2 * segment + 1);
    const uint64_t B = sh < max_seg_lambdas ? sh : max_seg_lambdas;
    return min_dist + lambda_m * float(B);
}

struct trav_result_t {
    vec3 origin;
    uint32_t empty;
    uint32_t ballistic;
    float dist;
    float region_depth;
    uint32_t front_face;
    // ballistic (single ray hit); device, aborted == 2: the primary triangle of an overflowed region (g8.h: g8_resolve_primary)
    uint32_t tuid;
    float bx, by;
    float pdist;
    // diffusive
    uint32_t ntris;
    uint32_t overflow;
    // statistics
    uint32_t n_ray_queries, n_cone_queries;
    uint32_t aborted;   // device: the per-lane work budget was exceeded, redo cooperatively
};

// integrator::traverse (traversal.hpp:94-172). `envelope` already has its origin offset for self-intersection.
WT_HD trav_result_t traverse(const scene_t& sc, const cone_t& envelope, float lambda_m, float distance, bool force_ray_tracing,
                             const stack_ref_t& stack, const uint_list_t& tris, bvh_counters_t* ctr = nullptr, uint32_t cone_budget = 0xFFFFFFFFu,
                             bool probe_first = false) {
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
        if (ads_intersect_ray(sc, ro, rd, range_t{0.f, distance}, stack, rh, ctr)) {
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

    const float min_ballistic_distance = 0.f;
    float dist = 0.f;
    for (uint32_t seg = 0;; ++seg) {
        const float ballistic_dist = max_ballistic_distance(lambda_m, seg, min_ballistic_distance);
        r.n_ray_queries++;
        if (ads_intersect_ray(sc, ro, rd, range_t{dist, fminf_(distance, dist + ballistic_dist * kBallisticScale)}, stack, rh, ctr)) {
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

        // attempt diffusive propagation
        const float min_df_prog = cone_axes(envelope, dist).x / 2.f;
        cone_hit_t ch;
        r.n_cone_queries++;
        // probe_first (device): stop the query as soon as its closest hit is known to be too near to be accepted (result-
        // equivalent: the reference completes the query and then discards it, traversal.hpp:146,157)
        bvh_traverse_cone(sc, envelope, range_t{dist, distance}, kMajorAxisToZScale, stack, tris, ch, ctr, cone_budget,
                          probe_first ? min_df_prog : -WT_INF);
        if (ch.aborted) {
            // hand-over state for the cooperative kernel: every query before this one is settled (rays missed, diffusive attempts
            // were too short); it resumes with the cone query of segment `seg` at distance `dist`
            r.aborted = 1;
            r.dist = dist;
            r.ntris = seg;
            r.n_cone_queries--;   // recounted by whoever completes it
            return r;
        }
        if (ch.too_short) continue;
        const bool df_empty = ch.ntris == 0 && ch.overflow == 0;   // (a list of capacity 0 — closest-hit-only queries — counts every hit as overflow)
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
        // too short: continue the ballistic path
    }
}

// Upper bound of [closest cone hit + region depth] from the closest hit t of the beam AXIS: the axis lies inside the cone, so the
// closest cone hit is not farther than t, and closest + 2 x axis(closest) grows with closest (slack: rounding of the two tests).
WT_HD float cone_axis_bound(const cone_t& envelope, float t_axis) {
    const float t = t_axis * 1.0001f + 1e-9f;
    return t + kMajorAxisToZScale * cone_axes(envelope, t).x;
}
// The triangle under the beam axis from the axis hit (see resolve_primary): marks r.aborted = 2.
WT_HD void primary_from_axis(const scene_t& sc, const cone_t& envelope, bool axis_hit, const ray_hit_t& ah, trav_result_t& r) {
    r.aborted = 2;
    r.tuid = kInvalid;
    if (!axis_hit) return;
    const range_t izr{r.dist, r.dist + r.region_depth};
    const tri_geo_t g = sc.tri_geo[ah.tuid];
    if (contains(grow(izr, cone_intersection_tolerance(envelope.o, g.a, g.b, g.c)), ah.dist)) {
        r.tuid = ah.tuid;
        r.bx = ah.bx;
        r.by = ah.by;
        r.pdist = ah.dist;
    }
}

// TRUE: triangle `tuid` alone makes the diffusive attempt that searches `sr` too short — the cone meets it less than `min_progress`
// behind the start of the search.  The closest hit of a cone query is the minimum over all triangles, so whenever this holds the full
// query ends "too short" as well (bvh_traverse_cone's early exit, traversal.hpp:146,157): the query need not run.
WT_HD bool cone_attempt_too_short_by(const scene_t& sc, const cone_t& cone, uint32_t tuid, const range_t& sr, float min_progress) {
    const tri_geo_t tri = sc.tri_geo[tuid];
    cone_tri_hit_t h;
    return intersect_cone_tri(cone, tri.a, tri.b, tri.c, tri.n, sr, h) && !(h.dist > sr.max) && h.dist - sr.min < min_progress;
}

// integrator::traverse, device form — same results as traverse() above (tests/test_oracle.py::test_traverse_axis_equals_traverse),
// less work:
//   * ONE closest-hit query of the beam axis over the whole range replaces the ray query of every ballistic segment: the segments
//     tile [0, distance) (traversal.hpp:39-57), so "segment k's ray query hits" == "the closest axis hit lies in segment k";
//   * that hit bounds every cone query from above (cone_axis_bound): a beam wider than the BVH's leaf boxes gets no useful
//     near-first order from its grown boxes and would otherwise test geometry far behind the surface it is about to hit;
//   * and it IS the triangle under the beam axis of the interaction region (find_closest_triangle) whenever the region's bounded
//     list overflowed (`primary_on_overflow`; a complete list is scanned like the reference does);
//   * a beam that leaves a surface spends its first diffusive attempts (58 % of all cone queries in the headline workload) being
//     rejected as "too short" by the very surface it left.  Two remembered triangles — the one the beam started from (`origin_tuid`)
//     and the one that made the previous attempt too short — are tested first (cone_attempt_too_short_by): 72 % of those attempts
//     are decided by one exact cone-triangle test instead of a BVH query (tools: oracle_profile_axis).
// Hand-over when the cone query exceeds `cone_budget` (r.aborted = 1): r.dist / r.ntris = distance / segment of that query, the
// axis hit in r.tuid (kInvalid: none) / r.bx / r.by / r.pdist / r.front_face, the last rejecting triangle in r.overflow.
// `resume`: continue a handed-over traversal — the cone query of segment resume->ntris at distance resume->dist, axis hit and
// query counts from the record (everything before is settled).
//
// Like the cone query the policy is written as resumable steps over an explicit state (axis_walk_t): aw_begin after the axis query,
// aw_next until it either returns true — a cone query is due: q describes it — or false — the traversal is over, `r` holds the record;
// aw_query_done feeds a finished query back.  traverse_axis is the sequential driver; k_trace drives 64 of them in lock step.
struct axis_walk_t {
    float lambda_m, distance;
    uint32_t axis_hit;
    ray_hit_t ah;
    uint32_t seg, first_seg_settled;   // first_seg_settled: resuming a handed-over traversal — segment `seg` is settled, `dist` is past it
    float dist;
    uint32_t short_tuid, origin_tuid;   // the two remembered triangles
    uint32_t n_ray_queries, n_cone_queries;
    uint32_t cone_budget, probe_first, primary_always;
    uint32_t use_cache;   // (diagnostic switch: 0 = the remembered triangles are not consulted)
    float min_df_prog;
    range_t sr;                 // search range of the current segment's diffusive attempt
    uint32_t cand_idx, cand;    // remembered triangles of the current segment tried so far (0: at the start of a segment) / the one to test
};
WT_HD void trav_result_init(trav_result_t& r, vec3 origin) {
    r.aborted = 0;
    r.origin = origin;
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
}
WT_HD void aw_ballistic_hit(const axis_walk_t& a, trav_result_t& r) {
    r.empty = 0;
    r.dist = a.ah.dist;
    r.tuid = a.ah.tuid;
    r.bx = a.ah.bx;
    r.by = a.ah.by;
    r.front_face = a.ah.front_face;
    r.ntris = 1;
}
// What the policy needs next.  AW_FINAL: nothing — the traversal is over, `r` is final.  AW_QUERY: the cone query `q` (begun on `stack`)
// has to run; feed its outcome to aw_query_done.  AW_TEST: one exact cone-triangle test of a remembered triangle — a.cand against a.sr
// with a.min_df_prog (cone_attempt_too_short_by) — whose answer goes to aw_test_done.  (The test is handed OUT instead of being done
// here so that a kernel can run the exact tests of many lanes — these and the survivors of the leaf filters — together.)
enum aw_need_e : int { AW_FINAL = 0, AW_QUERY = 1, AW_TEST = 2 };
WT_HD int aw_next(const scene_t& sc, const cone_t& envelope, bool force_ray_tracing, const stack_ref_t& stack, axis_walk_t& a, cone_query_t& q, trav_result_t& r) {
    if (a.cand_idx == 0) {   // (not in the middle of a segment's remembered-triangle tests)
        trav_result_init(r, envelope.o);
        if (force_ray_tracing || cone_is_ray(envelope)) {
            if (a.axis_hit) aw_ballistic_hit(a, r);
            r.n_ray_queries = a.n_ray_queries;
            r.n_cone_queries = a.n_cone_queries;
            return AW_FINAL;
        }
    }
    for (;; ++a.seg) {
        if (a.cand_idx == 0) {
            const float ballistic_dist = max_ballistic_distance(a.lambda_m, a.seg, 0.f);
            if (!a.first_seg_settled) {
                if (a.axis_hit && a.ah.dist <= fminf_(a.distance, a.dist + ballistic_dist * kBallisticScale)) {
                    aw_ballistic_hit(a, r);
                    break;
                }
                a.dist += ballistic_dist;
                if (ballistic_dist == WT_INF || a.dist >= a.distance) break;
            }
            a.first_seg_settled = 0;
            a.min_df_prog = cone_axes(envelope, a.dist).x / 2.f;
            a.n_cone_queries++;
            const float cone_max = a.axis_hit ? fminf_(a.distance, cone_axis_bound(envelope, a.ah.dist)) : a.distance;
            a.sr = range_t{a.dist, cone_max};
        }
        if (a.probe_first && a.use_cache) {   // (device form only: decides "too short" exactly like the query's own early exit)
            while (a.cand_idx < 2) {
                const uint32_t cand = a.cand_idx == 0 ? a.short_tuid : a.origin_tuid;
                const bool skip = cand == kInvalid || (a.cand_idx == 1 && cand == a.short_tuid);
                ++a.cand_idx;
                if (skip) continue;
                a.cand = cand;
                return AW_TEST;
            }
        }
        a.cand_idx = 0;
        cq_begin(sc, envelope, a.sr, kMajorAxisToZScale, stack, a.cone_budget, a.probe_first ? a.min_df_prog : -WT_INF, q);
        return AW_QUERY;
    }
    r.n_ray_queries = a.n_ray_queries;
    r.n_cone_queries = a.n_cone_queries;
    return AW_FINAL;
}
// the answer to AW_TEST; call aw_next again afterwards
WT_HD void aw_test_done(axis_walk_t& a, bool too_short) {
    if (too_short) {   // the attempt is settled without a query: on to the next segment
        a.cand_idx = 0;
        ++a.seg;
    }
}
// The outcome of the query aw_next asked for (after cq_end).  TRUE: the traversal is over, `r` is final; FALSE: call aw_next again.
WT_HD bool aw_query_done(const scene_t& sc, const cone_t& envelope, axis_walk_t& a, const cone_hit_t& ch, trav_result_t& r) {
    if (ch.aborted) {
        trav_result_init(r, envelope.o);
        r.aborted = 1;
        r.dist = a.dist;
        r.ntris = a.seg;
        r.n_ray_queries = a.n_ray_queries;
        r.n_cone_queries = a.n_cone_queries - 1;   // recounted by whoever completes it
        r.tuid = a.axis_hit ? a.ah.tuid : kInvalid;
        r.bx = a.ah.bx;
        r.by = a.ah.by;
        r.pdist = a.ah.dist;
        r.front_face = a.ah.front_face;
        r.overflow = a.short_tuid;
        return true;
    }
    if (ch.too_short) {
        a.short_tuid = ch.short_tuid;
        ++a.seg;
        return false;
    }
    const bool df_empty = ch.ntris == 0 && ch.overflow == 0;   // (a list of capacity 0 — closest-hit-only queries — counts every hit as overflow)
    if (df_empty || ch.dist - a.dist >= a.min_df_prog) {
        trav_result_init(r, envelope.o);
        r.n_ray_queries = a.n_ray_queries;
        r.n_cone_queries = a.n_cone_queries;
        r.ballistic = 0;
        r.empty = df_empty;
        r.dist = df_empty ? -WT_INF : ch.dist;
        r.front_face = ch.front_face;
        r.ntris = ch.ntris;
        r.overflow = ch.overflow;
        r.region_depth = df_empty ? 0.f : kMajorAxisToZScale * cone_axes(envelope, ch.dist).x;
        if (!df_empty && (a.primary_always || ch.overflow > 0)) primary_from_axis(sc, envelope, a.axis_hit != 0, a.ah, r);
        return true;
    }
    ++a.seg;   // too short: continue the ballistic path
    return false;
}
WT_HD void aw_begin(axis_walk_t& a, float lambda_m, float distance, bool axis_hit, const ray_hit_t& ah, uint32_t cone_budget, bool probe_first, bool primary_always,
                    uint32_t origin_tuid) {
    a.lambda_m = lambda_m;
    a.distance = distance;
    a.axis_hit = axis_hit ? 1u : 0u;
    a.ah = ah;
    a.seg = 0;
    a.first_seg_settled = 0;
    a.dist = 0.f;
    a.short_tuid = kInvalid;
    a.origin_tuid = origin_tuid;
    a.n_ray_queries = 1;
    a.n_cone_queries = 0;
    a.cone_budget = cone_budget;
    a.probe_first = probe_first ? 1u : 0u;
    a.primary_always = primary_always ? 1u : 0u;
    a.use_cache = 1;
    a.min_df_prog = 0.f;
    a.sr = range_t{0.f, 0.f};
    a.cand_idx = 0;
    a.cand = kInvalid;
}
// ... of a handed-over traversal (aw_query_done's `aborted` record)
WT_HD void aw_resume(axis_walk_t& a, float lambda_m, float distance, const trav_result_t& h, uint32_t cone_budget, bool probe_first, bool primary_always,
                     uint32_t origin_tuid) {
    ray_hit_t ah;
    ah.tuid = h.tuid;
    ah.bx = h.bx;
    ah.by = h.by;
    ah.dist = h.pdist;
    ah.front_face = h.front_face;
    aw_begin(a, lambda_m, distance, h.tuid != kInvalid, ah, cone_budget, probe_first, primary_always, origin_tuid);
    a.seg = h.ntris;
    a.first_seg_settled = 1;
    a.dist = h.dist;
    a.short_tuid = h.overflow;
    a.n_ray_queries = h.n_ray_queries;
    a.n_cone_queries = h.n_cone_queries;
}
WT_HD trav_result_t traverse_axis(const scene_t& sc, const cone_t& envelope, float lambda_m, float distance, bool force_ray_tracing,
                                  const stack_ref_t& stack, const uint_list_t& tris, bvh_counters_t* ctr = nullptr, uint32_t cone_budget = 0xFFFFFFFFu,
                                  bool probe_first = false, bool primary_always = false, uint32_t origin_tuid = kInvalid,
                                  const trav_result_t* resume = nullptr) {
    axis_walk_t a;
    if (resume)
        aw_resume(a, lambda_m, distance, *resume, cone_budget, probe_first, primary_always, origin_tuid);
    else {
        ray_hit_t ah;
        const bool axis_hit = ads_intersect_ray(sc, envelope.o, envelope.d, range_t{0.f, distance}, stack, ah, ctr);
        aw_begin(a, lambda_m, distance, axis_hit, ah, cone_budget, probe_first, primary_always, origin_tuid);
    }
    trav_result_t r;
    cone_query_t q;
    for (;;) {
        const int need = aw_next(sc, envelope, force_ray_tracing, stack, a, q, r);
        if (need == AW_FINAL) break;
        if (need == AW_TEST) {
            aw_test_done(a, cone_attempt_too_short_by(sc, envelope, a.cand, a.sr, a.min_df_prog));
            continue;
        }
        while (cq_running(q)) {
            while (q.s > 0 && q.leaf == 0) cq_node_step(lane_nodes(sc), envelope, stack, q, ctr);
            if (q.leaf != 0) cq_leaf_step(sc, envelope, stack, tris, q, ctr);
        }
        cq_end(envelope, tris, q);
        if (aw_query_done(sc, envelope, a, q.rec, r)) break;
    }
    return r;
}

// The triangle under the beam axis of a diffusive hit (find_closest_triangle, plt_bdpt_detail.hpp:362-389: the closest axis hit
// among the triangles of the interaction region) by ONE ray query over the region's z-slab instead of a scan of the region's
// triangle list: a triangle the axis hits inside the slab meets the cone inside the slab, i.e. is a region triangle, whatever the
// size of the region.  Marks r.aborted = 2: "primary in r.tuid / r.bx / r.by / r.pdist (kInvalid: the axis misses the region)".
WT_HD void resolve_primary(const scene_t& sc, const cone_t& envelope, const stack_ref_t& stack, trav_result_t& r, bvh_counters_t* ctr = nullptr) {
    const range_t izr{r.dist, r.dist + r.region_depth};
    const float wtol = cone_intersection_tolerance(envelope.o, sc.world_min, sc.world_max, sc.world_max);
    r.aborted = 2;
    r.tuid = kInvalid;
    r.n_ray_queries++;
    ray_hit_t rh;
    if (ads_intersect_ray(sc, envelope.o, envelope.d, grow(izr, wtol), stack, rh, ctr)) {
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

// The ordered, de-duplicated set of classified edges of an interaction region (traversal_common.hpp:124-148: the edges of every
// triangle that meets the traced cone inside the region's slab), for regions of ANY size: the walk descends only into subtrees
// that hold classified edges (bvh8_node_t::edge_mask) and tests only edge-bearing triangles, so a region of 10^5 smooth-mesh
// triangles costs a few node visits.  `out`: sorted ids (capacity `cap`); returns the count, `overflow` = ids dropped.
WT_HD uint32_t bvh_gather_edges(const scene_t& sc, const cone_t& tcone, const range_t& slab, const stack_ref_t& stack, uint32_t* out, uint32_t cap,
                                uint32_t& overflow) {
    overflow = 0;
    uint32_t n_out = 0;
    if (sc.n_nodes == 0) return 0;
    const vec3 ro = tcone.o, rd = tcone.d;
    const float ta = tcone.tan_alpha, ix = tcone.x0;
    int s = 1;
    stack.set(0, stack_entry_t{0.f, 1});
    while (s > 0) {
        const int32_t ptr = stack.get(s - 1).ptr;
        --s;
        uint32_t t0, cnt;
        if (ptr < 0) {
            const bvh8_leaf_t leaf = bvh_leaf_of(ptr);
            t0 = leaf.tris_ptr;
            cnt = leaf.count;
        } else {
            const bvh8_node_t& n = sc.nodes[ptr - 1];
            const uint32_t em = n.edge_mask;
            for (int i = 0; i < 8; ++i) {
                if (!((em >> i) & 1u)) continue;
                if (cone_box_outside(n.minx[i] - ro.x, n.miny[i] - ro.y, n.minz[i] - ro.z, n.maxx[i] - ro.x, n.maxy[i] - ro.y, n.maxz[i] - ro.z, rd, ta, ix, slab)) continue;
                const bool room = s < (int)stack.cap;   // (always: the pruned tree is a few levels of a handful of nodes)
                overflow += room ? 0u : 1u;
                if (room) {
                    stack.set(s, stack_entry_t{0.f, n.child[i]});
                    ++s;
                }
            }
            continue;
        }
        for (uint32_t t = 0; t < cnt; ++t) {
            const tri_meta_t m = sc.tri_meta[t0 + t];
            if (m.edge[0] == kInvalid && m.edge[1] == kInvalid && m.edge[2] == kInvalid) continue;
            const tri_geo_t tri = sc.tri_geo[t0 + t];
            cone_tri_hit_t h;
            if (!intersect_cone_tri(tcone, tri.a, tri.b, tri.c, tri.n, slab, h) || h.dist > slab.max) continue;
            for (int e = 0; e < 3; ++e) {
                const uint32_t id = m.edge[e];
                if (id == kInvalid) continue;
                uint32_t pos = 0;
                while (pos < n_out && out[pos] < id) ++pos;
                if (pos < n_out && out[pos] == id) continue;
                if (n_out == cap) {
                    overflow++;
                    continue;
                }
                for (uint32_t j = n_out; j > pos; --j) out[j] = out[j - 1];
                out[pos] = id;
                ++n_out;
            }
        }
    }
    return n_out;
}

}   // namespace wt
