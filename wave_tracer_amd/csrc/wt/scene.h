// wave_tracer_amd — flattened (struct-of-pointers) scene description consumed by the hot path.
//
// This is the drop-in boundary's payload: everything the per-sample
// integrator dereferences, baked once on the host.  The same POD layout is used with host pointers
// (CPU checker, scene baking) and with device pointers (HIP kernels); it contains no virtuals.
//
// Reference for the individual blocks:
//   triangles/edges      include/wt/ads/common.hpp:37-72, src/ads/bvh8w_constructor.cpp:119-151
//   BVH8 nodes           include/wt/ads/bvh8w/bvh8w_node.hpp:18-41, ads/bvh8w/common.hpp:29-43
//   materials            include/wt/bsdf/*.hpp, src/bsdf/*.cpp (flattened wrappers two_sided/scale/composite)
//   spectra              src/spectrum/*.cpp (baked to tables)
//   emitters             include/wt/emitter/{spot,area}.hpp
//   sensors / film       include/wt/sensor/sensor/{perspective,virtual_plane_sensor}.hpp, sensor/film/film.hpp
//   sampling tables      src/scene/scene_build_sensor_sampling_data.cpp:40-150
#pragma once
#include "core.h"

namespace wt {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;

struct alignas(16) tri_geo_t {   // 48 B, 3 x 16-B loads
    vec3 a, b, c, n;
};
struct tri_meta_t {   // 20 B
    uint32_t shape_idx, shape_tri_idx;
    uint32_t edge[3];   // edge_ab, edge_bc, edge_ca (kInvalid = none)
};
struct tri_shade_t {   // per triangle shading data (mesh/triangle.hpp), BVH order
    vec3 n0, n1, n2;
    vec2 uv0, uv1, uv2;
    vec3 dpdu;
    uint32_t has_uv;
};
struct edge_t {   // ads/common.hpp:53-72
    vec3 a, b, e;
    vec3 n1, t1, n2, t2;
    float alpha;
    uint32_t tri1, tri2;   // kInvalid = boundary edge
};

// 8-wide BVH node: child AABBs in SoA inside the node.
// child ptr: 0 empty, >0 internal node index+1, <0 -(leaf index+1); root ptr = 1.
struct alignas(16) bvh8_node_t {
    float minx[8], miny[8], minz[8];
    float maxx[8], maxy[8], maxz[8];
    int32_t child[8];
    uint32_t tris_start, tris_count;
    uint32_t edge_mask;   // bit i: the subtree of child i holds a triangle with a classified edge (ads/common.hpp:53-72); prunes the
                          // interaction-region edge gather (bvh_gather_edges) — a handful of silhouette edges in 10^5 triangles
    uint32_t pad[5];   // 256 B
};
struct bvh8_leaf_t {
    uint32_t tris_ptr, count;
};
// A child reference of an 8-wide node: 0 = none, > 0 = node index + 1, < 0 = a LEAF NAMED BY VALUE: -((tris_ptr << 3) | count), count in
// 1..7.  (The reference's node points into a leaf array, bvh8w_node.hpp:27-41; here the triangles of a leaf follow from the reference
// itself, which takes one dependent memory round trip out of every leaf visit of every traversal.  scene_t::leaves still lists the
// leaves for hosts that want to iterate them; the traversals do not read it.)

struct shape_t {
    int32_t material;
    int32_t emitter;   // -1 none
    float surface_area, recp_surface_area;
    uint32_t tri_offset;   // into shape_tri_tuid / shape_tri_cdf
    uint32_t tri_count;
};

// ---- spectra --------------------------------------------------------------------------------------
enum spectrum_type_e : int32_t { SPEC_CONST = 0, SPEC_TABLE = 1, SPEC_DISCRETE = 2 };
struct spectrum_t {
    int32_t type;
    float kmin, kmax;         // support [1/mm]; value 0 outside (table) / line position in kmin (discrete)
    uint32_t offset, count;   // into spectra_data: count knots uniformly spaced in k over [kmin,kmax]
    float c_re, c_im;         // constant value / discrete line value
    uint32_t is_complex;      // table holds count re-values followed by count im-values
};

// ---- materials -----------------------------------------------------------------------------------
enum material_type_e : int32_t { MAT_DIFFUSE = 0, MAT_DIELECTRIC = 1, MAT_SURFACE_SPM = 2, MAT_COMPOSITE = 3, MAT_MASK = 4 };
enum profile_type_e : int32_t { PROFILE_DIRAC = 0, PROFILE_FRACTAL = 1, PROFILE_GAUSSIAN = 2 };
struct material_t {
    int32_t type;
    uint32_t two_sided;     // bsdf/two_sided wrapper
    float scale;            // bsdf/scale wrapper (constant texture)
    int32_t refl_spec;      // diffuse: reflectance spectrum
    float refl_tex_scale;   // constant stand-in for texture modulation of reflectance
    int32_t ior_spec;       // dielectric / surface_spm: interior IOR spectrum (complex)
    int32_t ext_ior_spec;   // exterior IOR spectrum (-1 = 1)
    int32_t profile;        // surface_spm: PROFILE_*
    float roughness;        // fractal: perceptual roughness
    float gamma;            // fractal: log-log slope
    float gauss_sigma;      // gaussian: > 0: explicit rms `sigma` [1/mm]; otherwise parametrised by `roughness` like the fractal profile
    float refl_scale, trans_scale;
    // composite (bsdf/composite.hpp:26-140): spectral bins [kmin, kmax) [1/mm] -> child material; no BSDF outside the bins
    uint32_t n_bins;
    float bin_kmin[4], bin_kmax[4];   // kMaxCompositeBins (wt/bsdf.h)
    int32_t bin_child[4];
    // mask (src/bsdf/mask.cpp:24-92): nested material seen through a mask of opacity alpha
    int32_t nested;
    float mask_alpha;       // constant mask, used when mask_tex == 0
    // textures (the headers under include/wt/texture): texture index + 1, 0 = none (so that a zero-initialised record has no textures)
    uint32_t refl_tex;      // diffuse: reflectance = clamp01(spectrum * refl_tex_scale * texture) (scale.hpp wrapping a texture)
    uint32_t mask_tex;      // mask: opacity texture (0: the constant mask_alpha)
    uint32_t normal_tex;    // normalmap wrapper (bsdf/normalmap.hpp:48-62), flattened onto the material it wraps
    uint32_t normal_flip;
    // scale wrapper whose factor is a spectrum (bsdf/scale.hpp:78-97 with a spectrum in place of a constant): spectrum index + 1, 0 = none;
    // multiplies `scale`
    uint32_t scale_spec;
    uint32_t scale_tex;     // ... or a texture (scale->f(tquery).x): texture index + 1, 0 = none
    uint32_t rough_tex;     // fractal / gaussian profile: perceptual roughness from a texture (fractal.hpp:83-92: roughness_tex->f(query).x): index + 1
};

// ---- textures (include/wt/texture/texture.hpp:29-90) ------------------------------------------------------------------------------
// A texture record is one of constant / checkerboard / bitmap, with the two generic wrappers folded in: `transform` (uv' = M uv + t,
// texture/transform.hpp:35-44; identity when absent) applied before the lookup and `scale` by a constant (texture/scale.hpp:95-97; 1
// when absent) applied after it.  Bitmaps are float texels (linear, 1..4 channels: luminance, luminance+alpha, RGB, RGBA), rows from the
// image's top; luminance textures are wavelength independent (bitmap.hpp:84-99), RGB ones are only read through get_RGBA (normal maps).
// TEX_FUNCTION (texture/function.hpp, texture/mix.hpp): a real-valued expression of nested textures and of u, v, k, compiled by the host into
// a postfix program of (opcode, argument) float pairs in texture_data[offset .. offset + width): see texture_function below.
enum texture_type_e : int32_t { TEX_CONSTANT = 0, TEX_CHECKERBOARD = 1, TEX_BITMAP = 2, TEX_FUNCTION = 3 };
enum texture_op_e : int32_t { TOP_CONST = 0, TOP_U = 1, TOP_V = 2, TOP_K = 3, TOP_TEX = 4, TOP_ADD = 5, TOP_SUB = 6, TOP_MUL = 7, TOP_DIV = 8, TOP_NEG = 9, TOP_POW = 10, TOP_MIN = 11, TOP_MAX = 12, TOP_ABS = 13, TOP_SQRT = 14, TOP_SIN = 15, TOP_COS = 16, TOP_TAN = 17, TOP_EXP = 18, TOP_LOG = 19, TOP_FLOOR = 20, TOP_CEIL = 21, TOP_ROUND = 22, TOP_ASIN = 23, TOP_ACOS = 24, TOP_ATAN = 25, TOP_ATAN2 = 26, TOP_MIX = 27, TOP_LT = 28, TOP_LE = 29, TOP_GT = 30, TOP_GE = 31, TOP_EQ = 32, TOP_NE = 33, TOP_AND = 34, TOP_OR = 35, TOP_NOT = 36 };
enum texture_wrap_e : uint32_t { WRAP_BLACK = 0, WRAP_WHITE = 1, WRAP_CLAMP = 2, WRAP_REPEAT = 3, WRAP_MIRROR = 4 };
struct texture_t {
    int32_t type;
    float rgba[4];        // TEX_CONSTANT
    int32_t col1, col2;   // TEX_CHECKERBOARD: the two nested textures
    float m[4], t[2];     // transform: uv' = (m[0] u + m[1] v + t[0], m[2] u + m[3] v + t[1])
    float scale;
    uint32_t width, height, channels, offset;   // TEX_BITMAP: texel (x, y) channel c = texture_data[offset + (y * width + x) * channels + c]
                                                // TEX_FUNCTION: the program = texture_data[offset .. offset + width)
    uint32_t bilinear;    // the filter — 0: nearest, 1: bilinear, 2: bicubic (the name is the public header's)
    uint32_t uwrap, vwrap;
};

// ---- emitters ------------------------------------------------------------------------------------
enum emitter_type_e : int32_t { EMIT_SPOT = 0, EMIT_AREA = 1, EMIT_POINT = 2, EMIT_DIRECTIONAL = 3 };
struct emitter_t {
    int32_t type;
    int32_t spectrum;   // radiant intensity (spot) / radiance (area), value multiplies `scale`
    float scale;
    float phase_space_extent_scale;
    // spot
    vec3 position;
    frame_t frame;   // to_world rotation: local z = mean direction
    float cutoff, falloff, cos_cutoff, cos_falloff, recp_cutoff_range, max_tan_alpha;
    float extent;   // <=0: default 10 lambda
    // directional (infinite emitter): position = world centre, frame.n = direction TO the emitter; the target is the disk that
    // bounds the world AABB projected along that direction (directional.hpp:46-75)
    float target_radius, target_area, far_dist, tan_alpha_at_target;
    // area
    int32_t shape;
    // area, spatially varying radiance (area.hpp:103-116, src/emitter/area.cpp:153-260): radiance = scale x texture.f({uv, k}).x of the BITMAP
    // texture radiance_tex - 1 (0: `spectrum` alone), `spectrum` holds the texture's mean spectrum (emitter selection, spectral sampling), and positions are
    // drawn from per-triangle texel tables in texture_data[tab .. tab + tab_words): see wt/sources.h area_table_*
    int32_t radiance_tex;
    uint32_t tab, tab_words;
    // sampling tables
    float select_pmf;           // emitters_power_distribution.pdf
    int32_t k_dist;             // index into kdists
};

// spectral sampling distribution per emitter (emission x sensitivity product)
struct kdist_t {
    int32_t discrete;          // 1: single line at kmin with mass 1
    float kmin, kmax;
    uint32_t offset, count;    // pdf knots (count), cdf knots (count) in kdist_data: pdf[0..count), cdf[0..count)
};

// ---- sensor / film ------------------------------------------------------------------------------
enum sensor_type_e : int32_t { SENSOR_PERSPECTIVE = 0, SENSOR_VIRTUAL_PLANE = 1 };
struct sensor_t {
    int32_t type;
    uint32_t width, height, channels;
    uint32_t polarimetric;   // film stores the 4 Stokes components per channel (sensor_t<polarimetric>, film.hpp) instead of intensity
    uint32_t ray_trace_only;
    // film reconstruction filter
    float rfilter_sigma;   // in pixels (= .25 * rfilter_scale)
    int32_t rf_radius;
    uint32_t flip_x, flip_y;
    // response: per-channel spectrum ids
    int32_t response_spec[4];
    // perspective
    vec3 position;
    frame_t frame;            // camera to world rotation (t = right, b = up, n = view dir)
    float inv_cam[16];        // inverse(viewport * perspective), row-major 4x4
    float cam[16];            // viewport * perspective, row-major
    vec3 ddir_dx, ddir_dy;
    float sensor_area;        // prod(snsr_extent) [m^2]
    float element_extent_x;   // [m]
    float sourcing_tan_alpha;
    float phase_space_extent_scale;
    // virtual plane
    vec3 origin;              // sensor_origin (corner)
    vec2 extent, element_extent;
    float recp_area;
    float requested_tan_alpha;   // <0: none (MUB)
};

enum integrator_type_e : uint32_t { INTEGRATOR_BDPT = 0, INTEGRATOR_PATH_FORWARD = 1, INTEGRATOR_PATH_BACKWARD = 2 };
struct integrator_opts_t {
    int32_t max_depth;
    uint32_t integrator;   // INTEGRATOR_* (plt_bdpt, or plt_path with its transport direction)
    uint32_t MIS, RR, FSD, sensor_direct, emitter_direct;
    uint32_t force_ray_tracing;
    // test hooks: evaluate a single (s,t) strategy with unit MIS weight (0 = all strategies; v>0 selects v-1)
    uint32_t debug_only_s, debug_only_t;
};

// Fraunhofer FSD inverse-CDF LUT (interaction/fsd/fraunhofer/fsd_lut.hpp:27-69), regenerated on the host.
struct fsd_lut_t {
    uint32_t n_theta;   // Nsamples
    uint32_t m;         // Msamples (square)
    const float* icdf_theta1;
    const float* icdf_theta2;
    const float* icdf1;   // [m][m]
    const float* icdf2;
};

struct scene_t {
    // geometry
    const tri_geo_t* tri_geo;
    const tri_meta_t* tri_meta;
    const tri_shade_t* tri_shade;
    uint32_t n_tris;
    const edge_t* edges;
    uint32_t n_edges;
    const bvh8_node_t* nodes;
    uint32_t n_nodes;
    const bvh8_leaf_t* leaves;
    uint32_t n_leaves;
    vec3 world_min, world_max;
    // shapes
    const shape_t* shapes;
    uint32_t n_shapes;
    const uint32_t* shape_tri_tuid;   // per shape: mesh tri index -> tuid
    const float* shape_tri_cdf;       // per shape: area cdf (tri_count+1 entries each, concatenated with +shape index offset)
    // materials & spectra
    const material_t* materials;
    uint32_t n_materials;
    const spectrum_t* spectra;
    uint32_t n_spectra;
    const float* spectra_data;
    const texture_t* textures;
    uint32_t n_textures;
    const float* texture_data;
    // emitters
    const emitter_t* emitters;
    uint32_t n_emitters;
    const float* emitter_cdf;   // n_emitters+1
    const kdist_t* kdists;
    const float* kdist_data;
    // sensor
    sensor_t sensor;
    integrator_opts_t opts;
    fsd_lut_t lut;
};

// ---- spectrum evaluation ---------------------------------------------------------------------
WT_HD cplx spectrum_value(const scene_t& sc, int id, float k) {
    if (id < 0) return {1.f, 0.f};
    const spectrum_t s = sc.spectra[id];
    if (s.type == SPEC_CONST) return {s.c_re, s.c_im};
    if (s.type == SPEC_DISCRETE) return k == s.kmin ? cplx{s.c_re, s.c_im} : cplx{0.f, 0.f};
    if (k < s.kmin || k > s.kmax) return {0.f, 0.f};
    const float x = (k - s.kmin) / (s.kmax - s.kmin) * float(s.count - 1);
    uint32_t l = (uint32_t)x;
    if (l > s.count - 1) l = s.count - 1;
    const uint32_t h = l + 1 < s.count ? l + 1 : s.count - 1;
    const float f = x - float(l);
    const float* d = sc.spectra_data + s.offset;
    cplx r{d[l] * (1.f - f) + d[h] * f, 0.f};
    if (s.is_complex) r.im = d[s.count + l] * (1.f - f) + d[s.count + h] * f;
    return r;
}
WT_HD float spectrum_f(const scene_t& sc, int id, float k) { return spectrum_value(sc, id, k).re; }

// ---- texture evaluation ---------------------------------------------------------------------------------------------------------
struct rgba_t {
    float r, g, b, a;
};
WT_HD int tex_modulo(int a, int b) {
    const int r = a % b;
    return r < 0 ? r + b : r;
}
// bitmap/texture2d_storage.hpp:80-97 (-1: outside, constant texel)
WT_HD int tex_wrap_coord(uint32_t wrap, int c, int dim) {
    if (c >= 0 && c < dim) return c;
    switch (wrap) {
    case WRAP_CLAMP: return c < 0 ? 0 : (dim > 1 ? dim : 1) - 1;
    case WRAP_REPEAT: return tex_modulo(c, dim);
    case WRAP_MIRROR: {
        const int m2 = tex_modulo(c, 2 * dim);
        return m2 >= dim ? 2 * dim - 1 - m2 : m2;
    }
    default: return -1;
    }
}
// texture2d.hpp:233-268
WT_HD rgba_t tex_texel(const scene_t& sc, const texture_t& t, int x, int y) {
    x = tex_wrap_coord(t.uwrap, x, (int)t.width);
    y = tex_wrap_coord(t.vwrap, y, (int)t.height);
    if (x < 0 || y < 0) {
        const float v = (x < 0 ? t.uwrap : t.vwrap) == WRAP_BLACK ? 0.f : 1.f;
        return {v, v, v, 1.f};
    }
    const float* p = sc.texture_data + t.offset + ((size_t)y * t.width + (size_t)x) * t.channels;
    switch (t.channels) {
    case 1: return {p[0], p[0], p[0], 1.f};
    case 2: return {p[0], p[0], p[0], p[1]};
    case 3: return {p[0], p[1], p[2], 1.f};
    default: return {p[0], p[1], p[2], p[3]};
    }
}
WT_HD rgba_t tex_mix(rgba_t a, rgba_t b, float f) { return {a.r + (b.r - a.r) * f, a.g + (b.g - a.g) * f, a.b + (b.b - a.b) * f, a.a + (b.a - a.a) * f}; }
// texture2d.hpp:284-310, 356-392 (v is flipped: uv (0,0) is the image's bottom-left corner); filtered texels are clamped to be
// non-negative (the default texel_clamp_mode)
// the reference's cubic (texture2d.hpp:323-329), term by term
WT_HD float tex_cubic(float x, float p0, float p1, float p2, float p3) {
    return p1 + .5f * x * (-p0 + p2) + .5f * x * x * (2.f * p0 - 5.f * p1 + 4.f * p2 - p3) + .5f * x * x * x * (-p0 + 3.f * p1 - 3.f * p2 + p3);
}
// texture2d_t::bicubic_native (include/wt/bitmap/texture2d.hpp:316-343: Catmull-Rom over the 4 x 4 texels around the sample) — the reference's DEFAULT
// filter (texture2d_storage.hpp:73); negative lobes are clamped by the caller like every other filter's result.  The reference filters the four rows
// along x and then the results along y with p1 + x/2 (p2 - p0) + x^2/2 (2 p0 - 5 p1 + 4 p2 - p3) + x^3/2 (-p0 + 3 p1 - 3 p2 + p3); here the same
// polynomial as a weight per texel, w(x) x w(y), accumulated in ONE loop the compiler must not unroll around ONE texel fetch (equal to the nested
// form up to float rounding, tests/test_textures.py).  Why: sixteen inlined fetches with their wrap logic at every texture lookup site of every
// kernel multiplied the device code's compile time by eight; a non-inlined function, or row buffers in local arrays, cost every kernel that can
// meet a texture 50-120 spilled registers whether or not the scene has a bicubic bitmap (profiles/r06_ab_experiments.log).
WT_HD float tex_cubic_weight(float x, int i) {
    const float x2 = x * x, x3 = x2 * x;
    const float w0 = .5f * (-x + 2.f * x2 - x3), w1 = .5f * (2.f - 5.f * x2 + 3.f * x3), w2 = .5f * (x + 4.f * x2 - 3.f * x3), w3 = .5f * (-x2 + x3);
    return i == 0 ? w0 : (i == 1 ? w1 : (i == 2 ? w2 : w3));
}
WT_HD rgba_t tex_bicubic(const scene_t& sc, const texture_t& t, float u, float v) {
    const float fu = floorf(u), fv = floorf(v);
    const int iu = (int)fu, iv = (int)fv;
    const float fx = u - fu, fy = v - fv;
    rgba_t acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        const int x = i & 3, y = i >> 2;
        const float w = tex_cubic_weight(fx, x) * tex_cubic_weight(fy, y);
        const rgba_t q = tex_texel(sc, t, iu + x - 1, iv + y - 1);
        acc.r += w * q.r;
        acc.g += w * q.g;
        acc.b += w * q.b;
        acc.a += w * q.a;
    }
    return acc;
}
WT_HD rgba_t tex_bitmap(const scene_t& sc, const texture_t& t, vec2 uv) {
    uv.y = 1.f - uv.y;
    const float u = float(t.width) * uv.x - .5f, v = float(t.height) * uv.y - .5f;
    rgba_t r;
    if (!t.bilinear) {
        r = tex_texel(sc, t, (int)roundf(u), (int)roundf(v));
#if !defined(WT_NO_BICUBIC)
    } else if (t.bilinear == 2u) {
        r = tex_bicubic(sc, t, u, v);
#endif
    } else {
        const float fu = floorf(u), fv = floorf(v);
        const int iu = (int)fu, iv = (int)fv;
        const float fx = u - fu, fy = v - fv;
        r = tex_mix(tex_mix(tex_texel(sc, t, iu, iv), tex_texel(sc, t, iu + 1, iv), fx), tex_mix(tex_texel(sc, t, iu, iv + 1), tex_texel(sc, t, iu + 1, iv + 1), fx), fy);
    }
    return {fmaxf_(0.f, r.r), fmaxf_(0.f, r.g), fmaxf_(0.f, r.b), fmaxf_(0.f, r.a)};
}
// texture_t::get_RGBA.  Checkerboards nest (checkerboard.hpp:72-79: the parity of the integer parts of u and v picks the nested
// texture; its uv is the same query), at most 4 levels here.
WT_HD rgba_t texture_rgba(const scene_t& sc, int id, vec2 uv, bool* is_rgb = nullptr) {
    float scale = 1.f;
    if (is_rgb) *is_rgb = false;
    for (int depth = 0; depth < 4; ++depth) {
        const texture_t t = sc.textures[id];
        uv = vec2{t.m[0] * uv.x + t.m[1] * uv.y + t.t[0], t.m[2] * uv.x + t.m[3] * uv.y + t.t[1]};
        scale *= t.scale;
        if (t.type == TEX_CHECKERBOARD) {
            const int x = 2 * tex_modulo((int)uv.x, 2) - 1, y = 2 * tex_modulo((int)uv.y, 2) - 1;
            id = x * y == 1 ? t.col1 : t.col2;
            continue;
        }
        const rgba_t c = t.type == TEX_BITMAP ? tex_bitmap(sc, t, uv) : rgba_t{t.rgba[0], t.rgba[1], t.rgba[2], t.rgba[3]};
        if (is_rgb) *is_rgb = t.type == TEX_BITMAP && t.channels >= 3;
        return {c.r * scale, c.g * scale, c.b * scale, c.a};
    }
    return {0.f, 0.f, 0.f, 1.f};
}
// texture_t::f(query).x for luminance textures
WT_HD float texture_f(const scene_t& sc, int id, vec2 uv) { return texture_rgba(sc, id, uv).r; }

// colourspace::RGB_to_spectral::uplift (include/wt/spectrum/colourspace/RGB/RGB_to_spectral.hpp:27-84, the Smits-style uplift with ten
// 34-nm bins over 380..720 nm, 0 outside): the spectral value at wavenumber k [1/mm] of an RGB triple
WT_HD __attribute__((noinline)) float rgb_uplift(float r, float g, float b, float k) {
    static constexpr float white_I[10] = {1.0000f, 1.0000f, 0.9999f, 0.9993f, 0.9992f, 0.9998f, 1.0000f, 1.0000f, 1.0000f, 1.0000f};
    static constexpr float cyan_I[10] = {0.9710f, 0.9426f, 1.0007f, 1.0007f, 1.0007f, 1.0007f, 0.1564f, 0.0000f, 0.0000f, 0.0000f};
    static constexpr float magenta_I[10] = {1.0000f, 1.0000f, 0.968f, 0.22295f, 0.0000f, 0.0458f, 0.8369f, 1.0000f, 1.0000f, 0.9959f};
    static constexpr float yellow_I[10] = {0.0001f, 0.0000f, 0.1088f, 0.6651f, 1.0000f, 1.0000f, 0.9996f, 0.9586f, 0.9685f, 0.9840f};
    static constexpr float red_I[10] = {0.1012f, 0.0515f, 0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.8325f, 1.0149f, 1.0149f, 1.014f};
    static constexpr float green_I[10] = {0.0000f, 0.0000f, 0.0273f, 0.7937f, 1.0000f, 0.9418f, 0.1719f, 0.0000f, 0.0000f, 0.0025f};
    static constexpr float blue_I[10] = {1.0000f, 1.0000f, 0.8916f, 0.3323f, 0.0000f, 0.0000f, 0.0003f, 0.0369f, 0.0483f, 0.0496f};
    const float lambda_nm = 6.2831853f / k * 1e6f;
    if (!(lambda_nm >= 380.f && lambda_nm <= 720.f)) return 0.f;
    const int bin = (int)((lambda_nm - 380.f) / (720.f - 380.f) * 10.f);
    if (bin > 9) return 0.f;   // lambda = 720 nm exactly: the reference's eleventh, empty bin
    float I = 0.f;
    if (r <= g && r <= b) {
        I += white_I[bin] * r;
        if (g <= b) {
            I += cyan_I[bin] * (g - r);
            I += blue_I[bin] * (b - g);
        } else {
            I += cyan_I[bin] * (b - r);
            I += green_I[bin] * (g - b);
        }
    } else if (g <= r && g <= b) {
        I += white_I[bin] * g;
        if (r <= b) {
            I += magenta_I[bin] * (r - g);
            I += blue_I[bin] * (b - r);
        } else {
            I += magenta_I[bin] * (b - g);
            I += red_I[bin] * (r - b);
        }
    } else {
        I += white_I[bin] * b;
        if (r <= g) {
            I += yellow_I[bin] * (r - b);
            I += green_I[bin] * (g - r);
        } else {
            I += yellow_I[bin] * (g - b);
            I += red_I[bin] * (r - g);
        }
    }
    return I;
}
// texture_t::f(query).x at wavenumber k of a constant / checkerboard / bitmap texture: luminance textures are wavelength independent, RGB
// bitmaps are uplifted per lookup (bitmap.hpp:125-140)
WT_HD float texture_spectral_leaf(const scene_t& sc, int id, vec2 uv, float k) {
    bool rgb = false;
    const rgba_t c = texture_rgba(sc, id, uv, &rgb);
    return rgb ? rgb_uplift(c.r, c.g, c.b, k) : c.r;
}
// function_t::f / mix_t::f (texture/function.hpp:120-125, src/texture/function.cpp:92-107; texture/mix.hpp:96-106): the host compiled the
// expression — variables: the nested textures by name, u, v and k [1/mm] — into a postfix program; nested function textures are inlined, so a
// TOP_TEX operand is always a constant / checkerboard / bitmap texture, looked up at the SAME query (its own transform applies on top).
// Kept out of line: the kernels that never meet one keep their register allocation.
WT_HD __attribute__((noinline)) float texture_function(const scene_t& sc, uint32_t offset, uint32_t len, vec2 uv, float k) {
    float st[12];
    int sp = 0;
    const float* prog = sc.texture_data + offset;
    for (uint32_t i = 0; i + 1 < len; i += 2) {
        const int op = (int)prog[i];
        const float arg = prog[i + 1];
        if (op <= TOP_TEX) {   // operands
            if (sp >= 12) return 0.f;
            st[sp++] = op == TOP_CONST ? arg : op == TOP_U ? uv.x : op == TOP_V ? uv.y : op == TOP_K ? k : texture_spectral_leaf(sc, (int)arg, uv, k);
            continue;
        }
        const bool unary = op == TOP_NEG || (op >= TOP_ABS && op <= TOP_ATAN) || op == TOP_NOT;
        const int need = unary ? 1 : (op == TOP_MIX ? 3 : 2);
        if (sp < need) return 0.f;
        const float c = st[sp - 1], b = need >= 2 ? st[sp - 2] : 0.f, a = need >= 3 ? st[sp - 3] : 0.f;
        sp -= need;
        float r = 0.f;
        switch (op) {
        case TOP_ADD: r = b + c; break;
        case TOP_SUB: r = b - c; break;
        case TOP_MUL: r = b * c; break;
        case TOP_DIV: r = b / c; break;
        case TOP_NEG: r = -c; break;
        case TOP_POW: r = powf(b, c); break;
        case TOP_MIN: r = b < c ? b : c; break;
        case TOP_MAX: r = b > c ? b : c; break;
        case TOP_ABS: r = fabsf(c); break;
        case TOP_SQRT: r = sqrtf(c); break;
        case TOP_SIN: r = sinf(c); break;
        case TOP_COS: r = cosf(c); break;
        case TOP_TAN: r = tanf(c); break;
        case TOP_EXP: r = expf(c); break;
        case TOP_LOG: r = logf(c); break;
        case TOP_FLOOR: r = floorf(c); break;
        case TOP_CEIL: r = ceilf(c); break;
        case TOP_ROUND: r = roundf(c); break;
        case TOP_ASIN: r = asinf(c); break;
        case TOP_ACOS: r = acosf(c); break;
        case TOP_ATAN: r = atanf(c); break;
        case TOP_ATAN2: r = atan2f(b, c); break;
        case TOP_MIX: r = c == 0.f ? a : (c == 1.f ? b : a * (1.f - c) + b * c); break;   // mix(texture1, texture2, m): m::mix
        case TOP_LT: r = b < c ? 1.f : 0.f; break;
        case TOP_LE: r = b <= c ? 1.f : 0.f; break;
        case TOP_GT: r = b > c ? 1.f : 0.f; break;
        case TOP_GE: r = b >= c ? 1.f : 0.f; break;
        case TOP_EQ: r = b == c ? 1.f : 0.f; break;
        case TOP_NE: r = b != c ? 1.f : 0.f; break;
        case TOP_AND: r = (b != 0.f && c != 0.f) ? 1.f : 0.f; break;
        case TOP_OR: r = (b != 0.f || c != 0.f) ? 1.f : 0.f; break;
        case TOP_NOT: r = c == 0.f ? 1.f : 0.f; break;
        default: return 0.f;
        }
        st[sp++] = r;
    }
    return sp == 1 ? st[0] : 0.f;
}
// texture_t::f(query).x at wavenumber k
WT_HD float texture_spectral(const scene_t& sc, int id, vec2 uv, float k) {
    if (sc.textures[id].type != TEX_FUNCTION) return texture_spectral_leaf(sc, id, uv, k);
    const texture_t t = sc.textures[id];
    uv = vec2{t.m[0] * uv.x + t.m[1] * uv.y + t.t[0], t.m[2] * uv.x + t.m[3] * uv.y + t.t[1]};
    return t.scale * texture_function(sc, t.offset, t.width, uv, k);
}

}   // namespace wt
