// wave_tracer_amd — beams (elliptic-cone envelope + polarimetric radiometric payload), surface intersection
// records and beam sourcing geometry.
//
// Reference: include/wt/beam/beam_generic.hpp:38-194, include/wt/beam/beam.hpp:27-603,
//            include/wt/beam/beam_geometry.hpp:32-342, include/wt/beam/gaussian_wavefront.hpp:22-119,
//            include/wt/interaction/intersection.hpp:34-200, src/interaction/intersection.cpp:33-185,
//            include/wt/interaction/common.hpp:19-44.
#pragma once
#include "cone.h"
#include "polar.h"
#include "scene.h"

namespace wt {

constexpr float kBeamEnvelope = 3.f;   // gaussian_wavefront.hpp:25 (envelope = 3 sigma)
constexpr float kMubSbp = 0.25f;       // beam_geometry.hpp:38

enum transport_e : uint32_t { TRANSPORT_FORWARD = 0, TRANSPORT_BACKWARD = 1 };
WT_HD uint32_t flip_transport(uint32_t t) { return t == TRANSPORT_FORWARD ? TRANSPORT_BACKWARD : TRANSPORT_FORWARD; }

// ---- surface intersection record (intersection_surface_t) ------------------------------------------
struct footprint_t {
    vec2 x;   // major axis direction in the geo tangent frame
    float la, lb;
};
struct surface_t {
    vec3 wp;
    vec2 uv;
    vec2 bary;
    footprint_t footprint;
    uint32_t tuid;    // ADS triangle (kInvalid: dummy surface w/o shape, e.g. virtual sensors)
    uint32_t shape;   // kInvalid for dummy
    frame_t geo, shading;
};

// dummy surface (intersection.hpp:100-105)
WT_HD surface_t make_dummy_surface(vec3 n, vec3 p) {
    surface_t s;
    s.wp = p;
    s.uv = {0, 0};
    s.bary = {-1, -1};
    s.footprint = {{1, 0}, 0.f, 0.f};
    s.tuid = kInvalid;
    s.shape = kInvalid;
    s.geo = build_orthogonal_frame(n);
    s.shading = s.geo;
    return s;
}
// intersection_surface_t(shape, geo_n, mesh_tri_idx, bary, centre) (intersection.cpp:33-72).  No BSDF in the
// supported set perturbs the shading frame, so shading = build_shading_frame(interpolated n, dpdu).
WT_HD surface_t make_surface(const scene_t& sc, uint32_t tuid, vec3 geo_n, vec2 bary, vec3 centre) {
    const tri_shade_t sh = sc.tri_shade[tuid];
    const float bz = 1.f - bary.x - bary.y;
    surface_t s;
    s.wp = centre;
    s.bary = bary;
    s.uv = sh.has_uv ? sh.uv0 * bary.x + sh.uv1 * bary.y + sh.uv2 * bz : vec2{0.f, 0.f};
    s.footprint = {{1, 0}, 0.f, 0.f};
    s.tuid = tuid;
    s.shape = sc.tri_meta[tuid].shape_idx;
    const vec3 ns = normalize(sh.n0 * bary.x + sh.n1 * bary.y + sh.n2 * bz);
    s.geo = build_shading_frame(geo_n, sh.dpdu);
    s.shading = build_shading_frame(ns, sh.dpdu);
    if (sc.n_textures) {   // (scene-uniform: scenes without textures skip the lookups)
        // normalmap wrapper (bsdf/normalmap.hpp:48-62): the mapped normal, given in the shading frame, replaces the shading normal
        const material_t& m = sc.materials[sc.shapes[s.shape].material];
        const uint32_t nt = m.normal_tex;
        if (nt) {
            const rgba_t c = texture_rgba(sc, (int)nt - 1, s.uv);
            const float sg = m.normal_flip ? -1.f : 1.f;
            const vec3 n = normalize(vec3{(c.r * 2.f - 1.f) * sg, (c.g * 2.f - 1.f) * sg, c.b * 2.f - 1.f});
            s.shading = build_shading_frame(to_world(s.shading, n), sh.dpdu);
        }
    }
    return s;
}
// intersection_surface_t(shape, mesh_tri_idx, bary): centre at the barycentric point (intersection.cpp:74-82)
WT_HD surface_t make_surface_at_bary(const scene_t& sc, uint32_t tuid, vec2 bary) {
    const tri_geo_t g = sc.tri_geo[tuid];
    const vec3 p = g.a * bary.x + g.b * bary.y + g.c * (1.f - bary.x - bary.y);
    return make_surface(sc, tuid, g.n, bary, p);
}

// s-polarisation direction and sp frame (intersection.hpp:117-141)
WT_HD vec3 surface_s_direction(const surface_t& s, vec3 w) {
    const vec3 crs = cross(w, s.shading.n);
    const float l2 = length2(crs);
    const vec3 ret = l2 < 1e-14f ? s.shading.t : crs / sqrtf(l2);
    return dot(w, s.shading.n) < 0.f ? -ret : ret;
}
WT_HD frame_t surface_sp_frame(const surface_t& s, vec3 w) {
    const vec3 sd = surface_s_direction(s, w);
    const vec3 p = cross(sd, w);
    return frame_t{sd, dot(w, s.shading.n) < 0.f ? -p : p, w};
}

// self-intersection offset (intersection.cpp:148-185)
WT_HD vec3 triangle_fp_errors(vec3 a, vec3 b, vec3 c, vec3 ro) {
    const float c0 = 3e-6f, c1 = 5e-6f, c2 = 3e-6f;
    const vec3 v0 = vabs(a);
    const vec3 e1 = vabs(b - a), e2 = vabs(c - a);
    const vec3 extents = e1 + e2 + vabs(e1 - e2);
    const float extent = max_element(extents);
    const vec3 obj_err = (c0 + c2) * v0 + vec3{c1 * extent, c1 * extent, c1 * extent};
    const vec3 wrld_err = (c1 + c2) * vabs(ro);
    return obj_err + wrld_err;
}
WT_HD vec3 surface_offseted_ray_origin(const scene_t& sc, const surface_t& s, vec3 ro, vec3 rd) {
    if (s.tuid == kInvalid) return ro;
    const tri_geo_t g = sc.tri_geo[s.tuid];
    const vec3 err = triangle_fp_errors(g.a, g.b, g.c, ro);
    const vec3 ng = s.geo.n;
    const float offset_dist = dot(err, vabs(ng));
    const vec3 offset = offset_dist * ng;
    return ro + (dot(rd, offset) >= 0.f ? offset : -offset);
}

// ---- phase-space extent / sourcing geometry (beam_geometry.hpp) --------------------------------------
struct phase_space_extent_t {
    float spatial_extent;   // area [m^2]
    float tan_alpha;
    float k;
};
WT_HD phase_space_extent_t pse_enlarge(const phase_space_extent_t& e, float scale) {
    if (scale == 1.f) return e;
    return {e.spatial_extent * sqr(scale), e.tan_alpha * scale, e.k};
}
// minimum_uncertainty_tan_alpha(Length) (beam_geometry.hpp:108-120)
WT_HD float mub_tan_alpha_from_length(float len_m, float k) {
    return len_m > 0.f ? sqrtf(kMubSbp) * sqr(kBeamEnvelope) / k_times_len(k, len_m) : 0.f;
}
// minimum_uncertainty_spatial_extent(tan_alpha) -> spatial *length* (beam_geometry.hpp:165-179)
WT_HD float mub_spatial_length_from_tan_alpha(float tan_alpha, float k) {
    return tan_alpha > 0.f ? sqrtf(kMubSbp) * sqr(kBeamEnvelope) / (k * 1000.f * tan_alpha) : 0.f;
}
struct sourcing_geometry_t {
    vec3 x;
    vec2 initial_spatial_lengths;
    float tan_alpha;
    uint32_t has_surface;
    vec3 surface_n;   // geo.n of the sourcing surface
    float k;
};
WT_HD phase_space_extent_t sg_phase_space_extent(const sourcing_geometry_t& g) {
    return {g.initial_spatial_lengths.x * g.initial_spatial_lengths.y, g.tan_alpha, g.k};
}
// sourcing_geometry_t::source(length, tan_alpha, k)
WT_HD sourcing_geometry_t sg_source(float len, float tan_alpha, float k) {
    return {{1, 0, 0}, {len, len}, tan_alpha, 0, {0, 0, 1}, k};
}
// sourcing_geometry_t::source(extent)
WT_HD sourcing_geometry_t sg_source(const phase_space_extent_t& e) {
    const float l = sqrtf(e.spatial_extent);
    return {{1, 0, 0}, {l, l}, e.tan_alpha, 0, {0, 0, 1}, e.k};
}
// sourcing_geometry_t::source_mub_from(length, k)
WT_HD sourcing_geometry_t sg_source_mub_from_length(float len, float k) {
    return {{1, 0, 0}, {len, len}, mub_tan_alpha_from_length(len, k), 0, {0, 0, 1}, k};
}
// sourcing_geometry_t::envelope (beam_geometry.hpp:201-224)
WT_HD cone_t sg_envelope(const sourcing_geometry_t& g, vec3 ro, vec3 rd, float& self_intersection_distance) {
    if (g.has_surface) {
        const vec3 X = g.x * g.initial_spatial_lengths.x;
        const vec3 Y = cross(g.x, g.surface_n) * g.initial_spatial_lengths.y;
        return cone_through_ellipse(X, Y, g.surface_n, ro, rd, g.tan_alpha, &self_intersection_distance);
    }
    self_intersection_distance = 0.f;
    if (g.initial_spatial_lengths.x != g.initial_spatial_lengths.y) {
        const float ix = fmaxf_(g.initial_spatial_lengths.x, g.initial_spatial_lengths.y);
        const float e = fminf_(g.initial_spatial_lengths.x, g.initial_spatial_lengths.y) / ix;
        return make_cone(ro, rd, g.x, g.tan_alpha, e, ix);
    }
    return make_cone_iso(ro, rd, g.tan_alpha, g.initial_spatial_lengths.x);
}

// ---- beam ------------------------------------------------------------------------------------------
// forward transport : rad[0..3] = Stokes vector S, `frame` = frame of S, scale unused (=1)
// backward transport: rad[0..15] = Mueller operator M (row-major), `frame` = incident frame of M, `scale`
struct beam_t {
    cone_t env;
    float k;
    float self_intersection_distance;
    uint32_t transport;
    frame_t frame;
    float scale;
    float rad[16];
};

WT_HD mueller_t beam_M(const beam_t& b) {
    mueller_t M;
    for (int i = 0; i < 16; ++i) M.m[i] = b.rad[i];
    return M;
}
WT_HD void beam_set_M(beam_t& b, const mueller_t& M) {
    for (int i = 0; i < 16; ++i) b.rad[i] = M.m[i];
}
WT_HD stokes_t beam_S(const beam_t& b) { return {{b.rad[0], b.rad[1], b.rad[2], b.rad[3]}}; }
WT_HD void beam_set_S(beam_t& b, const stokes_t& S) {
    for (int i = 0; i < 4; ++i) b.rad[i] = S.s[i];
    for (int i = 4; i < 16; ++i) b.rad[i] = 0.f;
}
WT_HD float beam_intensity(const beam_t& b) { return b.transport == TRANSPORT_FORWARD ? b.rad[0] : b.rad[0] * b.scale; }
WT_HD void beam_scale(beam_t& b, float f) {
    if (b.transport == TRANSPORT_FORWARD) {
        b.rad[0] *= f;
        b.rad[1] *= f;
        b.rad[2] *= f;
        b.rad[3] *= f;
    } else
        b.scale *= f;
}
WT_HD bool beam_finite(const beam_t& b) {
    if (b.transport == TRANSPORT_FORWARD) return finitef(b.rad[0]) && finitef(b.rad[1]) && finitef(b.rad[2]) && finitef(b.rad[3]);
    for (int i = 0; i < 16; ++i)
        if (!finitef(b.rad[i])) return false;
    return finitef(b.scale);
}
WT_HD vec3 beam_dir(const beam_t& b) { return b.env.d; }
WT_HD vec3 beam_origin(const beam_t& b) { return b.env.o; }
WT_HD bool beam_is_ray(const beam_t& b) { return cone_is_ray(b.env); }

// forward beam from (ray, unpolarised intensity, k, sourcing geometry) (beam.hpp:283-291)
WT_HD beam_t make_forward_beam(vec3 ro, vec3 rd, float I, float k, const sourcing_geometry_t& sg) {
    beam_t b;
    b.env = sg_envelope(sg, ro, rd, b.self_intersection_distance);
    b.k = k;
    b.transport = TRANSPORT_FORWARD;
    b.frame = cone_frame(b.env);
    b.scale = 1.f;
    beam_set_S(b, stokes_unpolarized(I));
    return b;
}
// backward beam from (ray, scale, k, sourcing geometry) (beam.hpp:326-334): M = identity
WT_HD beam_t make_backward_beam(vec3 ro, vec3 rd, float scale, float k, const sourcing_geometry_t& sg) {
    beam_t b;
    b.env = sg_envelope(sg, ro, rd, b.self_intersection_distance);
    b.k = k;
    b.transport = TRANSPORT_BACKWARD;
    b.frame = cone_frame(b.env);
    b.scale = scale;
    beam_set_M(b, mueller_identity());
    return b;
}

// beam_generic_t::footprint / std_dev (beam_generic.hpp:102-114)
WT_HD vec3 beam_footprint(const beam_t& b, float dist) {
    const vec2 a = cone_axes(b.env, dist);
    return vec3{a.x, a.y, kMajorAxisToZScale * a.x};
}

// data.apply_bsdf (beam.hpp:56-70 forward, 168-181 backward)
WT_HD void beam_apply_bsdf(beam_t& b, const mueller_t& op, vec3 wo, const surface_t& surface, const frame_t& frame_after) {
    if (b.transport == TRANSPORT_FORWARD) {
        const frame_t SPin = surface_sp_frame(surface, b.frame.n);
        const frame_t SPout = surface_sp_frame(surface, wo);
        const stokes_t S = mueller_apply(op, beam_S(b), b.frame, SPin, frame_after, SPout);
        beam_set_S(b, S);
        b.frame = frame_after;
    } else {
        const frame_t SPin = surface_sp_frame(surface, wo);
        const frame_t SPout = surface_sp_frame(surface, b.frame.n);
        beam_set_M(b, mueller_compose(beam_M(b), op, b.frame, SPout));
        b.frame = SPin;
    }
}

// cone_through_ellipse(surface, ray, tan_alpha, &sid) (elliptic_cone.hpp:272-291)
WT_HD cone_t cone_through_surface_footprint(const surface_t& s, vec3 ro, vec3 rd, float tan_alpha, float* sid) {
    const vec2 a = s.footprint.x * s.footprint.la;
    const vec2 bb = vec2{-s.footprint.x.y, s.footprint.x.x} * s.footprint.lb;
    const vec3 wa = to_world(s.geo, a), wb = to_world(s.geo, bb);
    return cone_through_ellipse(wa, wb, s.geo.n, ro, rd, tan_alpha, sid);
}

// beam_t::transform_surface_interaction (beam.hpp:379-398)
WT_HD void beam_transform_surface_interaction(beam_t& b, const surface_t& surface, vec3 wo, const mueller_t& bsdfM, float weight) {
    float sid;
    b.env = cone_through_surface_footprint(surface, surface.wp, wo, b.env.tan_alpha, &sid);
    beam_apply_bsdf(b, weight * bsdfM, wo, surface, cone_frame(b.env));
    b.self_intersection_distance = sid;
}
// beam_t::transform_region_interaction (beam.hpp:409-428)
WT_HD void beam_transform_region_interaction(beam_t& b, vec3 wp, float dist, vec3 wo, float weight) {
    const vec3 axes_local = beam_footprint(b, dist);
    b.env = cone_through_ellipsoid(axes_local, cone_frame(b.env), wp, wo, b.env.tan_alpha);
    beam_scale(b, weight);
    b.frame = cone_frame(b.env);
    b.self_intersection_distance = 0.f;
}
// beam_t::transform_restart (beam.hpp:465-471)
WT_HD void beam_transform_restart(beam_t& b, vec3 wp, float dist) {
    b.env.o = wp;
    cone_set_x0(b.env, b.env.x0 + dist * b.env.tan_alpha);
    b.self_intersection_distance = 0.f;
}

// beam_generic_t::surface_footprint_static (beam_generic.hpp:165-190)
// (of the beam's envelope alone: all it reads of the beam)
WT_HD footprint_t cone_surface_footprint_static(const cone_t& env, const surface_t& surface, float beam_z_dist) {
    const vec2 a = cone_axes(env, beam_z_dist);
    const vec3 ls{a.x, a.y, kMajorAxisToZScale * a.x};
    const vec3 x = to_local(surface.geo, env.x);
    if (x.x != 0.f || x.y != 0.f) return {normalize(vec2{x.x, x.y}), ls.x, ls.y};
    const float avg = (ls.x + ls.y) / 2.f;
    return {{1.f, 0.f}, avg, avg};
}
WT_HD footprint_t beam_surface_footprint_static(const beam_t& b, const surface_t& surface, float beam_z_dist) {
    const vec3 ls = beam_footprint(b, beam_z_dist);
    const vec3 x = to_local(surface.geo, b.env.x);
    if (x.x != 0.f || x.y != 0.f) return {normalize(vec2{x.x, x.y}), ls.x, ls.y};
    const float avg = (ls.x + ls.y) / 2.f;
    return {{1.f, 0.f}, avg, avg};
}

// integrate_beams (beam.hpp:562-603): detector (backward) beam over radiation (forward) beam
WT_HD stokes_t integrate_beams(const beam_t& S, const beam_t& I) {
    if (beam_intensity(S) == 0.f || beam_intensity(I) == 0.f) return stokes_zero();
    return mueller_apply(beam_M(S), beam_S(I), I.frame, S.frame) * S.scale;
}

}   // namespace wt
