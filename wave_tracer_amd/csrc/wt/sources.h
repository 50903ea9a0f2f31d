// wave_tracer_amd — emitters, sensors and scene-level emitter/spectrum sampling.
//
// Reference: include/wt/emitter/spot.hpp:25-235 + src/emitter/spot.cpp:29-69,
//            include/wt/emitter/area.hpp:60-254 + src/emitter/area.cpp:35-160, src/scene/shape.cpp (sample_position),
//            include/wt/sensor/sensor/perspective.hpp:52-330,
//            include/wt/sensor/sensor/virtual_plane_sensor.hpp + src/sensor/virtual_plane_sensor.cpp:31-187,
//            include/wt/scene/scene.hpp:96-229, src/scene/scene_sensor.cpp:19-59.
#pragma once
#include "bsdf.h"

namespace wt {

struct emitter_sample_t {
    beam_t beam;
    float ppd, dpd;   // tagged
    uint32_t has_surface;
    surface_t surface;
};
struct emitter_direct_sample_t {
    int32_t emitter;
    float emitter_pdf;
    float dpd;   // tagged
    beam_t beam;
    uint32_t has_surface;
    surface_t surface;
};
struct sensor_element_t {
    uint32_t x, y;
    vec2 offset;
};
struct sensor_sample_t {
    beam_t beam;
    float ppd, dpd;   // tagged
    sensor_element_t element;
    uint32_t has_surface;
    surface_t surface;
};
struct sensor_direct_sample_t {
    beam_t beam;
    float dpd;   // tagged
    sensor_element_t element;
    uint32_t has_surface;
    surface_t surface;
};

// ================================ emitters ==========================================================
WT_HD bool emitter_is_area(const emitter_t& e) { return e.type == EMIT_AREA; }
WT_HD bool emitter_is_delta_position(const emitter_t& e) { return e.type == EMIT_SPOT || e.type == EMIT_POINT; }   // directional: false
WT_HD bool emitter_is_delta_direction(const emitter_t& e) { return e.type == EMIT_DIRECTIONAL; }
WT_HD bool emitter_is_infinite(const emitter_t& e) { return e.type == EMIT_DIRECTIONAL; }

// spot_t::compute_falloff (spot.hpp:65-70)
WT_HD float spot_falloff(const emitter_t& e, vec3 local_dir) {
    const float cos_theta = local_dir.z;
    if (cos_theta <= e.cos_cutoff) return 0.f;
    if (cos_theta >= e.cos_falloff) return 1.f;
    return (e.cutoff - acosf(cos_theta)) * e.recp_cutoff_range;
}
// spot_t::sourcing_geometry (spot.hpp:101-114)
WT_HD sourcing_geometry_t spot_sourcing_geometry(const emitter_t& e, float k) {
    const float initial_spatial_extent = e.extent > 0.f ? e.extent : 10.f * wavenum_to_wavelen_m(k);
    phase_space_extent_t se = pse_enlarge(sg_phase_space_extent(sg_source_mub_from_length(initial_spatial_extent, k)), e.phase_space_extent_scale);
    se.tan_alpha = fminf_(se.tan_alpha, e.max_tan_alpha);
    return sg_source(se);
}
// area_t::sourcing_geometry (area.hpp:135-147)
WT_HD sourcing_geometry_t area_sourcing_geometry(const emitter_t& e, float k) {
    const float initial_spatial_extent = 10.f * wavenum_to_wavelen_m(k);
    const phase_space_extent_t se =
        pse_enlarge(sg_phase_space_extent(sg_source_mub_from_length(initial_spatial_extent, k)), e.phase_space_extent_scale);
    return sg_source(se);
}
// point_t::sourcing_geometry (point.hpp:74-86)
WT_HD sourcing_geometry_t point_sourcing_geometry(const emitter_t& e, float k) {
    const float initial_spatial_extent = e.extent > 0.f ? e.extent : 10.f * wavenum_to_wavelen_m(k);
    const phase_space_extent_t se =
        pse_enlarge(sg_phase_space_extent(sg_source_mub_from_length(initial_spatial_extent, k)), e.phase_space_extent_scale);
    return sg_source(se);
}
// directional_t::sourcing_geometry (directional.hpp:117-126): MUB sourced into the solid angle the emitter subtends at the target
WT_HD sourcing_geometry_t directional_sourcing_geometry(const emitter_t& e, float k) {
    const float len = mub_spatial_length_from_tan_alpha(e.tan_alpha_at_target, k);
    sourcing_geometry_t g = sg_source_mub_from_length(len, k);
    g.tan_alpha = e.tan_alpha_at_target;
    return sg_source(pse_enlarge(sg_phase_space_extent(g), e.phase_space_extent_scale));
}
// directional_t::pdf_target_position (directional.hpp:176-184)
WT_HD float directional_pdf_target_position(const emitter_t& e, vec3 wp) {
    const vec3 l = to_local(e.frame, wp - e.position);
    return l.x * l.x + l.y * l.y <= sqr(e.target_radius) ? 1.f / e.target_area : 0.f;
}
// sampler_t::uniform_sphere (sampler.hpp:147-152)
WT_HD vec3 uniform_sphere(vec2 u) {
    const float z = 1.f - 2.f * u.x;
    const float rr = sqrtf(fmaxf_(0.f, 1.f - sqr(z)));
    const float phi = kTwoPi * u.y;
    return vec3{rr * cosf(phi), rr * sinf(phi), z};
}
WT_HD float emitter_spectral_value(const scene_t& sc, const emitter_t& e, float k) { return spectrum_f(sc, e.spectrum, k) * e.scale; }

// area_t::spectral_radiance (area.hpp:103-116): scale x radiance texture at the surface's uv, or scale x the (average) spectrum
WT_HD float area_spectral_radiance(const scene_t& sc, const emitter_t& e, const surface_t& surface, float k) {
    if (e.radiance_tex > 0) return e.scale * texture_spectral_leaf(sc, e.radiance_tex - 1, surface.uv, k);
    return emitter_spectral_value(sc, e, k);
}
// area_t::Le (area.hpp:156-166): radiance * max(0, d.ng)
WT_HD beam_t area_Le(const scene_t& sc, const emitter_t& e, vec3 ro, vec3 rd, float k, const surface_t& surface) {
    const float I = area_spectral_radiance(sc, e, surface, k) * fmaxf_(0.f, dot(rd, surface.geo.n));
    return make_forward_beam(ro, rd, I, k, area_sourcing_geometry(e, k));
}

// shape_t::sample_position (src/scene/shape.cpp:58-73)
WT_HD surface_t shape_sample_position(const scene_t& sc, int shape_idx, sampler_t& sampler, float& ppd) {
    const shape_t sh = sc.shapes[shape_idx];
    const vec3 r = sampler_r3(sampler);
    const float* cdf = sc.shape_tri_cdf + sh.tri_offset + shape_idx;
    const uint32_t idx = cdf_icdf(cdf, sh.tri_count, r.z);
    const vec2 bary = uniform_triangle(vec2{r.x, r.y});
    ppd = sh.recp_surface_area;
    return make_surface_at_bary(sc, sc.shape_tri_tuid[sh.tri_offset + idx], bary);
}

// ---- textured area emitters: per-triangle texel tables (src/emitter/area.cpp:153-260) -------------------------------------------------
// texture_data[e.tab ..): the triangle distribution's normalised cdf (T + 1 knots, T = the shape's triangle count: discrete_distribution_t::
// dcdf), then 4 words per triangle {texels, 1 / texels, texel_to_area_density, offset of its cdf from e.tab}, then per triangle the cdf of its
// texels (texels (texels + 1) / 2 + 1 knots): row b = 0 .. texels - 1 holds the cells a = 0 .. b, cell (a, b) covering the barycentrics
// alpha in [a, a + 1) / texels, beta in 1 - (b, b + 1] / texels.  Integers are stored as floats (the host refuses tables beyond 2^24 words).
struct area_table_tri_t {
    uint32_t texels;
    float recp_texels, texel_to_area_density;
    const float* cdf;   // texels (texels + 1) / 2 + 1 knots
};
WT_HD area_table_tri_t area_table_tri(const scene_t& sc, const emitter_t& e, uint32_t tid) {
    const float* tab = sc.texture_data + e.tab;
    const float* h = tab + (sc.shapes[e.shape].tri_count + 1) + 4 * tid;
    return {(uint32_t)h[0], h[1], h[2], tab + (uint32_t)h[3]};
}
// sampling_data_t::sample (area.cpp:218-247)
WT_HD surface_t area_table_sample(const scene_t& sc, const emitter_t& e, sampler_t& sampler, float& ppd) {
    const float rx = sampler_r(sampler), ry = sampler_r(sampler), rz = sampler_r(sampler), rw = sampler_r(sampler);   // sampler.r4()
    const shape_t sh = sc.shapes[e.shape];
    const float* tcdf = sc.texture_data + e.tab;
    const uint32_t tid = cdf_icdf(tcdf, sh.tri_count, rx);
    const float tpdf = tcdf[tid + 1] - tcdf[tid];
    const area_table_tri_t t = area_table_tri(sc, e, tid);
    const uint32_t cell = cdf_icdf(t.cdf, t.texels * (t.texels + 1) / 2, ry);
    const float cell_pdf = t.cdf[cell + 1] - t.cdf[cell];
    const uint32_t b = (uint32_t)ceilf(-.5f + sqrtf(.25f + float(2 * (cell + 1)))) - 1;
    const uint32_t a = cell - b * (b + 1) / 2;
    const float alpha = clampf(float(a) * t.recp_texels + rz * t.recp_texels, 0.f, 1.f);
    const float beta = clampf(1.f - (float(b) * t.recp_texels + rw * t.recp_texels), 0.f, 1.f - alpha);
    ppd = tpdf * cell_pdf * t.texel_to_area_density;
    return make_surface_at_bary(sc, sc.shape_tri_tuid[sh.tri_offset + tid], vec2{alpha, beta});
}
// sampling_data_t::pdf (area.cpp:248-271).  The cell indices are clamped into the table (the reference indexes its vector with whatever the
// rounding gives: on a triangle's border that can be one past a row)
WT_HD float area_table_pdf(const scene_t& sc, const emitter_t& e, const surface_t& surface) {
    if (surface.shape != (uint32_t)e.shape) return 0.f;
    const uint32_t tid = sc.tri_meta[surface.tuid].shape_tri_idx;
    const float* tcdf = sc.texture_data + e.tab;
    const float tpdf = tcdf[tid + 1] - tcdf[tid];
    const area_table_tri_t t = area_table_tri(sc, e, tid);
    if (t.texels == 0) return 0.f;
    const float fa = roundf(surface.bary.x * float(t.texels) - .5f), fb = roundf((1.f - surface.bary.y) * float(t.texels) - .5f);
    const uint32_t b = (uint32_t)clampf(fb, 0.f, float(t.texels - 1));
    const uint32_t a = (uint32_t)clampf(fa, 0.f, float(b));
    const uint32_t cell = b * (b + 1) / 2 + a;
    return tpdf * (t.cdf[cell + 1] - t.cdf[cell]) * t.texel_to_area_density;
}
// area_t::sample_position (area.cpp:109-120)
WT_HD surface_t area_sample_position(const scene_t& sc, const emitter_t& e, sampler_t& sampler, float& ppd) {
    if (e.radiance_tex > 0) return area_table_sample(sc, e, sampler, ppd);
    return shape_sample_position(sc, e.shape, sampler, ppd);
}
// area_t::pdf_position (area.cpp:122-130)
WT_HD float area_pdf_position(const scene_t& sc, const emitter_t& e, const surface_t* surface) {
    if (e.radiance_tex > 0) return surface ? area_table_pdf(sc, e, *surface) : 0.f;
    return sc.shapes[e.shape].recp_surface_area;
}

// emitter_t::sample
WT_HD emitter_sample_t emitter_sample(const scene_t& sc, int ei, float k, sampler_t& sampler) {
    const emitter_t e = sc.emitters[ei];
    emitter_sample_t r;
    r.has_surface = 0;
    if (e.type == EMIT_SPOT) {
        const float cutoff_sa = kTwoPi * (1.f - e.cos_cutoff);
        const vec3 local_wo = uniform_cone(cutoff_sa, sampler_r2(sampler));
        const vec3 wo = to_world(e.frame, local_wo);
        const float w = spot_falloff(e, local_wo);
        const float dpd = uniform_cone_pdf(cutoff_sa);
        r.beam = make_forward_beam(e.position, wo, emitter_spectral_value(sc, e, k), k, spot_sourcing_geometry(e, k));
        beam_scale(r.beam, w / dpd);
        r.ppd = pd_discrete(1.f);
        r.dpd = dpd;
    } else if (e.type == EMIT_DIRECTIONAL) {
        // directional_t::sample (src/emitter/directional.cpp:29-50): a point of the target disk, sourced from beyond the world
        const vec2 p = concentric_disk(sampler_r2(sampler)) * e.target_radius;
        const vec3 wp = e.position + to_world(e.frame, p);
        r.beam = make_forward_beam(wp + e.far_dist * e.frame.n, -e.frame.n, emitter_spectral_value(sc, e, k), k, directional_sourcing_geometry(e, k));
        beam_scale(r.beam, e.target_area);
        r.ppd = 1.f / e.target_area;
        r.dpd = pd_discrete(1.f);
    } else if (e.type == EMIT_POINT) {
        // point_t::sample (src/emitter/point.cpp:28-43)
        const vec3 d = uniform_sphere(sampler_r2(sampler));
        r.beam = make_forward_beam(e.position, d, emitter_spectral_value(sc, e, k), k, point_sourcing_geometry(e, k));
        beam_scale(r.beam, 4.f * kPi);
        r.ppd = pd_discrete(1.f);
        r.dpd = kInvTwoPi * .5f;
    } else {
        float ppd;
        r.surface = area_sample_position(sc, e, sampler, ppd);
        r.has_surface = 1;
        vec3 d = cosine_hemisphere(sampler_r2(sampler));
        const float dn = d.z;
        d = to_world(r.surface.geo, d);
        const float dpd = cosine_hemisphere_pdf(dn);
        float recp_pdf = 1.f / (dpd * ppd);
        if (dpd * ppd == 0.f) recp_pdf = 0.f;
        r.beam = area_Le(sc, e, r.surface.wp, d, k, r.surface);
        beam_scale(r.beam, recp_pdf);
        r.ppd = ppd;
        r.dpd = dpd;
    }
    return r;
}
// emitter_t::pdf_position (the surface: the point's, for an area emitter with a radiance texture)
WT_HD float emitter_pdf_position(const scene_t& sc, int ei, const surface_t* surface = nullptr) {
    const emitter_t e = sc.emitters[ei];
    if (e.type == EMIT_SPOT || e.type == EMIT_POINT) return pd_discrete(1.f);
    if (e.type == EMIT_DIRECTIONAL) return 0.f;   // infinite emitters have no position density (vertex.hpp:557-558)
    return area_pdf_position(sc, e, surface);
}
// emitter_t::pdf_direction (solid-angle density)
WT_HD float emitter_pdf_direction(const scene_t& sc, int ei, vec3 dir, const surface_t* surface) {
    const emitter_t e = sc.emitters[ei];
    if (e.type == EMIT_SPOT) return uniform_cone_pdf(kTwoPi * (1.f - e.cos_cutoff));
    if (e.type == EMIT_POINT) return kInvTwoPi * .5f;   // point.hpp pdf_direction: uniform sphere
    if (e.type == EMIT_DIRECTIONAL) return pd_discrete(veq(dir, -e.frame.n) ? 1.f : 0.f);   // directional.hpp:192-199
    const float dn = fmaxf_(0.f, dot(dir, surface->geo.n));
    return cosine_hemisphere_pdf(dn);
}
// area_t::pdf_direct (area.cpp:127-135)
WT_HD float area_pdf_direct(const scene_t& sc, const emitter_t& e, vec3 wp, vec3 ro, vec3 rd, const surface_t& surface) {
    const float ppd = area_pdf_position(sc, e, &surface);
    const float l2 = length2(wp - ro);
    const float dn = fmaxf_(0.f, dot(rd, surface.geo.n));
    const float recp_dn = dn > 0.f ? 1.f / dn : 0.f;
    return ppd * l2 * recp_dn;
}
// emitter_t::sample_direct
WT_HD emitter_direct_sample_t emitter_sample_direct(const scene_t& sc, int ei, vec3 wp, float k, sampler_t& sampler) {
    const emitter_t e = sc.emitters[ei];
    emitter_direct_sample_t r;
    r.emitter = ei;
    r.emitter_pdf = 0.f;
    r.has_surface = 0;
    if (e.type == EMIT_SPOT) {
        const vec3 dl = wp - e.position;
        const float recp_dist2 = 1.f / length2(dl);
        const vec3 d = dl * sqrtf(recp_dist2);
        const vec3 local_wo = to_local(e.frame, d);
        const float w = spot_falloff(e, local_wo);
        r.beam = make_forward_beam(e.position, d, emitter_spectral_value(sc, e, k), k, spot_sourcing_geometry(e, k));
        beam_scale(r.beam, w * recp_dist2);
        r.dpd = pd_discrete(1.f);
    } else if (e.type == EMIT_DIRECTIONAL) {
        // directional_t::sample_direct (src/emitter/directional.cpp:52-73)
        const vec3 l = to_local(e.frame, wp - e.position);
        const vec2 p{l.x, l.y};
        const float scale = dot(p, p) <= sqr(e.target_radius) ? 1.f : 0.f;
        const vec3 targetwp = e.position + to_world(e.frame, p);
        r.beam = make_forward_beam(targetwp + e.far_dist * e.frame.n, -e.frame.n, emitter_spectral_value(sc, e, k), k, directional_sourcing_geometry(e, k));
        beam_scale(r.beam, scale);
        r.dpd = pd_discrete(1.f);
    } else if (e.type == EMIT_POINT) {
        // point_t::sample_direct (src/emitter/point.cpp:45-61)
        const vec3 dl = wp - e.position;
        const float recp_dist2 = 1.f / length2(dl);
        const vec3 d = dl * sqrtf(recp_dist2);
        r.beam = make_forward_beam(e.position, d, emitter_spectral_value(sc, e, k), k, point_sourcing_geometry(e, k));
        beam_scale(r.beam, recp_dist2);
        r.dpd = pd_discrete(1.f);
    } else {
        float ppd;
        r.surface = area_sample_position(sc, e, sampler, ppd);
        r.has_surface = 1;
        const vec3 d = normalize(wp - r.surface.wp);
        const float dpd = area_pdf_direct(sc, e, wp, r.surface.wp, d, r.surface);
        const float recp_dpd = dpd > 0.f ? 1.f / dpd : 0.f;
        r.beam = area_Le(sc, e, r.surface.wp, d, k, r.surface);
        beam_scale(r.beam, recp_dpd);
        r.dpd = dpd;
    }
    return r;
}
// area_t::Li (area.cpp:35-53)
WT_HD stokes_t emitter_Li(const scene_t& sc, int ei, const beam_t& Sbeam, const surface_t& surface) {
    const emitter_t e = sc.emitters[ei];
    if (e.type != EMIT_AREA) return stokes_zero();
    const float dn = dot(-Sbeam.env.d, surface.geo.n);
    if (dn <= 0.f) return stokes_zero();
    beam_t Ibeam = area_Le(sc, e, surface.wp, -Sbeam.env.d, Sbeam.k, surface);
    beam_scale(Ibeam, 1.f / dn);
    return integrate_beams(Sbeam, Ibeam);
}

// ================================ scene-level sampling ===============================================
struct wavenumber_sample_t {
    float k;
    float wpd;   // tagged: discrete mass or density per (1/mm)
};
// piecewise-linear tabulated density over uniformly spaced k knots
WT_HD float kdist_pdf(const scene_t& sc, const kdist_t& d, float k) {
    if (d.discrete) return k == d.kmin ? 1.f : 0.f;
    if (k < d.kmin || k > d.kmax) return 0.f;
    const float* pdf = sc.kdist_data + d.offset;
    const float x = (k - d.kmin) / (d.kmax - d.kmin) * float(d.count - 1);
    uint32_t l = (uint32_t)x;
    if (l > d.count - 2) l = d.count - 2;
    const float f = x - float(l);
    return pdf[l] * (1.f - f) + pdf[l + 1] * f;
}
WT_HD wavenumber_sample_t kdist_sample(const scene_t& sc, const kdist_t& d, float u) {
    if (d.discrete) return {d.kmin, pd_discrete(1.f)};
    const float* pdf = sc.kdist_data + d.offset;
    const float* cdf = pdf + d.count;
    const uint32_t i = cdf_icdf(cdf, d.count - 1, u);
    const float dk = (d.kmax - d.kmin) / float(d.count - 1);
    const float p0 = pdf[i], p1 = pdf[i + 1];
    const float du = u - cdf[i];   // mass to cover inside the segment
    // solve p0*t + (p1-p0)*t^2/(2 dk) = du for t in [0,dk]
    float t;
    const float a = (p1 - p0) / (2.f * dk);
    if (fabsf(a) < 1e-12f * fmaxf_(p0, p1) || a == 0.f)
        t = p0 > 0.f ? du / p0 : 0.f;
    else {
        const float disc = fmaxf_(0.f, p0 * p0 + 4.f * a * du);
        t = 2.f * du / (p0 + sqrtf(disc));
    }
    t = clampf(t, 0.f, dk);
    const float k = d.kmin + (float(i) * dk + t);
    return {k, p0 + (p1 - p0) * (t / dk)};
}

struct emitter_k_sample_t {
    int32_t emitter;
    float emitter_pdf;
    wavenumber_sample_t wavenumber;
};
// scene_sensor_t::sample_emitter_and_spectrum (scene_sensor.cpp:35-59)
WT_HD emitter_k_sample_t scene_sample_emitter_and_spectrum(const scene_t& sc, sampler_t& sampler) {
    emitter_k_sample_t r;
    r.emitter = (int32_t)cdf_icdf(sc.emitter_cdf, sc.n_emitters, sampler_r(sampler));
    r.emitter_pdf = sc.emitters[r.emitter].select_pmf;
    r.wavenumber = kdist_sample(sc, sc.kdists[sc.emitters[r.emitter].k_dist], sampler_r(sampler));
    return r;
}
// scene_sensor_t::sum_spectral_pdf_for_all_emitters (scene_sensor.hpp:103-112)
WT_HD float scene_sum_spectral_pdf(const scene_t& sc, float k) {
    float s = 0.f;
    for (uint32_t i = 0; i < sc.n_emitters; ++i) s += sc.emitters[i].select_pmf * kdist_pdf(sc, sc.kdists[sc.emitters[i].k_dist], k);
    return s;
}
// scene_t::sample_emitter_direct (scene.hpp:127-138)
WT_HD emitter_direct_sample_t scene_sample_emitter_direct(const scene_t& sc, vec3 wp, float k, sampler_t& sampler) {
    const int32_t ei = (int32_t)cdf_icdf(sc.emitter_cdf, sc.n_emitters, sampler_r(sampler));
    const float pd = sc.emitters[ei].select_pmf;
    emitter_direct_sample_t s = emitter_sample_direct(sc, ei, wp, k, sampler);
    s.emitter_pdf = pd;
    beam_scale(s.beam, 1.f / pd);
    return s;
}

// ================================ sensors ===========================================================
WT_HD bool sensor_is_virtual(const sensor_t& s) { return s.type == SENSOR_VIRTUAL_PLANE; }
WT_HD bool sensor_is_delta_position(const sensor_t& s) { return s.type == SENSOR_PERSPECTIVE; }
WT_HD bool sensor_is_delta_direction(const sensor_t&) { return false; }

WT_HD void mat4_mul_point(const float* M, float x, float y, float z, float w, float out[4]) {
    for (int r = 0; r < 4; ++r) out[r] = M[r * 4 + 0] * x + M[r * 4 + 1] * y + M[r * 4 + 2] * z + M[r * 4 + 3] * w;
}
// perspective_t::point_on_sensor (perspective.hpp:70-74)
WT_HD vec3 persp_point_on_sensor(const sensor_t& s, vec2 film_pos) {
    float p[4];
    mat4_mul_point(s.inv_cam, film_pos.x, film_pos.y, 1.f, 1.f, p);
    return vec3{p[0], p[1], p[2]} / p[3];
}
// perspective_t::point_on_film (perspective.hpp:80-84)
WT_HD vec2 persp_point_on_film(const sensor_t& s, vec3 dir) {
    const vec3 p = dir / fabsf(dir.z);
    float d[4];
    mat4_mul_point(s.cam, p.x, p.y, 1.f, 1.f, d);
    return vec2{d[0], d[1]} / d[3];
}
WT_HD float persp_recp_sa_density(const sensor_t& s, vec3 d) { return s.sensor_area / sqr(0.01f) * (d.z * d.z * d.z); }
// perspective_t::sourcing_geometry (perspective.hpp:185-199)
WT_HD sourcing_geometry_t persp_sourcing_geometry(const sensor_t& s, float k) {
    const float initial_spatial_extent = s.element_extent_x * 0.25f * kBeamEnvelope;
    const phase_space_extent_t se =
        pse_enlarge(sg_phase_space_extent(sg_source(initial_spatial_extent, s.sourcing_tan_alpha, k)), s.phase_space_extent_scale);
    return sg_source(se);
}
// virtual_plane_sensor_t::sourcing_geometry (virtual_plane_sensor.hpp:117-133)
WT_HD sourcing_geometry_t vplane_sourcing_geometry(const sensor_t& s, float k) {
    const float initial_spatial_extent = (s.element_extent.x + s.element_extent.y) / 2.f * 0.25f * kBeamEnvelope;
    if (s.requested_tan_alpha >= 0.f) return sg_source(initial_spatial_extent, s.requested_tan_alpha, k);
    return sg_source_mub_from_length(initial_spatial_extent, k);
}
// virtual_plane_sensor_t::Se (virtual_plane_sensor.hpp:153-162); importance = 1/(pi * area)
WT_HD beam_t vplane_Se(const sensor_t& s, vec3 ro, vec3 rd, float k) {
    const float W = kInvPi * s.recp_area;
    const float dn = fmaxf_(0.f, dot(rd, s.frame.n));
    return make_backward_beam(ro, rd, W * dn, k, vplane_sourcing_geometry(s, k));
}
WT_HD sensor_element_t vplane_element_for_position(const sensor_t& s, vec3 wp) {
    const vec3 sp = wp - s.origin;
    const vec2 efp = vec2{dot(sp, s.frame.t), dot(sp, s.frame.b)} / s.element_extent;
    const uint32_t ex = (uint32_t)efp.x, ey = (uint32_t)efp.y;
    return {ex, ey, {efp.x - float(ex) - .5f, efp.y - float(ey) - .5f}};
}

// sensor_t::sample
WT_HD sensor_sample_t sensor_sample(const scene_t& sc, uint32_t px, uint32_t py, float k, sampler_t& sampler) {
    const sensor_t& s = sc.sensor;
    sensor_sample_t r;
    r.has_surface = 0;
    if (s.type == SENSOR_PERSPECTIVE) {
        const vec3 centre = persp_point_on_sensor(s, vec2{float(px) + .5f, float(py) + .5f});
        const vec2 pixel_offset = sampler_r2(sampler) - vec2{.5f, .5f};
        const vec3 dir_local = normalize(centre + pixel_offset.x * s.ddir_dx + pixel_offset.y * s.ddir_dy);
        const vec3 dir = to_world(s.frame, dir_local);
        const float recp_dpd = persp_recp_sa_density(s, dir_local);
        // Se(P, recp_dpd, k) * recp_dpd : scale = (1/recp_dpd) * recp_dpd
        r.beam = make_backward_beam(s.position, dir, 1.f / recp_dpd, k, persp_sourcing_geometry(s, k));
        beam_scale(r.beam, recp_dpd);
        r.ppd = pd_discrete(1.f);
        r.dpd = 1.f / recp_dpd;
        r.element = {px, py, pixel_offset};
    } else {
        const vec2 element_offset = sampler_r2(sampler) - vec2{.5f, .5f};
        const vec2 local = vec2{float(px) + element_offset.x + .5f, float(py) + element_offset.y + .5f} * s.element_extent;
        const vec3 p = s.origin + local.x * s.frame.t + local.y * s.frame.b;
        const float recp_ppd = 1.f / s.recp_area;
        const vec3 wo = cosine_hemisphere(sampler_r2(sampler));
        const float dpd = cosine_hemisphere_pdf(wo.z);
        r.beam = vplane_Se(s, p, to_world(s.frame, wo), k);
        beam_scale(r.beam, recp_ppd * (dpd > 0.f ? 1.f / dpd : 0.f));
        r.ppd = 1.f / recp_ppd;
        r.dpd = dpd;
        r.element = {px, py, element_offset};
        r.has_surface = 1;
        r.surface = make_dummy_surface(s.frame.n, p);
    }
    return r;
}
// sensor_t::sample_direct
WT_HD sensor_direct_sample_t sensor_sample_direct(const scene_t& sc, vec3 wp, float k, sampler_t& sampler) {
    const sensor_t& s = sc.sensor;
    sensor_direct_sample_t r;
    r.has_surface = 0;
    if (s.type == SENSOR_PERSPECTIVE) {
        const vec3 wdl = wp - s.position;
        const float recp_dist2 = 1.f / length2(wdl);
        const vec3 wd = wdl * sqrtf(recp_dist2);
        const vec3 dir_local = to_local(s.frame, wd);
        const vec2 fp = persp_point_on_film(s, dir_local);
        const float recp_sa = persp_recp_sa_density(s, dir_local);
        // (the reference converts possibly-negative floats to unsigned here; we reject them explicitly)
        const bool inside = dir_local.z > FLT_EPSILON && fp.x >= 0.f && fp.y >= 0.f && fp.x < float(s.width) && fp.y < float(s.height);
        const uint32_t ex = inside ? (uint32_t)fp.x : 0u, ey = inside ? (uint32_t)fp.y : 0u;
        const vec2 pixel_offset = vec2{fractf(fp.x), fractf(fp.y)} - vec2{.5f, .5f};
        r.beam = make_backward_beam(s.position, wd, 1.f / recp_sa, k, persp_sourcing_geometry(s, k));
        beam_scale(r.beam, recp_dist2 * (inside ? 1.f : 0.f));
        r.dpd = pd_discrete(1.f);
        r.element = {ex, ey, pixel_offset};
    } else {
        const vec2 splocal = sampler_r2(sampler) * s.extent;
        const vec3 sp = s.origin + splocal.x * s.frame.t + splocal.y * s.frame.b;
        const vec2 efp = splocal / s.element_extent;
        const uint32_t ex = (uint32_t)efp.x, ey = (uint32_t)efp.y;
        const vec2 element_offset = efp - vec2{float(ex), float(ey)} - vec2{.5f, .5f};
        const vec3 wdl = wp - sp;
        const float dist2 = length2(wdl);
        const vec3 wd = wdl / sqrtf(dist2);
        const vec3 wd_local = to_local(s.frame, wd);
        const float recp_dn = wd_local.z > 0.f ? 1.f / wd_local.z : 0.f;
        const float dpd = s.recp_area * dist2 * recp_dn;
        const float recp_dpd = dpd > 0.f ? 1.f / dpd : 0.f;
        r.beam = vplane_Se(s, sp, wd, k);
        beam_scale(r.beam, recp_dpd * recp_dn);
        r.dpd = dpd;
        r.element = {ex, ey, element_offset};
        r.has_surface = 1;
        r.surface = make_dummy_surface(s.frame.n, sp);
    }
    return r;
}
// virtual_coverage_sensor_t::Si (virtual_plane_sensor.cpp:65-103)
struct sensor_direct_connection_t {
    beam_t beam;
    sensor_element_t element;
    surface_t surface;
    bool valid;
};
WT_HD sensor_direct_connection_t vplane_Si(const scene_t& sc, const beam_t& beam, const range_t& range) {
    const sensor_t& s = sc.sensor;
    sensor_direct_connection_t r;
    r.valid = false;
    const vec3 n = s.frame.n;
    const float dn = dot(-beam.env.d, n);
    if (dn <= 0.f) return r;
    const vec3 a = s.origin;
    const vec3 b = s.origin + s.extent.x * s.frame.t;
    const vec3 c = s.origin + s.extent.y * s.frame.b;
    const vec3 d = s.origin + s.extent.x * s.frame.t + s.extent.y * s.frame.b;
    ray_tri_hit_t h1, h2;
    const bool i1 = intersect_ray_tri(beam.env.o, beam.env.d, a, b, c, range, h1);
    const bool i2 = intersect_ray_tri(beam.env.o, beam.env.d, c, b, d, range, h2);
    if (!i1 && !i2) return r;
    const vec3 p = beam.env.o + (i1 ? h1.dist : h2.dist) * beam.env.d;
    r.element = vplane_element_for_position(s, p);
    r.beam = vplane_Se(s, p, -beam.env.d, beam.k);
    beam_scale(r.beam, 1.f / dn);
    r.surface = make_dummy_surface(n, p);
    r.valid = true;
    return r;
}
// sensor_t::pdf_position / pdf_direction
WT_HD float sensor_pdf_position(const scene_t& sc) {
    return sc.sensor.type == SENSOR_PERSPECTIVE ? pd_discrete(1.f) : sc.sensor.recp_area;
}
WT_HD float sensor_pdf_direction(const scene_t& sc, vec3 dir) {
    const sensor_t& s = sc.sensor;
    const vec3 d = to_local(s.frame, dir);
    if (s.type == SENSOR_PERSPECTIVE) return d.z > FLT_EPSILON ? 1.f / persp_recp_sa_density(s, d) : 0.f;
    return cosine_hemisphere_pdf(fmaxf_(d.z, 0.f));
}

}   // namespace wt
