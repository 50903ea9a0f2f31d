// oracle/kat.cpp — C entry points exposing the restated physics primitives to the known-answer tests
// (tests/test_kat.py).                                                           *** TEST INFRASTRUCTURE ***
// Each wrapper just forwards to the function in wave_tracer_amd/csrc/wt/*.h that restates the cited reference code,
// so that numpy / scipy / brute-force references can pin it.
#include <cstring>

#include "../wave_tracer_amd/csrc/wt/bdpt.h"
#include "../wave_tracer_amd/csrc/wt/path.h"

using namespace wt;

extern "C" {

// K4: 2x2 SVD (math/linalg.hpp). A given column-major (a00,a01,a10,a11) glm style; out: Ucos,Usin,Vcos,Vsin,s1,s2
void kat_svd2(const float* A, float* out) {
    const svd_t s = svd2(mat2{A[0], A[1], A[2], A[3]});
    out[0] = s.Ucos; out[1] = s.Usin; out[2] = s.Vcos; out[3] = s.Vsin; out[4] = s.sigma1; out[5] = s.sigma2;
}
// K2: Fresnel (interaction/fresnel.hpp).  out: rs,rp,ts,tp,Ts,Tp,Z,tx,ty,tz
void kat_fresnel(float eta, const float* w, float* out) {
    const fresnel_t f = fresnel(cplx{eta, 0.f}, vec3{w[0], w[1], w[2]}, vec3{0, 0, 1});
    out[0] = f.rs.re; out[1] = f.rp.re; out[2] = f.ts.re; out[3] = f.tp.re; out[4] = f.Ts; out[5] = f.Tp; out[6] = f.Z;
    out[7] = f.t.x; out[8] = f.t.y; out[9] = f.t.z;
}
void kat_fresnel_conductor(float eta_re, float eta_im, float cos_i, float* out) {
    const float s = sqrtf(fmaxf_(0.f, 1.f - cos_i * cos_i));
    const fresnel_conductor_t f = fresnel_reflection(cplx{eta_re, eta_im}, vec3{s, 0, cos_i}, vec3{0, 0, 1});
    out[0] = f.rs.re; out[1] = f.rs.im; out[2] = f.rp.re; out[3] = f.rp.im;
}
// K3: Mueller algebra
void kat_mueller_fresnel(float fs_re, float fs_im, float fp_re, float fp_im, float* out16) {
    const mueller_t M = mueller_fresnel(cplx{fs_re, fs_im}, cplx{fp_re, fp_im});
    std::memcpy(out16, M.m, sizeof(M.m));
}
void kat_mueller_rotation(float ax, float ay, float bx, float by, float* out16) {
    const mueller_t M = mueller_rotation(vec2{ax, ay}, vec2{bx, by});
    std::memcpy(out16, M.m, sizeof(M.m));
}
void kat_stokes_reorient(const float* S, const float* f0, const float* f1, float* out4) {
    const frame_t a{{f0[0], f0[1], f0[2]}, {f0[3], f0[4], f0[5]}, {f0[6], f0[7], f0[8]}}, b{{f1[0], f1[1], f1[2]}, {f1[3], f1[4], f1[5]}, {f1[6], f1[7], f1[8]}};
    const stokes_t r = stokes_reorient(stokes_t{{S[0], S[1], S[2], S[3]}}, a, b);
    std::memcpy(out4, r.s, sizeof(r.s));
}
void kat_build_orthogonal_frame(const float* n, float* out9) {
    const frame_t f = build_orthogonal_frame(vec3{n[0], n[1], n[2]});
    const float v[9] = {f.t.x, f.t.y, f.t.z, f.b.x, f.b.y, f.b.z, f.n.x, f.n.y, f.n.z};
    std::memcpy(out9, v, sizeof(v));
}
// K5: cone primitives.  cone: o(3) d(3) tan_alpha x0 ecc ; returns hit flag, dist in out[0]
int kat_cone_tri(const float* c, const float* tri, float rmin, float rmax, float* out) {
    const vec3 d = normalize(vec3{c[3], c[4], c[5]});
    const cone_t cone = make_cone(vec3{c[0], c[1], c[2]}, d, build_orthogonal_frame(d).t, c[6], c[8], c[7]);
    const vec3 a{tri[0], tri[1], tri[2]}, b{tri[3], tri[4], tri[5]}, cc{tri[6], tri[7], tri[8]};
    const vec3 n = normalize(cross(b - a, cc - a));
    cone_tri_hit_t h;
    const bool hit = intersect_cone_tri(cone, a, b, cc, n, range_t{rmin, rmax}, h);
    out[0] = hit ? h.dist : -1.f;
    return hit;
}
// ... with the cone given by all its parameters (replays the queries of a render: oracle/indep/prims2.cpp's WT_SS_DUMP): rec = o3 d3 x3 x0 tan_alpha e a3 b3 c3 zmin zmax
int kat_cone_tri_raw(const float* r, float* out) {
    const cone_t cone = make_cone_raw(vec3{r[0], r[1], r[2]}, vec3{r[3], r[4], r[5]}, vec3{r[6], r[7], r[8]}, r[9], r[10], 1.f / r[11], r[11]);
    const vec3 a{r[12], r[13], r[14]}, b{r[15], r[16], r[17]}, cc{r[18], r[19], r[20]};
    const vec3 n = normalize(cross(b - a, cc - a));
    cone_tri_hit_t h;
    const bool hit = intersect_cone_tri(cone, a, b, cc, n, range_t{r[21], r[22]}, h);
    out[0] = hit ? h.dist : -1.f;
    return hit;
}
// cone_box_outside (wt/bvh.h: the conservative cone x AABB cull of the traversals): 1 = culled.  box: min3, max3 (world)
int kat_cone_box_outside(const float* c, const float* box, float rmin, float rmax) {
    const vec3 d = normalize(vec3{c[3], c[4], c[5]});
    const cone_t cone = make_cone(vec3{c[0], c[1], c[2]}, d, build_orthogonal_frame(d).t, c[6], c[8], c[7]);
    return cone_box_outside(box[0] - cone.o.x, box[1] - cone.o.y, box[2] - cone.o.z, box[3] - cone.o.x, box[4] - cone.o.y, box[5] - cone.o.z, cone.d, cone.tan_alpha,
                            cone.x0, range_t{rmin, rmax})
               ? 1
               : 0;
}
// cone_sphere_maybe on the triangle's bounding sphere (wt/cone.h: the first filter of the wave-cooperative queries): 1 = kept.  out4: the sphere
int kat_cone_tri_sphere_maybe(const float* c, const float* tri, float rmin, float rmax, float* out4) {
    const vec3 d = normalize(vec3{c[3], c[4], c[5]});
    const cone_t cone = make_cone(vec3{c[0], c[1], c[2]}, d, build_orthogonal_frame(d).t, c[6], c[8], c[7]);
    tri_bounding_sphere(vec3{tri[0], tri[1], tri[2]}, vec3{tri[3], tri[4], tri[5]}, vec3{tri[6], tri[7], tri[8]}, out4);
    return cone_sphere_maybe(cone, vec3{out4[0], out4[1], out4[2]}, out4[3], range_t{rmin, rmax}) ? 1 : 0;
}
int kat_cone_tri_maybe(const float* c, const float* tri, float rmin, float rmax) {
    const vec3 d = normalize(vec3{c[3], c[4], c[5]});
    const cone_t cone = make_cone(vec3{c[0], c[1], c[2]}, d, build_orthogonal_frame(d).t, c[6], c[8], c[7]);
    return cone_tri_maybe(cone, vec3{tri[0], tri[1], tri[2]}, vec3{tri[3], tri[4], tri[5]}, vec3{tri[6], tri[7], tri[8]}, range_t{rmin, rmax}) ? 1 : 0;
}
int kat_cone_contains(const float* c, const float* p) {
    const vec3 d = normalize(vec3{c[3], c[4], c[5]});
    const cone_t cone = make_cone(vec3{c[0], c[1], c[2]}, d, build_orthogonal_frame(d).t, c[6], c[8], c[7]);
    return cone_contains_local(cone, to_local(cone_frame(cone), vec3{p[0], p[1], p[2]} - cone.o), range_positive());
}
void kat_cone_local(const float* c, const float* p, float* out3) {
    const vec3 d = normalize(vec3{c[3], c[4], c[5]});
    const cone_t cone = make_cone(vec3{c[0], c[1], c[2]}, d, build_orthogonal_frame(d).t, c[6], c[8], c[7]);
    const vec3 l = to_local(cone_frame(cone), vec3{p[0], p[1], p[2]} - cone.o);
    out3[0] = l.x; out3[1] = l.y; out3[2] = l.z;
}
int kat_ray_tri(const float* o, const float* d, const float* tri, float* out3) {
    ray_tri_hit_t h;
    const bool hit = intersect_ray_tri(vec3{o[0], o[1], o[2]}, vec3{d[0], d[1], d[2]}, vec3{tri[0], tri[1], tri[2]}, vec3{tri[3], tri[4], tri[5]},
                                       vec3{tri[6], tri[7], tri[8]}, range_positive(), h);
    out3[0] = h.dist; out3[1] = h.bx; out3[2] = h.by;
    return hit;
}
// K6: minimum-uncertainty relations (beam_geometry.hpp)
float kat_mub_tan_alpha(float len_m, float k) { return mub_tan_alpha_from_length(len_m, k); }
float kat_mub_length(float tan_alpha, float k) { return mub_spatial_length_from_tan_alpha(tan_alpha, k); }
// gaussian over triangle (gauss.h)
float kat_gauss_triangle(const float* t) { return gauss_integrate_triangle_canonical(vec2{t[0], t[1]}, vec2{t[2], t[3]}, vec2{t[4], t[5]}); }
// K7: fractal profile (surface_profile/fractal.hpp)
float kat_fractal_psd(float roughness, float gamma, float k, const float* wi, const float* wo) {
    material_t m{};
    m.type = MAT_SURFACE_SPM; m.profile = PROFILE_FRACTAL; m.roughness = roughness; m.gamma = gamma;
    return profile_psd(m, vec3{wi[0], wi[1], wi[2]}, vec3{wo[0], wo[1], wo[2]}, k);
}
float kat_fractal_pdf(float roughness, float gamma, float k, const float* wi, const float* wo) {
    material_t m{};
    m.type = MAT_SURFACE_SPM; m.profile = PROFILE_FRACTAL; m.roughness = roughness; m.gamma = gamma;
    return profile_pdf(m, vec3{wi[0], wi[1], wi[2]}, vec3{wo[0], wo[1], wo[2]}, k);
}
float kat_fractal_alpha(float roughness, float gamma, float k, const float* wi, const float* wo) {
    material_t m{};
    m.type = MAT_SURFACE_SPM; m.profile = PROFILE_FRACTAL; m.roughness = roughness; m.gamma = gamma;
    return profile_alpha(m, vec3{wi[0], wi[1], wi[2]}, vec3{wo[0], wo[1], wo[2]}, k);
}
void kat_fractal_sample(float roughness, float gamma, float k, const float* wi, uint64_t seed, uint32_t n, float* out /* n x {wo3,pdf,psd} */) {
    material_t m{};
    m.type = MAT_SURFACE_SPM; m.profile = PROFILE_FRACTAL; m.roughness = roughness; m.gamma = gamma;
    for (uint32_t i = 0; i < n; ++i) {
        sampler_t s = make_sampler(seed, i, 0);
        const profile_sample_t p = profile_sample(m, vec3{wi[0], wi[1], wi[2]}, k, s);
        out[5 * i] = p.wo.x; out[5 * i + 1] = p.wo.y; out[5 * i + 2] = p.wo.z; out[5 * i + 3] = p.pdf; out[5 * i + 4] = p.psd;
    }
}
// gaussian profile (surface_profile/gaussian.hpp); what: 0 psd, 1 pdf, 2 alpha
float kat_gaussian(int what, float roughness, float sigma, float k, const float* wi, const float* wo) {
    material_t m{};
    m.type = MAT_SURFACE_SPM; m.profile = PROFILE_GAUSSIAN; m.roughness = roughness; m.gauss_sigma = sigma; m.gamma = 3.f;
    const vec3 a{wi[0], wi[1], wi[2]}, b{wo[0], wo[1], wo[2]};
    return what == 0 ? profile_psd(m, a, b, k) : (what == 1 ? profile_pdf(m, a, b, k) : profile_alpha(m, a, b, k));
}
void kat_gaussian_sample(float roughness, float sigma, float k, const float* wi, uint64_t seed, uint32_t n, float* out /* n x {wo3,pdf,psd} */) {
    material_t m{};
    m.type = MAT_SURFACE_SPM; m.profile = PROFILE_GAUSSIAN; m.roughness = roughness; m.gauss_sigma = sigma; m.gamma = 3.f;
    for (uint32_t i = 0; i < n; ++i) {
        sampler_t s = make_sampler(seed, i, 0);
        const profile_sample_t p = profile_sample(m, vec3{wi[0], wi[1], wi[2]}, k, s);
        out[5 * i] = p.wo.x; out[5 * i + 1] = p.wo.y; out[5 * i + 2] = p.wo.z; out[5 * i + 3] = p.pdf; out[5 * i + 4] = p.psd;
    }
}
// K8: Fraunhofer FSD kernel functions
float kat_fsd_alpha1(float x, float y) { return fsd_alpha1(x, y); }
float kat_fsd_alpha2(float x, float y) { return fsd_alpha2(x, y); }
float kat_fsd_chi_e(float x, float y) { return fsd_chi_e(vec2{x, y}); }
// K9: film reconstruction weights
void kat_film_weights(float sigma, int radius, float ox, float oy, float* out /* wx[5], wy[5], recp_total */) {
    sensor_t s{};
    s.rfilter_sigma = sigma;
    s.rf_radius = radius;
    const rfilter_weights_t w = film_rfilter_weights(s, vec2{ox, oy});
    std::memcpy(out, w.wx, 5 * sizeof(float));
    std::memcpy(out + 5, w.wy, 5 * sizeof(float));
    out[10] = w.recp_total;
}
// RNG
void kat_philox(uint64_t seed, uint64_t sample_id, uint32_t stream, uint32_t n, float* out) {
    sampler_t s = make_sampler(seed, sample_id, stream);
    for (uint32_t i = 0; i < n; ++i) out[i] = sampler_r(s);
}
void kat_philox_raw(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }
void kat_cosine_hemisphere(float u0, float u1, float* out3) {
    const vec3 d = cosine_hemisphere(vec2{u0, u1});
    out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}
// eft
float kat_diff_prod(float a, float b, float c, float d) { return diff_prod(a, b, c, d); }

// scene-level: spectral sampling distribution checks
float kat_kdist_pdf(const void* scene_host, int emitter, float k) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    return kdist_pdf(sc, sc.kdists[sc.emitters[emitter].k_dist], k);
}
float kat_kdist_sample(const void* scene_host, int emitter, float u, float* pdf_out) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    const wavenumber_sample_t w = kdist_sample(sc, sc.kdists[sc.emitters[emitter].k_dist], u);
    *pdf_out = w.wpd;
    return w.k;
}
// the per-lookup RGB uplift of spectral bitmap textures (wt/scene.h: rgb_uplift)
float kat_rgb_uplift(float r, float g, float b, float k) { return rgb_uplift(r, g, b, k); }
float kat_spectrum(const void* scene_host, int id, float k, float* im) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    const cplx v = spectrum_value(sc, id, k);
    if (im) *im = v.im;
    return v.re;
}
// triangle soup of the flattened scene (BVH order): n_tris x {a,b,c,n} for brute-force ADS checks
uint32_t kat_scene_tris(const void* scene_host, float* out12) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    if (out12) std::memcpy(out12, sc.tri_geo, sizeof(tri_geo_t) * sc.n_tris);
    return sc.n_tris;
}
int kat_material_ior_spec(const void* scene_host, int material) { return static_cast<const scene_t*>(scene_host)->materials[material].ior_spec; }
int kat_material_refl_spec(const void* scene_host, int material) { return static_cast<const scene_t*>(scene_host)->materials[material].refl_spec; }

// K1: UTD transition function and wedge diffraction coefficients (interaction/fsd/utd.hpp)
void kat_utd_F(float x, float* out2) {
    const cplx f = utd_F(x);
    out2[0] = f.re;
    out2[1] = f.im;
}
// wedge: {v3, l, nff3, tff3, nbf3, alpha}; out: Ds.re, Ds.im, Dh.re, Dh.im
void kat_wedge_UTD(const float* wd, float k, const float* wi, const float* wo, float ro, float* out4) {
    utd_wedge_t w{{wd[0], wd[1], wd[2]}, wd[3], {wd[4], wd[5], wd[6]}, {wd[7], wd[8], wd[9]}, {wd[10], wd[11], wd[12]}, wd[13], 0u};
    const utd_ret_t r = wedge_UTD(w, k, vec3{wi[0], wi[1], wi[2]}, vec3{wo[0], wo[1], wo[2]}, ro);
    out4[0] = r.Ds.re; out4[1] = r.Ds.im; out4[2] = r.Dh.re; out4[3] = r.Dh.im;
}
int kat_wedge_diffraction_point(const float* wd, const float* src, const float* dst, float* out3) {
    utd_wedge_t w{{wd[0], wd[1], wd[2]}, wd[3], {wd[4], wd[5], wd[6]}, {wd[7], wd[8], wd[9]}, {wd[10], wd[11], wd[12]}, wd[13], 0u};
    vec3 p{0, 0, 0};
    const bool ok = wedge_diffraction_point(w, vec3{src[0], src[1], src[2]}, vec3{dst[0], dst[1], dst[2]}, p);
    out3[0] = p.x; out3[1] = p.y; out3[2] = p.z;
    return ok ? 1 : 0;
}
int kat_wedge_diffraction_point_dir(const float* wd, const float* src, const float* wo, float* out3) {
    utd_wedge_t w{{wd[0], wd[1], wd[2]}, wd[3], {wd[4], wd[5], wd[6]}, {wd[7], wd[8], wd[9]}, {wd[10], wd[11], wd[12]}, wd[13], 0u};
    vec3 p{0, 0, 0};
    const bool ok = wedge_diffraction_point_dir(w, vec3{src[0], src[1], src[2]}, vec3{wo[0], wo[1], wo[2]}, p);
    out3[0] = p.x; out3[1] = p.y; out3[2] = p.z;
    return ok ? 1 : 0;
}
void kat_edge_ellipsoid(const float* p0, const float* p1, const float* c, const float* x, const float* y, const float* axes, float* out2) {
    const vec2 t = intersect_edge_ellipsoid(vec3{p0[0], p0[1], p0[2]}, vec3{p1[0], p1[1], p1[2]}, vec3{c[0], c[1], c[2]}, vec3{x[0], x[1], x[2]},
                                            vec3{y[0], y[1], y[2]}, vec3{axes[0], axes[1], axes[2]});
    out2[0] = t.x;
    out2[1] = t.y;
}
// UTD aperture of the given scene edges seen from `src` through the region at `wp`: samples n directions and returns per sample
// {wo3, weight, is_direct, pdf(wo)}; the total field |ts|^2,|th|^2 (no occlusion tests) towards dst in out_field
uint32_t kat_utd_aperture(const void* scene_host, const uint32_t* edge_ids, uint32_t n_ids, const float* wp, const float* frame9, const float* size3,
                          const float* src, float k, uint64_t seed, uint32_t n, float* out /* n x 6 */) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    static utd_edge_rec_t recs[kUtdMaxEdges];
    utd_aperture_t ap;
    const frame_t fr{{frame9[0], frame9[1], frame9[2]}, {frame9[3], frame9[4], frame9[5]}, {frame9[6], frame9[7], frame9[8]}};
    const vec3 s{src[0], src[1], src[2]}, p{wp[0], wp[1], wp[2]};
    ap.edge_offset = 0;
    ap.edge_cap = kUtdMaxEdges;
    utd_build_aperture(sc, p, fr, vec3{size3[0], size3[1], size3[2]}, normalize(s - p), k, edge_ids, n_ids, ap, utd_edges_ref_t{recs, 1});
    for (uint32_t i = 0; i < n; ++i) {
        sampler_t smp = make_sampler(seed, i, 7);
        const utd_sample_t us = utd_sample(sc, ap, utd_edges_ref_t{recs, 1}, s, smp);
        float* o = out + 6 * i;
        o[0] = us.wo.x; o[1] = us.wo.y; o[2] = us.wo.z; o[3] = us.weight; o[4] = (float)us.is_direct;
        o[5] = us.weight > 0.f && !us.is_direct ? utd_pdf(sc, ap, utd_edges_ref_t{recs, 1}, s, us.wo) : 0.f;
    }
    return ap.n_edges;
}
uint32_t kat_scene_edges(const void* scene_host, uint32_t first, uint32_t n, float* out /* n x {a3,b3,n1 3,n2 3,alpha} */) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    for (uint32_t i = 0; i < n && first + i < sc.n_edges; ++i) {
        const edge_t e = sc.edges[first + i];
        float* o = out + 13 * i;
        o[0] = e.a.x; o[1] = e.a.y; o[2] = e.a.z; o[3] = e.b.x; o[4] = e.b.y; o[5] = e.b.z;
        o[6] = e.n1.x; o[7] = e.n1.y; o[8] = e.n1.z; o[9] = e.n2.x; o[10] = e.n2.y; o[11] = e.n2.z; o[12] = e.alpha;
    }
    return sc.n_edges;
}

// K10 (MIS weights sum to one) reduces to: every density a walk STORES when it samples equals the density the MIS code EVALUATES
// for the same transition.  out: n x {sampled dpd (tagged), evaluated pdf, M00 * dpd, f00(wi, wo), wo.z}
void kat_material_sample_consistency(const void* scene_host, int mat, const float* wi3, float k, uint32_t transport, uint64_t seed, uint32_t n, float* out) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    const vec3 wi{wi3[0], wi3[1], wi3[2]};
    for (uint32_t i = 0; i < n; ++i) {
        sampler_t s = make_sampler(seed, i, 3);
        const bsdf_sample_t bs = material_sample(sc, mat, wi, k, transport, s);
        float* o = out + 5 * i;
        o[0] = bs.valid ? bs.dpd : 0.f;
        o[1] = bs.valid ? material_pdf(sc, mat, wi, bs.wo, k, transport) : 0.f;
        o[2] = bs.valid ? bs.M.m[0] * bs.dpd : 0.f;
        o[3] = bs.valid ? material_f(sc, mat, wi, bs.wo, k, transport).m[0] : 0.f;
        o[4] = bs.wo.z;
    }
}
// out: n x {sampled dpd, sensor_pdf_direction(dir), sampled ppd (tagged), sensor_pdf_position}
void kat_sensor_sample_consistency(const void* scene_host, uint32_t px, uint32_t py, float k, uint64_t seed, uint32_t n, float* out) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    for (uint32_t i = 0; i < n; ++i) {
        sampler_t s = make_sampler(seed, i, 4);
        const sensor_sample_t ss = sensor_sample(sc, px, py, k, s);
        float* o = out + 4 * i;
        o[0] = ss.dpd;
        o[1] = sensor_pdf_direction(sc, ss.beam.env.d);
        o[2] = ss.ppd;
        o[3] = sensor_pdf_position(sc);
    }
}
// emitted flux estimate: mean over n emitter samples of the sourced beam's intensity (= flux / spectral unit); also returns the
// emitter's spectral value at k through *value
double kat_emitter_mean_flux(const void* scene_host, int ei, float k, uint64_t seed, uint32_t n, float* value) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    double acc = 0;
    for (uint32_t i = 0; i < n; ++i) {
        sampler_t s = make_sampler(seed, i, 8);
        acc += (double)beam_intensity(emitter_sample(sc, ei, k, s).beam);
    }
    if (value) *value = emitter_spectral_value(sc, sc.emitters[ei], k);
    return acc / n;
}
// out: n x {sampled dpd (tagged), emitter_pdf_direction(dir), sampled ppd (tagged), emitter_pdf_position}
void kat_emitter_sample_consistency(const void* scene_host, int ei, float k, uint64_t seed, uint32_t n, float* out) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    for (uint32_t i = 0; i < n; ++i) {
        sampler_t s = make_sampler(seed, i, 5);
        const emitter_sample_t es = emitter_sample(sc, ei, k, s);
        float* o = out + 4 * i;
        o[0] = es.dpd;
        o[1] = emitter_pdf_direction(sc, ei, es.beam.env.d, es.has_surface ? &es.surface : nullptr);
        o[2] = es.ppd;
        o[3] = emitter_pdf_position(sc, ei, es.has_surface ? &es.surface : nullptr);
    }
}
// Textured area emitters (wt/sources.h area_table_*; src/emitter/area.cpp:153-271).  kat_area_table: the emitter's tables as stored —
// returns the number of words (0: no radiance texture) and copies up to `cap` of them.  kat_area_samples: n position samples, out: n x
// {mesh triangle, alpha, beta, sampled ppd, pdf_position at the sampled surface, u, v, radiance at k}
uint32_t kat_area_table(const void* scene_host, int ei, float* out, uint32_t cap) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    const emitter_t& e = sc.emitters[ei];
    if (e.type != EMIT_AREA || e.radiance_tex <= 0) return 0;
    for (uint32_t i = 0; i < e.tab_words && i < cap; ++i) out[i] = sc.texture_data[e.tab + i];
    return e.tab_words;
}
void kat_area_samples(const void* scene_host, int ei, float k, uint64_t seed, uint32_t n, float* out) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    const emitter_t e = sc.emitters[ei];
    for (uint32_t i = 0; i < n; ++i) {
        sampler_t s = make_sampler(seed, i, 5);
        float ppd = 0.f;
        const surface_t srf = area_sample_position(sc, e, s, ppd);
        float* o = out + 8 * i;
        o[0] = (float)sc.tri_meta[srf.tuid].shape_tri_idx;
        o[1] = srf.bary.x;
        o[2] = srf.bary.y;
        o[3] = ppd;
        o[4] = area_pdf_position(sc, e, &srf);
        o[5] = srf.uv.x;
        o[6] = srf.uv.y;
        o[7] = area_spectral_radiance(sc, e, srf, k);
    }
}

// Fraunhofer aperture of ALL scene edges seen by a beam (cone6 = {o3, d3}, tan_alpha, x0) at distance `dist`: n samples,
// out: n x {wo3 (aperture frame), sampled dpd, fsd_pdf(wo), weight}; returns the number of aperture segments
uint32_t kat_fsd_sample_consistency(const void* scene_host, const float* cone6, float tan_alpha, float x0, float dist, float k, uint64_t seed, uint32_t n,
                                    float* out) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    const vec3 d = normalize(vec3{cone6[3], cone6[4], cone6[5]});
    const cone_t env = make_cone_iso(vec3{cone6[0], cone6[1], cone6[2]}, d, tan_alpha, x0);
    const frame_t fr = cone_frame(env);
    const vec2 ax = cone_axes(env, dist);
    const vec2 sigma{ax.x / kBeamEnvelope, ax.y / kBeamEnvelope};
    static fsd_edge_t edges[kFsdMaxEdges];
    static uint32_t ids[4096];
    const uint32_t n_ids = sc.n_edges < 4096 ? sc.n_edges : 4096;
    for (uint32_t i = 0; i < n_ids; ++i) ids[i] = i;
    fsd_aperture_t ap;
    ap.edge_offset = 0;
    ap.edge_cap = kFsdMaxEdges;
    const fsd_edges_ref_t ed{edges, 1};
    cone_t beam = env;
    beam.o = env.o + dist * d;   // the aperture is built in the frame at the interaction point (bdpt_walk_step passes the beam itself)
    fsd_build_aperture(sc, fr, k, 1.f, env, ids, n_ids, sigma, ap, ed);
    for (uint32_t i = 0; i < n; ++i) {
        sampler_t s = make_sampler(seed, i, 6);
        const fsd_sample_t fs = fsd_sample(sc, ap, ed, s);
        float* o = out + 6 * i;
        o[0] = fs.wo.x; o[1] = fs.wo.y; o[2] = fs.wo.z; o[3] = fs.dpd; o[4] = fs.dpd > 0.f ? fsd_pdf(ap, ed, fs.wo) : 0.f; o[5] = fs.weight;
    }
    return ap.n_edges;
}

// K8b: synthetic Fraunhofer aperture (edges given directly in fsd units: n x {e.x,e.y,v.x,v.y,a_b,iab_2}); psi02/P0/pdfs derived like
// free_space_diffraction.cpp:106-128.  ap_out = {P0, P0_pdf, psi02, edge pdfs...}
static void kat_make_aperture(const float* edges, uint32_t n_edges, float k, fsd_aperture_t& ap, fsd_edge_t* store) {
    const fsd_edges_ref_t ed{store, 1};
    ap.edge_offset = 0;
    ap.edge_cap = kFsdMaxEdges;
    fsd_build_state_t st = fsd_build_begin(frame_t{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, k, 1.f, vec2{1.f, 1.f}, ap);
    for (uint32_t i = 0; i < n_edges && i < kFsdMaxEdges; ++i) {
        fsd_edge_t fe;
        fe.e = {edges[6 * i], edges[6 * i + 1]};
        fe.v = {edges[6 * i + 2], edges[6 * i + 3]};
        fe.ab = edges[6 * i + 4];
        fe.iab = edges[6 * i + 5];
        fe.pdf = fsd_Pj(fe);
        ed.set(ap.n_edges++, fe);
        st.P_total += fe.pdf;
    }
    fsd_build_finish(k, st, ap, ed);
}
// n proposals (sampleN) and n rejection-sampled directions (fsd_run_tries): out = n x {prop.x, prop.y, acc.x, acc.y, accepted}
// ignore_dead: run the reference's full loop even when the aperture is classified dead (wt/fsd.h: kFsdDeadRatio); tries_out (optional): tries
// per direction; dead_out (optional): the classification
void kat_fsd_aperture_sample2(const void* scene_host, const float* edges, uint32_t n_edges, float k, uint64_t seed, uint32_t n, float* out, float* ap_out,
                              int ignore_dead, uint32_t* tries_out, uint32_t* dead_out) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    static fsd_edge_t store[kFsdMaxEdges];
    fsd_aperture_t ap;
    kat_make_aperture(edges, n_edges, k, ap, store);
    if (dead_out) *dead_out = ap.dead;
    if (ignore_dead) ap.dead = 0;
    const fsd_edges_ref_t ed{store, 1};
    ap_out[0] = ap.P0;
    ap_out[1] = ap.P0_pdf;
    ap_out[2] = ap.psi02;
    for (uint32_t i = 0; i < ap.n_edges; ++i) ap_out[3 + i] = ed.get(i).pdf;
    for (uint32_t i = 0; i < n; ++i) {
        sampler_t s = make_sampler(seed, i, 7);
        const vec2 xi = fsd_sampleN(sc, ap, ed, s);
        sampler_t s2 = make_sampler(seed, i, 8);
        fsd_try_t r{{0.f, 0.f}, 0.f, 0u};
        const uint32_t t = fsd_run_tries(sc, ap, ed, s2, fsd_tries_base(s2), 0, fsd_max_tries(ap), r);
        float* o = out + 5 * i;
        o[0] = xi.x; o[1] = xi.y; o[2] = r.x.x; o[3] = r.x.y; o[4] = t != 0xFFFFFFFFu ? 1.f : 0.f;
        if (tries_out) tries_out[i] = t != 0xFFFFFFFFu ? t + 1u : fsd_max_tries(ap);
    }
}
void kat_fsd_aperture_sample(const void* scene_host, const float* edges, uint32_t n_edges, float k, uint64_t seed, uint32_t n, float* out, float* ap_out) {
    kat_fsd_aperture_sample2(scene_host, edges, n_edges, k, seed, n, out, ap_out, 0, nullptr, nullptr);
}
// out = n x {ASF (fsd.hpp:143-146), sampling_density (fsd.hpp:122-127)} at the given xi
void kat_fsd_aperture_eval(const float* edges, uint32_t n_edges, float k, const float* xi, uint32_t n, float* out) {
    static fsd_edge_t store[kFsdMaxEdges];
    fsd_aperture_t ap;
    kat_make_aperture(edges, n_edges, k, ap, store);
    const fsd_edges_ref_t ed{store, 1};
    for (uint32_t i = 0; i < n; ++i) {
        out[2 * i] = fsd_ASF(ap, ed, vec2{xi[2 * i], xi[2 * i + 1]});
        out[2 * i + 1] = fsd_sampling_density(ap, ed, vec2{xi[2 * i], xi[2 * i + 1]});
    }
}
// |sum_j Psi_j(xi)|^2 alone (fsd.hpp:143-146 without the masks chi_e / chi_0 and the 0-th order term): what tests/test_second_source.py compares
// with the Fourier integral over the aperture polygon
void kat_fsd_asf_unclamped(const float* edges, uint32_t n_edges, float k, const float* xi, uint32_t n, float* out) {
    static fsd_edge_t store[kFsdMaxEdges];
    fsd_aperture_t ap;
    kat_make_aperture(edges, n_edges, k, ap, store);
    const fsd_edges_ref_t ed{store, 1};
    for (uint32_t i = 0; i < n; ++i) out[i] = fsd_ASF_unclamped(ap, ed, vec2{xi[2 * i], xi[2 * i + 1]});
}
// raw LUT draw in zeta space (fsd_lut.hpp:50-69)
void kat_fsd_lut_sample(const void* scene_host, int which, uint64_t seed, uint32_t n, float* out2) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    for (uint32_t i = 0; i < n; ++i) {
        sampler_t s = make_sampler(seed, i, 9);
        const vec2 z = fsd_lut_sample(sc.lut, sampler_r3(s), which == 0);
        out2[2 * i] = z.x;
        out2[2 * i + 1] = z.y;
    }
}

}   // extern "C"

// texture addressing (wt/scene.h: tex_wrap_coord) for tests/test_textures.py
extern "C" int kat_tex_wrap(uint32_t mode, int c, int dim) { return wt::tex_wrap_coord(mode, c, dim); }
// one lookup of a 1-channel bitmap texture (wt/scene.h: tex_bitmap) with the given filter (0 nearest, 1 bilinear, 2 bicubic) and wrap modes
extern "C" float kat_tex_bitmap(const float* texels, uint32_t w, uint32_t h, uint32_t filter, uint32_t uwrap, uint32_t vwrap, float u, float v) {
    wt::scene_t sc;
    std::memset(&sc, 0, sizeof(sc));
    sc.texture_data = texels;
    wt::texture_t t;
    std::memset(&t, 0, sizeof(t));
    t.type = wt::TEX_BITMAP;
    t.width = w;
    t.height = h;
    t.channels = 1;
    t.offset = 0;
    t.bilinear = filter;
    t.uwrap = uwrap;
    t.vwrap = vwrap;
    return wt::tex_bitmap(sc, t, wt::vec2{u, v}).r;
}
