// wave_tracer_amd — UTD (uniform theory of diffraction) free-space diffraction: wedge diffraction coefficients, Keller-cone
// diffraction points, the free-space-diffraction aperture of an interaction region and its sampling.
//
// Reference: include/wt/interaction/fsd/utd.hpp:20-172 (UTDa, UTDF, diffraction_point, UTD),
//            include/wt/interaction/fsd/common.hpp:20-100 (wedge_edge_t, fsd_aperture_t),
//            include/wt/interaction/fsd/free_space_diffraction.hpp:31-102,
//            src/interaction/fsd/free_space_diffraction.cpp:20-234 (aperture ctor, sample, pdf, f),
//            include/wt/math/intersect/misc.hpp:40-72 (intersect_edge_ellipsoid).
//
// The reference evaluates the transition function F through libcerf's complex erfc (`cerfc`, utd.hpp:42; the submodule is
// empty in the reference checkout, version unknown).  Here erfc(e^{i pi/4} sqrt(x)) is evaluated from the Maclaurin series of
// erf in double precision (|x| < 6: at most 48 terms, absolute error < 1e-12; pinned against scipy.special.erfc in
// tests/test_kat.py), then rounded to float like the reference's c_t{cerfc(...)}.
//
// The aperture of one interaction region is stored compactly (edge id + front-face bit + the two clamping parameters,
// 12 B per wedge) and the wedge is rebuilt from the scene's edge table when evaluated.  The number of wedges per aperture is not bounded
// (the reference's is a std::vector): an aperture owns `edge_cap` records at `edge_offset` of a pool, sized by a counting pass before it
// is built (utd_count_wedges); kUtdMaxEdges is only the size of the fixed scratch arrays of the CPU checker and the KATs.
#pragma once
#include "scene.h"
#include "rng.h"

namespace wt {

constexpr uint32_t kUtdMaxEdges = 4096;
constexpr float kUtdMinSinBeta = 1e-3f;    // utd.hpp:20
constexpr float kUtdIsSigmaScale = 45.f;   // free_space_diffraction.cpp:20

// glm::mod (floor modulo)
WT_HD float modf_floor(float x, float y) { return x - y * floorf(x / y); }

// utd.hpp:25-31
template <int sgn>
WT_HD float utd_a(float phi, float n) {
    const float N = roundf((float)(sgn * kPi + phi) * kInvTwoPi / n);
    return 2.f * sqr(cosf(kPi * n * N - phi / 2.f));
}

#if defined(WT_SECOND_SOURCE) && !defined(__HIP_DEVICE_COMPILE__)
// (oracle/indep/prims2.cpp, see the note at WT_SS_ACTIVE in wt/cone.h: the wedge's soft / hard diffraction coefficients from the textbook form of
// Kouyoumjian & Pathak in f64, the transition function from a quadrature of its defining Fresnel integral — no erfc series, no asymptote)
extern "C" void ss_wedge_utd(double n, double k_Li, double k_ro, double sin_beta, double phii, double phio, double out_DsDh[4]);
#define WT_SS_UTD 1
#else
#define WT_SS_UTD 0
#endif
// utd.hpp:36-57.  |x| < 6: (1+i) sqrt(pi/2) sqrt|x| e^{i|x|} erfc(e^{i pi/4} sqrt|x|); otherwise the 4-term asymptote.
WT_HD cplx utd_F(float x) {
    const float absx = fabsf(x);
    cplx result;
    if (absx < 6.f) {
        const double ax = (double)absx;
        const double s = sqrt(ax);
        // erf(z) = 2/sqrt(pi) z sum_n (-z^2)^n / (n! (2n+1)),  z = e^{i pi/4} s,  -z^2 = -i ax
        double tr = 1.0, ti = 0.0;   // (-i ax)^n / n!
        double sr = 1.0, si = 0.0;   // running sum
        for (int n = 1; n < 64; ++n) {
            // multiply by (-i ax)/n : (tr + i ti)(-i a) = ti a - i tr a
            const double a = ax / (double)n;
            const double nr = ti * a, ni = -tr * a;
            tr = nr;
            ti = ni;
            const double inv = 1.0 / (double)(2 * n + 1);
            sr += tr * inv;
            si += ti * inv;
            if (fabs(tr) + fabs(ti) < 1e-18) break;
        }
        const double c = 0.70710678118654752440 * s * 1.12837916709551257390;   // Re z = Im z = s/sqrt2, times 2/sqrt(pi)
        // erf = (c + i c)(sr + i si)
        const double er = c * (sr - si), ei = c * (sr + si);
        const double cr = 1.0 - er, ci = -ei;   // erfc
        // (1+i) * sqrt(pi/2) * s * e^{i ax} * erfc
        const double m = 1.25331413731550025121 * s;
        const double e_r = cos(ax), e_i = sin(ax);
        const double pr = e_r * cr - e_i * ci, pi_ = e_r * ci + e_i * cr;
        result = cplx{(float)(m * (pr - pi_)), (float)(m * (pr + pi_))};
    } else {
        const float r = 1.f / (2.f * absx);
        const float r2 = r * r, r3 = r2 * r, r4 = r2 * r2;
        result = cplx{1.f - 3.f * r2 + 75.f * r4, r - 15.f * r3};
    }
    return x < 0.f ? conj(result) : result;
}

// fsd/common.hpp:42-88
struct utd_wedge_t {
    vec3 v;       // mid point of the (clamped) edge
    float l;      // length
    vec3 nff, tff, nbf;
    float alpha;
    uint32_t edge;
};
WT_HD vec3 wedge_e(const utd_wedge_t& w) { return cross(w.nff, w.tff); }

struct utd_ret_t {
    cplx Ds, Dh;
};

// wedge_edge_t::diffraction_point(src, dst) (utd.hpp:62-78)
WT_HD bool wedge_diffraction_point(const utd_wedge_t& w, vec3 src, vec3 dst, vec3& p) {
    const vec3 e = wedge_e(w);
    const float sl = length(vec2{dot(src - w.v, w.tff), dot(src - w.v, w.nff)});
    const float dl = length(vec2{dot(dst - w.v, w.tff), dot(dst - w.v, w.nff)});
    const float dist = dot(e, src - w.v) + dot(dst - src, e) * sl / (sl + dl);
    if (fabsf(dist) > w.l / 2.f) return false;
    p = w.v + e * dist;
    if (veq(p, src) || veq(p, dst)) return false;
    return true;
}
// wedge_edge_t::diffraction_point(src, wo) (utd.hpp:83-107)
WT_HD bool wedge_diffraction_point_dir(const utd_wedge_t& w, vec3 src, vec3 wo, vec3& p) {
    const vec3 e = wedge_e(w);
    const float cos_beta = dot(wo, e);
    const float sin_beta = sqrtf(fmaxf_(0.f, 1.f - sqr(cos_beta)));
    if (sin_beta < kUtdMinSinBeta) return false;
    const float sl = length(vec2{dot(src - w.v, w.tff), dot(src - w.v, w.nff)});
    const vec3 prj_src = w.v + dot(src - w.v, e) * e;
    p = prj_src + sl * (cos_beta / sin_beta) * e;
    if (length2(p - w.v) > sqr(w.l / 2.f)) return false;
    if (veq(p, src)) return false;
    return true;
}

// wedge_edge_t::UTD (utd.hpp:112-171); the s/h frames of UTD_ret_t are not consumed by plt_path (TODO: polarization there)
WT_HD utd_ret_t wedge_UTD(const utd_wedge_t& w, float k, vec3 wi, vec3 wo, float ro) {
    const vec3 e = wedge_e(w);
    const float n = 2.f - w.alpha * kInvPi;

    const float sin_beta2 = fmaxf_(0.f, 1.f - sqr(dot(wi, e)));
    const float sin_beta = sqrtf(sin_beta2);
    const float phii = atan2f(dot(w.nff, wi), dot(w.tff, wi));
    const float phio = atan2f(dot(w.nff, wo), dot(w.tff, wo));

    const float Li = ro * sin_beta2;
    const float kLi = k_times_len(k, Li);

#if WT_SS_UTD
    {
        const float t1s = modf_floor(phii + phio, kPi2), t2s = modf_floor(phii - phio, kPi2);   // (the reference's exclusion of the grazing directions, as below)
        if (fabsf(t1s) < 1e-5f || fabsf(t2s) < 1e-5f) return utd_ret_t{cplx{0.f, 0.f}, cplx{0.f, 0.f}};
        double o[4];
        ss_wedge_utd((double)n, (double)kLi, (double)k_times_len(k, ro), (double)sin_beta, (double)phii, (double)phio, o);
        return utd_ret_t{cplx{(float)o[0], (float)o[1]}, cplx{(float)o[2], (float)o[3]}};
    }
#endif
    const float a1 = utd_a<+1>(phii - phio, n);
    const float a2 = utd_a<-1>(phii - phio, n);
    const float a3 = utd_a<+1>(phii + phio, n);
    const float a4 = utd_a<-1>(phii + phio, n);
    const cplx F1 = utd_F(kLi * a1);
    const cplx F2 = utd_F(kLi * a2);
    const cplx F3 = utd_F(kLi * a3);
    const cplx F4 = utd_F(kLi * a4);
    const cplx D1 = (-1.f / tanf((kPi + (phii - phio)) / (2.f * n))) * F1;
    const cplx D2 = (-1.f / tanf((kPi - (phii - phio)) / (2.f * n))) * F2;
    const cplx D3 = (-1.f / tanf((kPi + (phii + phio)) / (2.f * n))) * F3;
    const cplx D4 = (-1.f / tanf((kPi - (phii + phio)) / (2.f * n))) * F4;

    const float kro = k_times_len(k, ro);
    const float Dm = 1.f / (2.f * n * sqrtf(kro) * sin_beta) * 0.39894228040143267794f;
    const cplx D = cpolar(Dm, -kPi4);

    const float t1 = modf_floor(phii + phio, kPi2);
    const float t2 = modf_floor(phii - phio, kPi2);
    const bool zero = fabsf(t1) < 1e-5f || fabsf(t2) < 1e-5f;
    const cplx Ds = zero ? cplx{0.f, 0.f} : D1 + D2 - (D3 + D4);
    const cplx Dh = zero ? cplx{0.f, 0.f} : D1 + D2 + (D3 + D4);
    return utd_ret_t{-(D * Ds), -(D * Dh)};
}

// intersect_edge_ellipsoid (math/intersect/misc.hpp:40-72); {0,0} when there is no intersection
WT_HD vec2 intersect_edge_ellipsoid(vec3 point0, vec3 point1, vec3 centre, vec3 x, vec3 y, vec3 axes) {
    const vec3 z = cross(x, y);
    point0 = point0 - centre;
    point1 = point1 - centre;
    const vec3 p0 = vec3{dot(point0, x) / axes.x, dot(point0, y) / axes.y, dot(point0, z) / axes.z};
    const vec3 p1 = vec3{dot(point1, x) / axes.x, dot(point1, y) / axes.y, dot(point1, z) / axes.z};
    const vec3 d = p1 - p0;
    const float a = dot(d, d);
    const float b = dot(p0, d) * 2.f;
    const float c = dot(p0, p0) - 1.f;
    const float det2 = b * b - 4.f * a * c;
    if (det2 <= 0.f || a == 0.f) return vec2{0.f, 0.f};
    const float recp_a = 1.f / a;
    const float det = sqrtf(det2);
    float t1 = .5f * (-b - signf(b) * det) * recp_a;
    float t2 = t1 == 0.f ? -b * recp_a : c * recp_a / t1;
    if (t1 > t2) {
        const float t = t1;
        t1 = t2;
        t2 = t;
    }
    return vec2{t1, t2};
}

// ---- aperture -------------------------------------------------------------------------------------------------------
struct utd_edge_rec_t {
    uint32_t id_front;   // edge id | (face 1 is the front face) << 31
    float t1, t2;        // clamped parameters of the part of the edge inside the interaction region
};
struct utd_edges_ref_t {
    utd_edge_rec_t* p;
    size_t stride;
    WT_HD utd_edge_rec_t& operator[](uint32_t i) const { return p[i * stride]; }
};
struct utd_aperture_t {
    uint32_t n_edges;
    uint32_t overflow;
    float k;
    vec3 interaction_wp;
    uint32_t edge_offset, edge_cap;   // this aperture's wedge records: [edge_offset, edge_offset + edge_cap) of the caller's record array
};

WT_HD utd_wedge_t utd_wedge(const scene_t& sc, const utd_edge_rec_t& r) {
    const uint32_t id = r.id_front & 0x7FFFFFFFu;
    const bool f1 = (r.id_front >> 31) != 0;
    const edge_t ed = sc.edges[id];
    const vec3 v1 = mix3(ed.a, ed.b, r.t1), v2 = mix3(ed.a, ed.b, r.t2);
    utd_wedge_t w;
    w.v = (v1 + v2) / 2.f;
    w.l = length(v2 - v1);
    w.nff = f1 ? ed.n1 : ed.n2;
    w.tff = f1 ? ed.t1 : ed.t2;
    w.nbf = f1 ? ed.n2 : ed.n1;
    w.alpha = ed.alpha;
    w.edge = id;
    return w;
}

// free_space_diffraction_t ctor (free_space_diffraction.cpp:23-79).  `emit(record)` receives the wedges in edge order; returns their number.
template <class EdgeIds, class Emit>
WT_HD uint32_t utd_aperture_wedges(const scene_t& sc, vec3 interaction_wp, const frame_t& region_frame, vec3 region_size, vec3 wi, const EdgeIds& edge_ids,
                                   uint32_t n_edge_ids, Emit&& emit) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < n_edge_ids; ++i) {
        const uint32_t id = edge_ids[i];
        const edge_t ed = sc.edges[id];
        const bool f1_is_front = dot(wi, ed.n1) > 0.f;
        const vec3 nff = f1_is_front ? ed.n1 : ed.n2;
        // light incident from inside the wedge?
        if (dot(wi, nff) <= 0.f) continue;
        float t1 = 0.f, t2 = 1.f;
        if (vfinite(region_size)) {
            const vec2 t = intersect_edge_ellipsoid(ed.a, ed.b, interaction_wp, region_frame.t, region_frame.b, region_size);
            t1 = clamp01(t.x);
            t2 = clamp01(t.y);
        }
        const vec3 v1 = mix3(ed.a, ed.b, t1), v2 = mix3(ed.a, ed.b, t2);
        if (veq(v1, v2)) continue;
        emit(utd_edge_rec_t{id | (f1_is_front ? 0x80000000u : 0u), t1, t2});
        ++n;
    }
    return n;
}
// the number of wedges utd_build_aperture will store: sizes the aperture's storage (device: a bump allocation from the round's pool)
template <class EdgeIds>
WT_HD uint32_t utd_count_wedges(const scene_t& sc, vec3 interaction_wp, const frame_t& region_frame, vec3 region_size, vec3 wi, const EdgeIds& edge_ids,
                                uint32_t n_edge_ids) {
    return utd_aperture_wedges(sc, interaction_wp, region_frame, region_size, wi, edge_ids, n_edge_ids, [](const utd_edge_rec_t&) {});
}
// `out`: the aperture's records (ap.edge_cap of them; wedges beyond are counted in ap.overflow — cannot happen after utd_count_wedges)
template <class EdgeIds>
WT_HD void utd_build_aperture(const scene_t& sc, vec3 interaction_wp, const frame_t& region_frame, vec3 region_size, vec3 wi, float k, const EdgeIds& edge_ids,
                              uint32_t n_edge_ids, utd_aperture_t& ap, const utd_edges_ref_t& out) {
    ap.n_edges = 0;
    ap.overflow = 0;
    ap.k = k;
    ap.interaction_wp = interaction_wp;
    utd_aperture_wedges(sc, interaction_wp, region_frame, region_size, wi, edge_ids, n_edge_ids, [&](const utd_edge_rec_t& r) {
        if (ap.n_edges < ap.edge_cap)
            out[ap.n_edges++] = r;
        else
            ap.overflow++;
    });
}

// free_space_diffraction_t::pdf (free_space_diffraction.cpp:155-195): angle density [1/rad]
WT_HD float utd_pdf(const scene_t& sc, const utd_aperture_t& ap, const utd_edges_ref_t& edges, vec3 src, vec3 wo) {
    if (ap.n_edges == 0) return 0.f;
    float ret = 0.f;
    for (uint32_t i = 0; i < ap.n_edges; ++i) {
        const utd_wedge_t w = utd_wedge(sc, edges[i]);
        vec3 p;
        if (!wedge_diffraction_point_dir(w, src, wo, p)) continue;
        const vec3 ui = src - p;
        if ((dot(wo, w.nff) <= 0.f && dot(wo, w.nbf) <= 0.f) || (dot(ui, w.nff) <= 0.f && dot(ui, w.nbf) <= 0.f)) continue;
        const float ri = length(ui);
        const vec3 wi = ui / ri;
        const float phii = atan2f(dot(w.nff, wi), dot(w.tff, wi));
        const float phio = atan2f(dot(w.nff, wo), dot(w.tff, wo));
        const float sigma = sqrtf(kUtdIsSigmaScale / k_times_len(ap.k, ri));
        const float mean_phi1 = kPi + phii, mean_phi2 = kPi - phii;
        float x1 = fabsf(modf_floor(phio - mean_phi1, kTwoPi));
        float x2 = fabsf(modf_floor(phio - mean_phi2, kTwoPi));
        if (x1 > kPi) x1 -= kTwoPi;
        if (x2 > kPi) x2 -= kTwoPi;
        // angle density along the Keller cone
        const float apd = 0.39894228040143267794f / sigma * (expf(-.5f * sqr(x1 / sigma)) + expf(-.5f * sqr(x2 / sigma))) / 2.f;
        ret += apd;
    }
    return ret / (float)(ap.n_edges + 1);
}

struct utd_sample_t {
    vec3 wo;
    float weight;
    uint32_t is_direct;
};
// free_space_diffraction_t::sample (free_space_diffraction.cpp:81-153).  A rejected sample is the reference's value-initialised
// sample_ret_t: wo = (0,0,1), weight = 0.
WT_HD utd_sample_t utd_sample(const scene_t& sc, const utd_aperture_t& ap, const utd_edges_ref_t& edges, vec3 src, sampler_t& smp) {
    const utd_sample_t none{vec3{0.f, 0.f, 1.f}, 0.f, 0u};
    const int n = (int)ap.n_edges;
    // sampler_t::uniform_int_interval(0, n+1) (sampler.hpp:112-115)
    int eidx = (int)(sampler_r(smp) * (float)(n + 1));
    if (eidx > n) eidx = n;
    if (eidx == n) {
        // sampled the direct term
        const vec3 wi = normalize(src - ap.interaction_wp);
        return utd_sample_t{-wi, (float)(n + 1), 1u};
    }
    const utd_wedge_t w = utd_wedge(sc, edges[(uint32_t)eidx]);
    const vec3 e = wedge_e(w);
    const vec3 p = w.v + (sampler_r(smp) - .5f) * w.l * e;
    const vec3 ui = src - p;
    if (dot(ui, w.nff) <= 0.f && dot(ui, w.nbf) <= 0.f) return none;
    const float ri = length(ui);
    const vec3 wi = ui / ri;
    const float phii = atan2f(dot(w.nff, wi), dot(w.tff, wi));
    const float sigma = sqrtf(kUtdIsSigmaScale / k_times_len(ap.k, ri));
    const float sample = sigma * normal2d(sampler_r2(smp)).x;
    const float mean_phi1 = kPi + phii, mean_phi2 = kPi - phii;
    const float phio = (sampler_r(smp) < .5f ? mean_phi1 : mean_phi2) + sample;
    const float cos_beta = dot(wi, e);
    const float sin_beta = sqrtf(fmaxf_(0.f, 1.f - sqr(cos_beta)));
    const vec3 wo = sin_beta * (cosf(phio) * w.tff + sinf(phio) * w.nff) - cos_beta * e;
    if (dot(wo, w.nff) <= 0.f && dot(wo, w.nbf) <= 0.f) return none;
    if (sin_beta < kUtdMinSinBeta) return none;
    const float dpd = utd_pdf(sc, ap, edges, src, wo);
    if (dpd == 0.f) return none;
    return utd_sample_t{wo, 1.f / dpd, 0u};
}

// One diffracting edge of free_space_diffraction_t::f (free_space_diffraction.cpp:197-234)
struct utd_diffracting_edge_t {
    utd_ret_t utd;
    uint32_t edge;
    vec3 p;
    float ri, ro;
};
WT_HD bool utd_f_edge(const scene_t& sc, const utd_aperture_t& ap, const utd_edge_rec_t& rec, vec3 src, vec3 dst, utd_diffracting_edge_t& out) {
    const utd_wedge_t w = utd_wedge(sc, rec);
    vec3 p;
    if (!wedge_diffraction_point(w, src, dst, p)) return false;
    const vec3 ui = src - p, uo = dst - p;
    // ignore into-wedge rays
    if ((dot(uo, w.nff) <= 0.f && dot(uo, w.nbf) <= 0.f) || (dot(ui, w.nff) <= 0.f && dot(ui, w.nbf) <= 0.f)) return false;
    const float ri = length(ui), ro = length(uo);
    const vec3 wi = ui / ri, wo = uo / ro;
    const utd_ret_t f = wedge_UTD(w, ap.k, wi, wo, ro);
    if (ceq(f.Dh, cplx{0.f, 0.f}) && ceq(f.Ds, cplx{0.f, 0.f})) return false;
    out.utd = f;
    out.edge = w.edge;
    out.p = p;
    out.ri = ri;
    out.ro = ro;
    return true;
}

}   // namespace wt
