// wave_tracer_amd — Fraunhofer free-space diffraction (FSD) BSDF: aperture construction, ASF evaluation, LUT
// importance sampling with rejection.
//
// Reference: include/wt/interaction/fsd/fraunhofer/fsd.hpp:20-185,
//            include/wt/interaction/fsd/fraunhofer/free_space_diffraction.hpp:34-135,
//            src/interaction/fsd/fraunhofer/free_space_diffraction.cpp:22-129,
//            src/interaction/fsd/fraunhofer/fsd_sampler.cpp:38-156,
//            include/wt/interaction/fsd/fraunhofer/fsd_lut.hpp:25-86.
//
// An aperture is a header + a bounded array of edge segments.  In the reference the per-edge amplitudes are
// complex numbers whose imaginary (a_b) resp. real (iab_2) parts are identically zero (free_space_diffraction.cpp:
// 66,72: `ca = a`), so they are stored as two real numbers here.
#pragma once
#include "beam.h"
#include "gauss.h"
#include "rng.h"

namespace wt {

constexpr float kFsdPA1 = 0.0049361075794549872500f;
constexpr float kFsdPA2 = 0.21899789398059305541f;
constexpr float kFsdP0Sigma = 0.288675134594813f / 4.f;
constexpr float kFsdUnitM = 1e-3f;       // fsd_unit = 1 mm
constexpr float kFsdWo2Cutoff = .85f;
constexpr uint32_t kFsdMaxEdges = 4096;   // segments per aperture: effectively unbounded, like the reference's std::vector (overflow is counted)
// An aperture is DEAD when the coherent sum of its segments' amplitudes vanishes against their incoherent sum (ratio of the powers at the
// eight probe directions of the 0-th order estimate below this): a doubled scene edge — the rim of a thin plate modelled by two coincident
// faces — enters as two segment chains of opposite direction whose amplitudes cancel to rounding (measured ratios 1e-13..1e-14; a real slit
// would have to be narrower than ~50 nm to come near the threshold).  Its scattering function is rounding noise, the acceptance test
// u g < f / n of the rejection sampler cannot pass, and the reference's loop spins through all n x 1024 tries before it reports failure:
// in the headline workload 13 % of the apertures with 8-15 segments, 85 % of all tries of the pass (round 4, oracle fsd histogram).
// fsd_max_tries is 0 for such an aperture: the same outcome (sample failed, walk ends) without the tries.
constexpr float kFsdDeadRatio = 1e-10f;
#if !defined(__HIPCC__)
inline float g_fsd_dead_ratio = kFsdDeadRatio;   // CPU checker only: 0 switches the classification off (tools/fsd_dead_effect.py measures what it changes)
#define WT_FSD_DEAD_RATIO ::wt::g_fsd_dead_ratio
#else
#define WT_FSD_DEAD_RATIO ::wt::kFsdDeadRatio
#endif

struct fsd_edge_t {
    vec2 e, v;    // edge vector, mid point (in fsd units = mm)
    float ab;     // a_b   = ca - cb           (real)
    float iab;    // iab_2 = i * (ca + cb)/2   -> stores (ca+cb)/2
    float pdf;
};
struct fsd_aperture_t {
    uint32_t n_edges;
    float P0, P0_pdf, psi02, recp_I;
    float k;
    frame_t frame;
    uint32_t overflow;
    uint32_t edge_offset, edge_cap;   // this aperture's segment records: [edge_offset, edge_offset + edge_cap) of the pool's edge array
    uint32_t dead;                    // the segments' amplitudes cancel identically: the rejection loop cannot accept (fsd_build_finish)
};
// edges of one aperture: contiguous AoS (an aperture is read many times by the one lane that owns it)
struct fsd_edges_ref_t {
    fsd_edge_t* p;
    uint32_t stride;   // unused (kept 1)
    WT_HD fsd_edge_t get(uint32_t i) const { return p[i]; }
    WT_HD void set(uint32_t i, const fsd_edge_t& e) const { p[i] = e; }
};

WT_HD float fsd_alpha1(float x, float y) { return x == 0.f ? 0.f : kInvTwoPi * y / (x * (x * x + y * y)) * (cosf(x / 2.f) - sincf_(x / 2.f)); }
WT_HD float fsd_alpha2(float x, float y) { return x == 0.f ? 0.f : kInvTwoPi * y / (x * x + y * y) * sincf_(x / 2.f); }
#if defined(WT_SECOND_SOURCE) && !defined(__HIP_DEVICE_COMPILE__)
// (oracle/indep/prims2.cpp, see the note at WT_SS_ACTIVE in wt/cone.h: the two amplitudes of a segment — a_b alpha_1 and iab_2 alpha_2 — from a
// quadrature in f64 of the boundary line integral they are the closed form of)
extern "C" void ss_fraunhofer_segment(const float e[2], float ab, float iab, const float xi[2], float out_a1_a2[2]);
#define WT_SS_FSD 1
#else
#define WT_SS_FSD 0
#endif
WT_HD float fsd_chi_e(vec2 xi) {
    const float chi = 0.830092714835359f;
    const float t = 1.f + chi * dot(xi, xi);
    const float t2 = t * t, t3 = t2 * t;
    return fmaxf_(0.f, 1.f - (3.f / t2 - 2.f / t3));
}
WT_HD float fsd_chi_0(vec2 xi) {
    xi = xi / kFsdP0Sigma;
    return expf(-.5f * dot(xi, xi));
}
// zeta = xi * Xi, Xi = mat2(e, m), m = (e.y,-e.x)   (vec * mat: (dot(xi,e), dot(xi,m)))
WT_HD vec2 fsd_zeta(const fsd_edge_t& e, vec2 xi) { return {dot(xi, e.e), xi.x * e.e.y - xi.y * e.e.x}; }
WT_HD cplx fsd_Psi(const fsd_edge_t& e, vec2 xi) {
#if WT_SS_FSD
    float a12[2];
    ss_fraunhofer_segment(&e.e.x, e.ab, e.iab, &xi.x, a12);
    const float a1 = a12[0], a2 = a12[1];
#else
    const vec2 z = fsd_zeta(e, xi);
    const float a1 = e.ab * fsd_alpha1(z.x, z.y);
    const float a2 = e.iab * fsd_alpha2(z.x, z.y);
#endif
    const float ee2 = length2(e.e);
    return cpolar(ee2, -dot(e.v, xi)) * cplx{a1, a2};
}
WT_HD float fsd_Psi2(const fsd_edge_t& e, vec2 xi) {
#if WT_SS_FSD
    float a12[2];
    ss_fraunhofer_segment(&e.e.x, e.ab, e.iab, &xi.x, a12);
    const float a1 = a12[0], a2 = a12[1];
#else
    const vec2 z = fsd_zeta(e, xi);
    const float a1 = e.ab * fsd_alpha1(z.x, z.y);
    const float a2 = e.iab * fsd_alpha2(z.x, z.y);
#endif
    return sqr(length2(e.e)) * (a1 * a1 + a2 * a2);
}
WT_HD float fsd_ASF_unclamped(const fsd_aperture_t& ap, const fsd_edges_ref_t& ed, vec2 xi) {
    cplx amp{0.f, 0.f};
    for (uint32_t i = 0; i < ap.n_edges; ++i) amp = amp + fsd_Psi(ed.get(i), xi);
    return cnorm(amp);
}
WT_HD float fsd_ASF(const fsd_aperture_t& ap, const fsd_edges_ref_t& ed, vec2 xi) {
    return fsd_ASF_unclamped(ap, ed, xi) * fsd_chi_e(xi) + ap.psi02 * fsd_chi_0(xi);
}
WT_HD float fsd_sampling_density(const fsd_aperture_t& ap, const fsd_edges_ref_t& ed, vec2 xi) {
    float d = 0.f;
    for (uint32_t i = 0; i < ap.n_edges; ++i) d += fsd_Psi2(ed.get(i), xi);
    return d * fsd_chi_e(xi) + ap.P0 * kInvTwoPi / sqr(kFsdP0Sigma) * fsd_chi_0(xi);
}
WT_HD float fsd_Pj(const fsd_edge_t& e) {
    const float l4 = sqr(length2(e.e));
    return l4 * kFsdPA1 * sqr(e.ab) + l4 * kFsdPA2 * sqr(e.iab);
}

// free_space_diffraction_t ctor (free_space_diffraction.cpp:22-129).
// `edge_ids`: the (deduplicated, sorted) ADS edge ids of the interaction region.  `sigma` = wavefront std-dev.
struct fsd_build_state_t {
    vec2 cse;   // wave_function.envelope()
    float max_edge_length;
    float P_total;
};
WT_HD fsd_build_state_t fsd_build_begin(const frame_t& frame, float k, float total_power, vec2 sigma, fsd_aperture_t& ap) {
    ap.k = k;
    ap.frame = frame;
    ap.n_edges = 0;
    ap.overflow = 0;
    ap.dead = 0;
    ap.recp_I = total_power > 0.f ? 1.f / total_power : 0.f;
    fsd_build_state_t st;
    st.cse = sigma * kBeamEnvelope;
    st.max_edge_length = .33f * fmaxf_(st.cse.x, st.cse.y);
    st.P_total = 0.f;
    return st;
}
// The aperture segments one scene edge contributes, in order, each passed to emit(fe) (free_space_diffraction.cpp:47-104): only
// silhouette edges, clipped to the beam's envelope ellipse, cut into pieces no longer than a third of the envelope radius.
template <class Emit>
WT_HD void fsd_edge_segments(const scene_t& sc, const frame_t& frame, const cone_t& beam, vec2 sigma, vec2 cse, float max_edge_length, uint32_t edge_id,
                             Emit&& emit) {
    const edge_t edge = sc.edges[edge_id];
    // only the projected silhouette
    if (dot(beam.d, edge.n1) * dot(beam.d, edge.n2) >= 0.f) return;
    const vec3 la = to_local(frame, edge.a - beam.o), lb = to_local(frame, edge.b - beam.o);
    const vec2 u1{la.x, la.y}, u2{lb.x, lb.y};
    float t1 = 0.f, t2 = 1.f;
    const vec2 q1 = u1 / cse, q2 = u2 / cse;
    if (!(dot(q1, q1) <= 1.f) || !(dot(q2, q2) <= 1.f)) {
        const edge_ellipse_t intr = intersect_edge_ellipse(u1, u2, cse.x, cse.y);
        if (intr.points == 0) return;
        t1 = fmaxf_(0.f, intr.t1);
        t2 = fminf_(1.f, intr.t2);
    }
    const float len = length(mix2(u1, u2, t1) - mix2(u1, u2, t2));
    // max(1, int(round(len/max_edge_length) + .5))
    int segments = (int)(roundf(len / max_edge_length) + .5f);
    if (segments < 1) segments = 1;
    const float seg = 1.f / float(segments);
    vec2 v1 = mix2(u1, u2, t1);
    float a = sqrtf(wavefront_intensity(sigma, v1));
    for (int i = 0; i < segments; ++i) {
        const float tt = mixf(t1, t2, float(i + 1) * seg);
        const vec2 v2 = mix2(u1, u2, tt);
        const float b = sqrtf(wavefront_intensity(sigma, v2));
        if (a > 0.f || b > 0.f) {
            fsd_edge_t fe;
            fe.v = ((v1 + v2) / 2.f) / kFsdUnitM;
            fe.e = (v2 - v1) / kFsdUnitM;
            fe.ab = a - b;
            fe.iab = (a + b) / 2.f;
            fe.pdf = fsd_Pj(fe);
            if (fe.pdf > 0.f) emit(fe);
        }
        v1 = v2;
        a = b;
    }
}
// Upper bound of the number of segments fsd_edge_segments emits for one scene edge (its geometry part: silhouette test, clipping,
// subdivision; segments whose beam amplitude vanishes are skipped there).  Sizes an aperture's storage before it is built.
WT_HD uint32_t fsd_count_segments(const scene_t& sc, const frame_t& frame, const cone_t& beam, vec2 cse, float max_edge_length, uint32_t edge_id) {
    const edge_t edge = sc.edges[edge_id];
    if (dot(beam.d, edge.n1) * dot(beam.d, edge.n2) >= 0.f) return 0;
    const vec3 la = to_local(frame, edge.a - beam.o), lb = to_local(frame, edge.b - beam.o);
    const vec2 u1{la.x, la.y}, u2{lb.x, lb.y};
    float t1 = 0.f, t2 = 1.f;
    const vec2 q1 = u1 / cse, q2 = u2 / cse;
    if (!(dot(q1, q1) <= 1.f) || !(dot(q2, q2) <= 1.f)) {
        const edge_ellipse_t intr = intersect_edge_ellipse(u1, u2, cse.x, cse.y);
        if (intr.points == 0) return 0;
        t1 = fmaxf_(0.f, intr.t1);
        t2 = fminf_(1.f, intr.t2);
    }
    const float len = length(mix2(u1, u2, t1) - mix2(u1, u2, t2));
    const int segments = (int)(roundf(len / max_edge_length) + .5f);
    return segments < 1 ? 1u : (uint32_t)segments;
}
WT_HD void fsd_build_add_edge(const scene_t& sc, const frame_t& frame, const cone_t& beam, vec2 sigma, fsd_build_state_t& st, uint32_t edge_id,
                              fsd_aperture_t& ap, const fsd_edges_ref_t& ed) {
    fsd_edge_segments(sc, frame, beam, sigma, st.cse, st.max_edge_length, edge_id, [&](const fsd_edge_t& fe) {
        if (ap.n_edges < ap.edge_cap) {
            ed.set(ap.n_edges++, fe);
            st.P_total += fe.pdf;
        } else
            ap.overflow++;
    });
}
WT_HD void fsd_build_finish(float k, fsd_build_state_t& st, fsd_aperture_t& ap, const fsd_edges_ref_t& ed);
template <class EdgeIdList>
WT_HD void fsd_build_aperture(const scene_t& sc, const frame_t& frame, float k, float total_power, const cone_t& beam, const EdgeIdList& edge_ids,
                              uint32_t n_edge_ids, vec2 sigma, fsd_aperture_t& ap, const fsd_edges_ref_t& ed) {
    fsd_build_state_t st = fsd_build_begin(frame, k, total_power, sigma, ap);
    for (uint32_t ei = 0; ei < n_edge_ids; ++ei) fsd_build_add_edge(sc, frame, beam, sigma, st, edge_ids[ei], ap, ed);
    fsd_build_finish(k, st, ap, ed);
}
WT_HD void fsd_build_finish(float k, fsd_build_state_t& st, fsd_aperture_t& ap, const fsd_edges_ref_t& ed) {
    float P_total = st.P_total;
    // power in the 0-th order lobe (8-point average on a circle of radius 3*P0_sigma)
    const float psi0r = 3.f * kFsdP0Sigma;
    const vec2 dirs[8] = {{-kInvSqrt2, -kInvSqrt2}, {-1, 0}, {-kInvSqrt2, kInvSqrt2}, {0, 1}, {kInvSqrt2, kInvSqrt2}, {1, 0}, {kInvSqrt2, -kInvSqrt2}, {0, -1}};
    float acc = 0.f, inc = 0.f;
    for (int i = 0; i < 8; ++i) {   // fsd_ASF_unclamped at the probe direction, and the incoherent sum of the same terms beside it
        cplx amp{0.f, 0.f};
        for (uint32_t j = 0; j < ap.n_edges; ++j) {
            const cplx psi = fsd_Psi(ed.get(j), psi0r * dirs[i]);
            amp = amp + psi;
            inc += cnorm(psi);
        }
        acc += cnorm(amp);
    }
    ap.psi02 = acc / 8.f;
    ap.dead = (ap.n_edges >= 2 && acc < WT_FSD_DEAD_RATIO * inc) ? 1u : 0u;
    ap.P0 = (kTwoPi * sqr(kFsdP0Sigma) * ap.psi02) / sqr(k * 1.f);   // k [1/mm] * fsd_unit [mm]
    P_total += ap.P0;
    if (P_total > 0.f) {
        const float rp = 1.f / P_total;
        ap.P0_pdf = ap.P0 * rp;
        for (uint32_t i = 0; i < ap.n_edges; ++i) {
            fsd_edge_t e = ed.get(i);
            e.pdf *= rp;
            ed.set(i, e);
        }
    } else {
        ap.P0_pdf = 1.f;
        ap.n_edges = 0;
    }
}

// ---- LUT sampling (fsd_lut.hpp:36-69) ------------------------------------------------------------
WT_HD float fsd_lut_lerp1(float x, const float* tbl, uint32_t S) {
    x *= float(S - 1);
    uint32_t l = (uint32_t)x;
    if (l > S - 1) l = S - 1;
    const uint32_t h = l + 1 < S ? l + 1 : S - 1;
    const float f = fractf(x);
    return f * tbl[h] + (1.f - f) * tbl[l];
}
WT_HD float fsd_lut_lerp2(float x, float rx, const float* tbl, uint32_t S) {
    x *= float(S - 1);
    uint32_t l = (uint32_t)x;
    if (l > S - 1) l = S - 1;
    const uint32_t h = l + 1 < S ? l + 1 : S - 1;
    const float f = fractf(x);
    return f * fsd_lut_lerp1(rx, tbl + (size_t)h * S, S) + (1.f - f) * fsd_lut_lerp1(rx, tbl + (size_t)l * S, S);
}
WT_HD vec2 fsd_lut_sample(const fsd_lut_t& lut, vec3 rand3, bool a1) {
    const float* it = a1 ? lut.icdf_theta1 : lut.icdf_theta2;
    const float* ic = a1 ? lut.icdf1 : lut.icdf2;
    const float theta = fsd_lut_lerp1(rand3.x, it, lut.n_theta);
    const float theta_fract = theta * 2.f / kPi;
    const float r = fmaxf_(0.f, fsd_lut_lerp2(theta_fract, rand3.y, ic, lut.m));
    vec2 zeta = r * vec2{cosf(theta), sinf(theta)};
    int q = (int)(rand3.z * 4.f);
    if (q > 3) q = 3;
    zeta.x *= (((q + 1) / 2) % 2 == 0) ? 1.f : -1.f;
    zeta.y *= ((q / 2) % 2 == 0) ? 1.f : -1.f;
    return zeta;
}
// sampleN (fsd_sampler.cpp:55-70)
WT_HD vec2 fsd_sampleN(const scene_t& sc, const fsd_aperture_t& ap, const fsd_edges_ref_t& ed, sampler_t& sampler) {
    // sampler.discrete<true>(n+1, ...): one uniform, linear scan
    const float p = sampler_r(sampler);
    float cdf = 0.f;
    uint32_t sel = ap.n_edges;   // last
    for (uint32_t i = 0; i < ap.n_edges; ++i) {
        const float ep = i == 0 ? ap.P0_pdf : ed.get(i - 1).pdf;
        cdf += ep;
        if (p < cdf) {
            sel = i;
            break;
        }
    }
    if (sel == 0) return kFsdP0Sigma * normal2d(sampler_r2(sampler));
    const fsd_edge_t e = ed.get(sel - 1);
    // sample1: pick alpha1 or alpha2 lobe  (discrete<2>({A,B}), un-normalised)
    const float A = sqr(e.ab), B = sqr(e.iab);
    const float pp = sampler_r(sampler) * (A + B);
    const bool use_a1 = pp < A;
    const vec2 zeta = fsd_lut_sample(sc.lut, sampler_r3(sampler), use_a1);
    // zeta * inverse(Xi)
    const mat2 Xi = mkmat2(e.e, vec2{e.e.y, -e.e.x});
    return mul(zeta, inverse(Xi));
}

#ifdef WT_PROFILE_CONE_TRI
inline unsigned long long g_fsd_hist[8][24] = {{0}};
#endif
struct fsd_sample_t {
    vec3 wo;
    float dpd;   // solid-angle density (0: failed)
    float weight;
};
// ---- fsd_sampler_t::sample (rejection, fsd_sampler.cpp:72-110) + free_space_diffraction_t::sample ------------------------
// Random-number layout (ours; the reference draws from a sequential engine): try t of a rejection loop owns the kFsdDrawsPerTry
// draws starting at  base + t*kFsdDrawsPerTry  of the walk's Philox stream (base = the stream position at entry, rounded up to
// a multiple of 4; a try uses 4 or 6 draws).  Tries are therefore independent of each other: the CPU checker runs them in
// order, the device runs the first few per lane and the rest 64 at a time across a wavefront (wtgpu.hip: k_interact), and both
// accept the same (lowest) try.  After the loop the stream continues behind the last try that was looked at.
constexpr uint32_t kFsdDrawsPerTry = 8;
constexpr uint32_t kFsdInlineTries = 8;   // device: tries a lane runs on its own before asking its wavefront for help

struct fsd_try_t {
    vec2 x;
    float f;
    uint32_t accept;
};
WT_HD uint32_t fsd_max_tries(const fsd_aperture_t& ap) { return ap.dead ? 0u : ap.n_edges * 1024u; }
WT_HD uint32_t fsd_tries_base(const sampler_t& s) { return (s.draws + 3u) & ~3u; }
// one try; `s` positioned at the try's first draw
// fsd_sampling_density and fsd_ASF of the same point in ONE pass over the segments (same per-segment arithmetic; a try of the
// rejection loop needs both, and the per-segment amplitudes alpha1 / alpha2 are what they cost)
WT_HD void fsd_density_and_ASF(const fsd_aperture_t& ap, const fsd_edges_ref_t& ed, vec2 xi, float& g, float& f) {
    cplx amp{0.f, 0.f};
    float d = 0.f;
    for (uint32_t i = 0; i < ap.n_edges; ++i) {
        const fsd_edge_t e = ed.get(i);
#if WT_SS_FSD
        float a12[2];
        ss_fraunhofer_segment(&e.e.x, e.ab, e.iab, &xi.x, a12);
        const float a1 = a12[0], a2 = a12[1];
#else
        const vec2 z = fsd_zeta(e, xi);
        const float a1 = e.ab * fsd_alpha1(z.x, z.y);
        const float a2 = e.iab * fsd_alpha2(z.x, z.y);
#endif
        const float ee2 = length2(e.e);
        amp = amp + cpolar(ee2, -dot(e.v, xi)) * cplx{a1, a2};
        d += sqr(ee2) * (a1 * a1 + a2 * a2);
    }
    const float ce = fsd_chi_e(xi), c0 = fsd_chi_0(xi);
    g = d * ce + ap.P0 * kInvTwoPi / sqr(kFsdP0Sigma) * c0;
    f = cnorm(amp) * ce + ap.psi02 * c0;
}
WT_HD fsd_try_t fsd_try(const scene_t& sc, const fsd_aperture_t& ap, const fsd_edges_ref_t& ed, sampler_t s) {
    fsd_try_t r;
    r.x = fsd_sampleN(sc, ap, ed, s);
    float g;
    fsd_density_and_ASF(ap, ed, r.x, g, r.f);
    r.accept = ap.n_edges > 1 ? (sampler_r(s) * g < r.f * (1.f / float(ap.n_edges)) ? 1u : 0u) : 1u;
    return r;
}
// tries [t0,t1) in order; returns the index of the first accepted one (result in `out`) or 0xFFFFFFFF
WT_HD uint32_t fsd_run_tries(const scene_t& sc, const fsd_aperture_t& ap, const fsd_edges_ref_t& ed, const sampler_t& stream_sampler, uint32_t base,
                             uint32_t t0, uint32_t t1, fsd_try_t& out) {
    for (uint32_t t = t0; t < t1; ++t) {
        const fsd_try_t r = fsd_try(sc, ap, ed, sampler_at(stream_sampler, base + t * kFsdDrawsPerTry));
        if (r.accept) {
            out = r;
            return t;
        }
    }
    return 0xFFFFFFFFu;
}
WT_HD uint32_t fsd_draws_after(uint32_t base, uint32_t last_try) { return base + (last_try + 1u) * kFsdDrawsPerTry; }
WT_HD fsd_sample_t fsd_finalize(const fsd_aperture_t& ap, bool accepted, vec2 xi, float f) {
    const float pdf = accepted ? f * ap.recp_I : 0.f;
    const float scale = ap.k * 1.f;
    if (pdf > 0.f) {
        const vec2 zeta = xi / scale;
        const vec2 wol{zeta.x / sqrtf(1.f + sqr(zeta.x)), zeta.y / sqrtf(1.f + sqr(zeta.y))};
        const float wo2 = length2(wol);
        if (wo2 < kFsdWo2Cutoff) return {vec3{wol.x, wol.y, sqrtf(1.f - wo2)}, pdf, 1.f};
    }
    return {vec3{0, 0, 1}, 0.f, 0.f};
}
WT_HD fsd_sample_t fsd_sample(const scene_t& sc, const fsd_aperture_t& ap, const fsd_edges_ref_t& ed, sampler_t& sampler) {
    const uint32_t max_tries = fsd_max_tries(ap), base = fsd_tries_base(sampler);
    fsd_try_t r{{0.f, 0.f}, 0.f, 0u};
    const uint32_t t = fsd_run_tries(sc, ap, ed, sampler, base, 0, max_tries, r);
    const bool accepted = t != 0xFFFFFFFFu;
#ifdef WT_PROFILE_CONE_TRI
    {
        const uint32_t tries = accepted ? t + 1 : max_tries;
        int eb = 0; while ((1u << (eb + 1)) <= ap.n_edges && eb < 7) ++eb;
        int tb = 0; while ((1u << (tb + 1)) <= tries && tb < 23) ++tb;
        g_fsd_hist[eb][accepted ? tb : 23]++;
    }
#endif
    sampler_seek(sampler, fsd_draws_after(base, accepted ? t : max_tries - 1u));
    return fsd_finalize(ap, accepted, r.x, r.f);
}
// free_space_diffraction_t::pdf / f (free_space_diffraction.hpp:112-133)
WT_HD float fsd_pdf(const fsd_aperture_t& ap, const fsd_edges_ref_t& ed, vec3 wolocal) {
    const float wo2 = sqr(wolocal.x) + sqr(wolocal.y);
    if (wolocal.z <= 0.f || wo2 >= kFsdWo2Cutoff) return 0.f;
    const float scale = ap.k * 1.f;
    const vec2 zeta{wolocal.x / sqrtf(1.f - sqr(wolocal.x)), wolocal.y / sqrtf(1.f - sqr(wolocal.y))};
    const vec2 xi = scale * zeta;
    const float pdf = fsd_ASF(ap, ed, xi) * ap.recp_I;
    return (0.f <= pdf && pdf < 1e+2f) ? pdf : 0.f;
}

}   // namespace wt
