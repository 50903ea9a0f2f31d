// wave_tracer_amd — integral of the beam's Gaussian cross-section over a projected triangle.
//
// Reference: src/math/gaussian2d.cpp:101-192 (gaussian2d_t::integrate_triangle), used by
//            plt_bdpt_detail.hpp:391-416 to compute the power blocked by front/back-facing triangles.
//
// The reference evaluates  I = (1/2pi) * Int_T exp(-|x|^2/2) dA  (T in canonical, sigma-normalised coordinates)
// with a 0.002-step brute-force quadrature (up to 9e6 exp() per triangle) or a 4-term erf fit, and documents
// "accuracy usually within 1-3% rel. err." (gaussian2d.hpp).  On a GPU neither is acceptable, so the same
// quantity is evaluated *exactly up to quadrature error* through the polar form
//      I = (1/2pi) * Sum_edges  Int_{phi_p}^{phi_q} ( 1 - exp(-h^2 / (2 cos^2 phi)) ) dphi ,
// (h = distance of the edge's line from the origin, phi measured from the line's normal) with 8-point Gauss-Legendre rules: one in phi
// on |phi| <= pi/4, and — round 5 — log-graded pieces in u = ln(h tan phi) beyond, where the integrand has a boundary layer of width ~h
// (gauss_edge_side).  Deterministic; agrees with a double-precision quadrature of the same integral to 4e-6 over random triangles of every
// scale, edges through the beam axis included (tests/test_kat.py::test_gaussian_triangle_integral; round 4's rule: 1.2e-3 on such edges).
#pragma once
#include "core.h"

namespace wt {

// The two integrands of an edge term: in the polar angle phi (KIND 0, par = h^2 / 2) and in u = ln(h tan phi) (KIND 1, par = h)
template <int KIND>
WT_HD float gauss_edge_integrand(float par, float x) {
    if (KIND == 0) {
        const float c = cosf(x);
        return 1.f - expf(-par / (c * c));
    }
    const float s = expf(x);
    return (1.f - expf(-0.5f * (par * par + s * s))) * (par * s / (par * par + s * s));
}
// 8-point Gauss-Legendre rule on [a, b]
template <int KIND>
WT_HD float gauss_gl8(float a, float b, float par) {
    const float c = 0.5f * (a + b), r = 0.5f * (b - a);
    const float s = 0.3626837833783620f * (gauss_edge_integrand<KIND>(par, c + r * 0.1834346424956498f) + gauss_edge_integrand<KIND>(par, c - r * 0.1834346424956498f)) +
                    0.3137066458778873f * (gauss_edge_integrand<KIND>(par, c + r * 0.5255324099163290f) + gauss_edge_integrand<KIND>(par, c - r * 0.5255324099163290f)) +
                    0.2223810344533745f * (gauss_edge_integrand<KIND>(par, c + r * 0.7966664774136267f) + gauss_edge_integrand<KIND>(par, c - r * 0.7966664774136267f)) +
                    0.1012285362903763f * (gauss_edge_integrand<KIND>(par, c + r * 0.9602898564975363f) + gauss_edge_integrand<KIND>(par, c - r * 0.9602898564975363f));
    return s * r;
}
// Int ( 1 - exp(-h^2 / (2 cos^2 phi)) ) dphi over the part of an edge beyond phi = pi/4, in the variable u = ln s, s = h tan phi — which is simply
// the coordinate ALONG the edge's line measured from the foot of the perpendicular (no tangent to evaluate): [s_lo, s_hi], h <= s_lo < s_hi,
// phi_hi = atan(s_hi / h).  Towards pi/2 the integrand climbs from 1 - exp(-h^2) to 1 inside a layer of width ~h, which no fixed rule in phi
// resolves for an edge whose line passes close to the beam axis (h << 1: round 4's two 8-point halves were off by up to 1.2e-3 there); in u the
// integrand  (1 - exp(-(h^2 + s^2) / 2)) h s / (h^2 + s^2)  is smooth, and pieces of 1.25 in u get it to float rounding.  Beyond s = 8 the
// exponential is gone: closed form.
WT_HD float gauss_edge_side(float h, float s_lo, float s_hi, float phi_hi) {
    float tail = 0.f;
    if (s_hi > 8.f) {
        tail = phi_hi - atan2f(fmaxf_(s_lo, 8.f), h);
        s_hi = 8.f;
    }
    if (s_hi <= s_lo) return tail;
    const float u1 = logf(s_lo), u2 = logf(s_hi);
    const int n = (int)fmaxf_(1.f, ceilf((u2 - u1) / 1.25f));
    const float du = (u2 - u1) / (float)n;
    float sum = 0.f;
    for (int i = 0; i < n; ++i) sum += gauss_gl8<1>(u1 + du * (float)i, i + 1 == n ? u2 : u1 + du * (float)(i + 1), h);
    return sum + tail;
}
WT_HD float gauss_edge_term(vec2 p, vec2 q) {
    const vec2 d = q - p;
    const float len = length(d);
    if (len == 0.f) return 0.f;
    const vec2 t = d / len;
    const vec2 n{t.y, -t.x};          // normal; h = dot(p,n) signed distance of the line
    float h = dot(p, n);
    vec2 nn = n;
    if (h < 0.f) {
        h = -h;
        nn = -n;
    }
    // angles of p and q measured from the normal direction nn, in the basis (nn, tt) with tt = rot90(nn); y = h tan phi: the coordinate along the line
    const vec2 tt{-nn.y, nn.x};
    const float yp = dot(p, tt), yq = dot(q, tt);
    const float phip = atan2f(yp, dot(p, nn));
    const float phiq = atan2f(yq, dot(q, nn));
    if (h == 0.f) return 0.f;   // the line passes through the origin: rho = 0 => integrand 0
    // [phip, phiq] in three parts: |phi| <= pi/4, i.e. |y| <= h (the integrand is smooth in phi: one 8-point rule), and the two sides beyond
    const float kQuarterPi = 0.78539816339744831f;
    const bool fwd = phiq >= phip;
    const float lo = fwd ? phip : phiq, hi = fwd ? phiq : phip, ylo = fwd ? yp : yq, yhi = fwd ? yq : yp;
    float total = 0.f;
    const float a = fmaxf_(lo, -kQuarterPi), b = fminf_(hi, kQuarterPi);
    if (b > a) total += gauss_gl8<0>(a, b, 0.5f * h * h);
    if (yhi > h) total += gauss_edge_side(h, fmaxf_(ylo, h), yhi, hi);
    if (ylo < -h) total += gauss_edge_side(h, fmaxf_(-yhi, h), -ylo, -lo);
    return (fwd ? total : -total) * kInvTwoPi;
}

// Integral of the standard normal over triangle (a,b,c) given in canonical coordinates; result in [0,1].
WT_HD float gauss_integrate_triangle_canonical(vec2 a, vec2 b, vec2 c) {
    // orientation: make CCW so that the signed sum is positive
    const float area2 = (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x);
    if (area2 == 0.f) return 0.f;
    // Triangles much smaller than the beam (edges < sigma/10: the triangles of a finely tessellated mesh inside a wide beam, where
    // an interaction region holds 10^3..10^5 of them): the edge-midpoint rule, exact for quadratics — relative error < (h/sigma)^4 / 50
    // = 2e-6, below the fp32 rounding of the sum it enters — instead of ~150 transcendentals.
    {
        const vec2 ab = b - a, bc = c - b, ca = a - c;
        if (fmaxf_(dot(ab, ab), fmaxf_(dot(bc, bc), dot(ca, ca))) < 0.01f) {
            const vec2 m0 = 0.5f * (a + b), m1 = 0.5f * (b + c), m2 = 0.5f * (c + a);
            const float g = expf(-0.5f * dot(m0, m0)) + expf(-0.5f * dot(m1, m1)) + expf(-0.5f * dot(m2, m2));
            return clamp01(fabsf(area2) * (0.5f / 3.f) * kInvTwoPi * g);
        }
    }
    float s = gauss_edge_term(a, b) + gauss_edge_term(b, c) + gauss_edge_term(c, a);
    if (area2 < 0.f) s = -s;
    return clamp01(s);
}

// gaussian_wavefront_t::integrate_triangle: points in metres on the beam cross-section, sigma in metres
WT_HD float wavefront_integrate_triangle(vec2 sigma, vec2 pa, vec2 pb, vec2 pc) {
    if (sigma.x == 0.f || sigma.y == 0.f) return 0.f;
    const vec2 rs{1.f / sigma.x, 1.f / sigma.y};
    return gauss_integrate_triangle_canonical(pa * rs, pb * rs, pc * rs);
}
// gaussian2d pdf with sigma (axis aligned, zero mean); arguments in metres/1m
WT_HD float wavefront_intensity(vec2 sigma, vec2 x) {
    if (sigma.x == 0.f || sigma.y == 0.f) return (x.x == 0.f && x.y == 0.f) ? WT_INF : 0.f;
    const vec2 u = x / sigma;
    return kInvTwoPi / (sigma.x * sigma.y) * expf(-dot(u, u) / 2.f);
}

}   // namespace wt
