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
// (h = distance of the edge's line from the origin, phi measured from the line's normal) with an 8-point
// Gauss-Legendre rule on up to two sub-intervals per edge.  This is deterministic, branch-light and agrees with
// the closed forms for half-planes/wedges to ~1e-6 (tests/test_kat.py::test_gauss_triangle).
#pragma once
#include "core.h"

namespace wt {

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
    // angles of p and q measured from the normal direction nn, in the basis (nn, tt) with tt = rot90(nn)
    const vec2 tt{-nn.y, nn.x};
    const float phip = atan2f(dot(p, tt), dot(p, nn));
    const float phiq = atan2f(dot(q, tt), dot(q, nn));
    if (h == 0.f) return 0.f;   // the line passes through the origin: rho = 0 => integrand 0
    // 8-point Gauss-Legendre on [phip,phiq], split in two halves for accuracy on long edges
    const float xs[4] = {0.1834346424956498f, 0.5255324099163290f, 0.7966664774136267f, 0.9602898564975363f};
    const float ws[4] = {0.3626837833783620f, 0.3137066458778873f, 0.2223810344533745f, 0.1012285362903763f};
    const float h2 = 0.5f * h * h;
    float total = 0.f;
    for (int half = 0; half < 2; ++half) {
        const float a = half == 0 ? phip : 0.5f * (phip + phiq);
        const float b = half == 0 ? 0.5f * (phip + phiq) : phiq;
        const float c = 0.5f * (a + b), r = 0.5f * (b - a);
        float s = 0.f;
        for (int i = 0; i < 4; ++i) {
            const float c1 = cosf(c + r * xs[i]), c2 = cosf(c - r * xs[i]);
            s += ws[i] * ((1.f - expf(-h2 / (c1 * c1))) + (1.f - expf(-h2 / (c2 * c2))));
        }
        total += s * r;
    }
    return total * kInvTwoPi;
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
