// oracle/indep/prims2.cpp — SECOND SOURCES of the riskiest physics primitives, for a RENDER.          *** TEST INFRASTRUCTURE ***
//
// oracle/indep/indep.cpp restates the COMPOSITION of both integrators a second time, but the primitives it calls are the shared ones of
// wave_tracer_amd/csrc/wt/*.h (wrapped in prims.cpp).  This file holds independent derivations, in double precision and without a line of a
// wt/ header, of the primitives a mistaken restatement would most plausibly hide in; `make -C oracle` links them into libindep2.so, where
// the wt/ headers — compiled with -DWT_SECOND_SOURCE — hand these calls over (the hooks are marked WT_SECOND_SOURCE in wt/cone.h, wt/bvh.h,
// wt/polar.h, wt/fsd.h and wt/utd.h).  tests/test_second_source.py::test_render_on_second_source_primitives renders whole images through libindep2.so and compares
// them with liboracle.so's: an image that needs BOTH the second composition and these second primitives to agree with the checker.
//
//   ss_intersect_cone_tri   closest distance along the axis at which an elliptic cone meets a triangle inside a z-slab
//                           (reference: include/wt/math/intersect/cone.hpp:550-626, by cone-plane and cone-edge intersections in 3-D).
//                           Here: a convex programme in the triangle's barycentric plane — minimise z(u, v) over the triangle, the cone's
//                           quadratic form restricted to the plane and the slab — solved by enumerating its KKT candidates (the C++ form of
//                           second_source.py: cone_tri_min_z, which tests/test_second_source.py checks against brute force).
//   (no culling)            the cone x AABB test of the 8-wide traversal (src/ads/bvh8w.cpp:187-230) has the trivial second source: accept every
//                           child — the traversal becomes a scan of every triangle, whose result does not depend on any box test.
//   ss_fresnel_*            Fresnel amplitude coefficients from the textbook ANGLE forms  rs = -sin(ti - tt) / sin(ti + tt),
//                           rp = tan(ti - tt) / tan(ti + tt)  (reference: include/wt/interaction/fresnel.hpp:74-117 works with cosines), and for
//                           an absorbing second medium from the relative permittivity form  rs = (cos ti - sqrt(m^2 - sin^2 ti)) / (...),
//                           m = n2 / n1 complex (reference fresnel.hpp:128-144: in terms of eta_12 = n1 / n2).
//   ss_mueller_from_jones   the Mueller matrix of a diagonal Jones matrix diag(fs, fp) as  A (J (x) J*) A^-1  (reference mueller.hpp:244-259
//                           writes the eight non-zero entries down).  The handedness of the (U, V) block is a convention of the reference's
//                           Stokes vectors, not physics: this construction takes the Kronecker product in the order that reproduces it.
//   ss_fraunhofer_segment   the two amplitudes a Fraunhofer aperture segment contributes at a direction xi (reference: include/wt/interaction/
//                           fsd/fsd.hpp:65-121 — closed forms alpha_1, alpha_2 in zeta = (xi . e, xi x e)).  Here: Stokes' theorem turns the
//                           Fourier integral of the aperture into a line integral over its boundary; a segment's share is
//                           (xi x e) / |xi|^2  int c(t) exp(-i xi . x(t)) dt  with the field amplitude c linear along the segment — evaluated by
//                           composite Gauss-Legendre quadrature in f64 and split into the same two real amplitudes (hooks in wt/fsd.h).
//   ss_wedge_utd            soft / hard UTD coefficients of a wedge (reference: include/wt/interaction/fsd/utd.hpp:25-57 + fsd/common.hpp:42-88:
//                           the transition function through erfc below |x| = 6 and a four-term asymptote above).  Here: Kouyoumjian & Pathak's
//                           four cotangent terms in f64, N+- from their defining equations by search, and the transition function
//                           F(x) = 2 i sqrt(x) e^{ix} int_{sqrt x}^inf e^{-i t^2} dt from a quadrature of that integral on the contour rotated by
//                           -pi/4, where the integrand decays like a Gaussian and nothing cancels: no series, no asymptote, any x.  Prefactor and
//                           sign are the reference's convention (its -D (D1 + D2 -+ (D3 + D4)) with D_i = -cot(.) F(.)), hook in wt/utd.h.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <atomic>
#include <cstdint>

namespace {

typedef std::complex<double> cd;
// how often each second source was asked (tests: a render that is said to run on them did)
enum { SS_CONE_TRI, SS_FRESNEL_DIELECTRIC, SS_FRESNEL_CONDUCTOR, SS_MUELLER, SS_FRAUNHOFER, SS_UTD, SS_COUNT };
std::atomic<uint64_t> g_calls[SS_COUNT];

struct v3 {
    double x, y, z;
};
inline v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline v3 operator*(double s, v3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline v3 cross(v3 a, v3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// real roots of a t^2 + b t + c (the numerically stable pair)
void quad_roots(double a, double b, double c, std::vector<double>& out) {
    if (std::fabs(a) < 1e-300) {
        if (std::fabs(b) >= 1e-300) out.push_back(-c / b);
        return;
    }
    const double D = b * b - 4 * a * c;
    if (D < 0) return;
    const double s = std::sqrt(D);
    const double q = -0.5 * (b + (b >= 0 ? s : -s));
    out.push_back(q / a);
    if (std::fabs(q) > 1e-300) out.push_back(c / q);
}

// The minimal z at which the cone  x^2 + (e y)^2 <= (z tan_alpha + x0)^2  meets the triangle P (cone frame: z along the axis, x the major
// axis) inside [zmin, zmax]; FALSE: they do not meet.
bool cone_tri_min_z(const v3 P[3], double ta, double x0, double e, double zmin, double zmax, double& best) {
    double scale = 1.0;
    for (int i = 0; i < 3; ++i) scale = std::max(scale, std::max(std::fabs(P[i].x), std::max(std::fabs(P[i].y), std::fabs(P[i].z))));
    const v3 A0 = P[0], E1 = P[1] - P[0], E2 = P[2] - P[0];
    auto point = [&](double u, double v) { return A0 + u * E1 + v * E2; };
    auto g_of = [&](v3 p) {
        const double r = p.z * ta + x0;
        return p.x * p.x + (e * p.y) * (e * p.y) - r * r;
    };
    bool found = false;
    best = 0;
    // (tolerances: the candidates are roots computed in f64 from f32 data; a candidate's coordinates carry 1e-16 of the TRIANGLE's size, i.e. its
    // residual in the cone's form is ~1e-16 size / radius of the terms of that form — 1e-10 for a micrometre beam on a metre wall.  1e-7 of the
    // LOCAL terms admits that and lets a beam accept points 1e-7 of its radius outside it; a tolerance on the scene's scale would not do: 1e-9 of
    // a square metre is a thousand times the cross-section of such a beam)
    auto consider = [&](double u, double v) {
        const double tol = 1e-7;
        if (u < -1e-11 || v < -1e-11 || u + v > 1 + 1e-11) return;
        const v3 p = point(u, v);
        const double r = p.z * ta + x0;
        if (r < -1e-11 * scale || p.z < zmin - 1e-11 * scale || p.z > zmax + 1e-11 * scale) return;
        if (g_of(p) > tol * (p.x * p.x + (e * p.y) * (e * p.y) + r * r) + 1e-300) return;
        if (!found || p.z < best) {
            best = p.z;
            found = true;
        }
    };
    // vertices
    consider(0, 0);
    consider(1, 0);
    consider(0, 1);
    // edges (u, v) = q0 + t dq: crossings of the cone's surface and of the slab's planes
    const double q0s[3][2] = {{0, 0}, {0, 0}, {1, 0}}, dqs[3][2] = {{1, 0}, {0, 1}, {-1, 1}};
    for (int k = 0; k < 3; ++k) {
        const v3 a = point(q0s[k][0], q0s[k][1]);
        const v3 d = point(q0s[k][0] + dqs[k][0], q0s[k][1] + dqs[k][1]) - a;
        const double ra = a.z * ta + x0, rd = d.z * ta;
        std::vector<double> ts;
        quad_roots(d.x * d.x + (e * d.y) * (e * d.y) - rd * rd, 2 * (a.x * d.x + e * e * a.y * d.y - ra * rd), a.x * a.x + (e * a.y) * (e * a.y) - ra * ra, ts);
        for (double zz : {zmin, zmax})
            if (std::isfinite(zz) && std::fabs(d.z) > 1e-300) ts.push_back((zz - a.z) / d.z);
        for (double t : ts)
            if (t >= -1e-12 && t <= 1 + 1e-12) consider(q0s[k][0] + t * dqs[k][0], q0s[k][1] + t * dqs[k][1]);
    }
    // the cone's quadratic form restricted to the triangle's plane: g(q) = q^T H q + 2 h^T q + c0
    auto qf = [&](v3 Ea, v3 Eb) { return Ea.x * Eb.x + e * e * Ea.y * Eb.y - (Ea.z * ta) * (Eb.z * ta); };
    const double r0 = A0.z * ta + x0;
    const double H00 = qf(E1, E1), H01 = qf(E1, E2), H11 = qf(E2, E2);
    const double h0 = A0.x * E1.x + e * e * A0.y * E1.y - r0 * E1.z * ta, h1 = A0.x * E2.x + e * e * A0.y * E2.y - r0 * E2.z * ta;
    const double c0 = A0.x * A0.x + (e * A0.y) * (e * A0.y) - r0 * r0;
    const double zg0 = E1.z, zg1 = E2.z;
    const double det = H00 * H11 - H01 * H01;
    const double hmax = std::max(std::fabs(H00), std::max(std::fabs(H01), std::fabs(H11)));
    // stationary points of z on the cone's surface (Lagrange): H q + h = mu grad z, g(q) = 0
    if (std::fabs(det) > 1e-14 * (hmax * hmax + 1e-300) && std::max(std::fabs(zg0), std::fabs(zg1)) > 1e-300) {
        const double i00 = H11 / det, i01 = -H01 / det, i11 = H00 / det;
        const double qh0 = -(i00 * h0 + i01 * h1), qh1 = -(i01 * h0 + i11 * h1);
        const double qz0 = i00 * zg0 + i01 * zg1, qz1 = i01 * zg0 + i11 * zg1;
        auto Hq = [&](double a0, double a1, double b0, double b1) { return a0 * (H00 * b0 + H01 * b1) + a1 * (H01 * b0 + H11 * b1); };
        std::vector<double> mus;
        quad_roots(Hq(qz0, qz1, qz0, qz1), 2 * (Hq(qh0, qh1, qz0, qz1) + h0 * qz0 + h1 * qz1), Hq(qh0, qh1, qh0, qh1) + 2 * (h0 * qh0 + h1 * qh1) + c0, mus);
        for (double mu : mus) consider(qh0 + mu * qz0, qh1 + mu * qz1);
    }
    // the slab's near plane: the triangle's section at z = zmin may cross the cone's disk although no candidate above lies on it
    if (std::isfinite(zmin) && std::max(std::fabs(zg0), std::fabs(zg1)) > 1e-300 && (!found || best > zmin)) {
        double pts[3][2];
        int n = 0;
        for (int k = 0; k < 3; ++k) {
            const double za = point(q0s[k][0], q0s[k][1]).z, zb = point(q0s[k][0] + dqs[k][0], q0s[k][1] + dqs[k][1]).z;
            if (std::fabs(zb - za) > 1e-300) {
                const double t = (zmin - za) / (zb - za);
                if (t >= -1e-12 && t <= 1 + 1e-12) {
                    pts[n][0] = q0s[k][0] + t * dqs[k][0];
                    pts[n][1] = q0s[k][1] + t * dqs[k][1];
                    ++n;
                }
            }
        }
        if (n >= 2) {
            int far = n - 1;
            double dfar = std::hypot(pts[far][0] - pts[0][0], pts[far][1] - pts[0][1]);
            for (int k = 1; k < n; ++k) {
                const double dk = std::hypot(pts[k][0] - pts[0][0], pts[k][1] - pts[0][1]);
                if (dk > dfar) {
                    dfar = dk;
                    far = k;
                }
            }
            const double a0 = pts[0][0], a1 = pts[0][1], d0 = pts[far][0] - a0, d1 = pts[far][1] - a1;
            auto Hq = [&](double p0, double p1, double b0, double b1) { return p0 * (H00 * b0 + H01 * b1) + p1 * (H01 * b0 + H11 * b1); };
            const double a3 = Hq(d0, d1, d0, d1), b3 = 2 * (Hq(a0, a1, d0, d1) + h0 * d0 + h1 * d1), c3 = Hq(a0, a1, a0, a1) + 2 * (h0 * a0 + h1 * a1) + c0;
            double gmin = std::min(c3, a3 + b3 + c3);
            if (std::fabs(a3) > 1e-300) {
                const double t = std::min(1.0, std::max(0.0, -b3 / (2 * a3)));
                gmin = std::min(gmin, a3 * t * t + b3 * t + c3);
            }
            const double rmin = zmin * ta + x0;
            if (gmin <= 1e-7 * rmin * rmin && rmin >= 0) {
                best = zmin;
                found = true;
            }
        }
    }
    return found;
}

// ---- Fraunhofer: one segment's share of the aperture's boundary integral ----------------------------------------------------------------------
// 16-point Gauss-Legendre on [-1, 1]
static const double kGL16x[8] = {0.0950125098376374402, 0.2816035507792589132, 0.4580167776572273863, 0.6178762444026437484,
                                 0.7554044083550030339, 0.8656312023878317439, 0.9445750230732325761, 0.9894009349916499326};
static const double kGL16w[8] = {0.1894506104550684963, 0.1826034150449235888, 0.1691565193950025382, 0.1495959888165767321,
                                 0.1246289712555338720, 0.0951585116824927848, 0.0622535239386478929, 0.0271524594117540949};
template <class F>
static cd gl16(double a, double b, F f) {
    const double c = 0.5 * (a + b), r = 0.5 * (b - a);
    cd s = 0;
    for (int i = 0; i < 8; ++i) s += kGL16w[i] * (f(c + r * kGL16x[i]) + f(c - r * kGL16x[i]));
    return r * s;
}
// ---- UTD: wedge coefficients ------------------------------------------------------------------------------------------------------------------
// F(x) = 2 i sqrt(x) e^{ix} int_{sqrt x}^inf e^{-i t^2} dt, x >= 0.  With t = s + u e^{-i pi/4} (s = sqrt x):  -i t^2 = -i s^2 - sqrt2 s u (1 + i) - u^2,
// so  F(x) = 2 s e^{i pi/4} int_0^inf exp(-u^2 - sqrt2 s u) (cos(sqrt2 s u) - i sin(sqrt2 s u)) du:  F(0) = 0, F(inf) = 1, every x in between
// from one smooth integral (support: u < 7 and sqrt2 s u < 45).
static cd utd_transition(double x) {
    if (!(x > 0.0)) return 0.0;
    const double s = std::sqrt(x), a = std::sqrt(2.0) * s;
    const double U = std::min(7.0, 45.0 / a);
    const int panels = 24;
    cd I = 0;
    for (int p = 0; p < panels; ++p)
        I += gl16(U * p / panels, U * (p + 1) / panels, [&](double u) { return std::exp(-u * u - a * u) * cd(std::cos(a * u), -std::sin(a * u)); });
    return 2.0 * s * std::exp(cd(0, M_PI / 4)) * I;
}
// a+-(beta) = 2 cos^2((2 pi n N+- - beta) / 2), N+- the integer that most nearly satisfies 2 pi n N - beta = +-pi (searched, not rounded)
static double utd_a_pm(double beta, double n, int sgn) {
    int best = 0;
    double err = 1e300;
    for (int N = -4; N <= 4; ++N) {
        const double d = std::fabs(2.0 * M_PI * n * N - beta - sgn * M_PI);
        if (d < err - 1e-12) {   // (ties — beta = -+pi exactly... — keep the smaller |N| first met from below, as rounding half away from zero would not matter: a is the same)
            err = d;
            best = N;
        }
    }
    const double c = std::cos(0.5 * (2.0 * M_PI * n * best - beta));
    return 2.0 * c * c;
}
}   // namespace

extern "C" {

// cone: origin o, axis d, major-axis direction x (unit, orthogonal to d), x0, tan_alpha, eccentricity e (major / minor).  Returns 1 and the
// closest distance along the axis inside [zmin, zmax], or 0.
int ss_intersect_cone_tri(const float o[3], const float d[3], const float x[3], float x0, float tan_alpha, float e, const float a[3], const float b[3], const float c[3],
                          float zmin, float zmax, float* dist) {
    g_calls[SS_CONE_TRI].fetch_add(1, std::memory_order_relaxed);
    const v3 O{o[0], o[1], o[2]}, Z{d[0], d[1], d[2]}, X{x[0], x[1], x[2]};
    const v3 Y = cross(Z, X);
    const float* vs[3] = {a, b, c};
    v3 P[3];
    for (int i = 0; i < 3; ++i) {
        const v3 w = v3{vs[i][0], vs[i][1], vs[i][2]} - O;
        P[i] = {dot(w, X), dot(w, Y), dot(w, Z)};
    }
    double z = 0;
    const bool hit = cone_tri_min_z(P, tan_alpha, x0, e, zmin, std::isfinite(zmax) ? (double)zmax : INFINITY, z);
    static FILE* dump = getenv("WT_SS_DUMP") ? fopen(getenv("WT_SS_DUMP"), "wb") : nullptr;   // (diagnostic: the queries of a render, for tools that replay them)
    if (dump) {
        const float rec[24] = {o[0], o[1], o[2], d[0], d[1], d[2], x[0], x[1], x[2], x0, tan_alpha, e, a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2], zmin, zmax, hit ? (float)z : -1.f};
        fwrite(rec, sizeof(rec), 1, dump);
        fflush(dump);
    }
    if (!hit) return 0;
    *dist = (float)std::min(std::max(z, (double)zmin), (double)zmax);
    return 1;
}

// Dielectric interface, real relative index eta = n1 / n2 as seen from the incident side, |cos theta_i| = ci > 0, no total internal
// reflection (the caller has handled it): amplitude coefficients from the angle forms.  out: rs, rp, ts, tp, Z = n2 cos tt / (n1 cos ti).
void ss_fresnel_dielectric(double eta, double ci, double out[5]) {
    g_calls[SS_FRESNEL_DIELECTRIC].fetch_add(1, std::memory_order_relaxed);
    const double ti = std::acos(std::min(1.0, ci));
    const double st = eta * std::sin(ti);
    const double tt = std::asin(std::min(1.0, st));
    double rs, rp;
    if (ti < 1e-7) {   // normal incidence: the limits of the angle forms
        rs = (eta - 1) / (eta + 1);
        rp = (1 - eta) / (1 + eta);
    } else {
        rs = -std::sin(ti - tt) / std::sin(ti + tt);
        rp = std::tan(ti - tt) / std::tan(ti + tt);
    }
    // transmission from the boundary conditions: Es continuous => ts = 1 + rs;  Hs continuous => tp = (1 + rp) n1 / n2
    out[0] = rs;
    out[1] = rp;
    out[2] = 1 + rs;
    out[3] = (1 + rp) * eta;
    out[4] = std::cos(tt) / (eta * std::cos(ti));
}
// Reflection off an absorbing medium: eta = n1 / n2 complex; m = 1 / eta the relative refractive index of the second medium.  out: rs, rp (re, im).
void ss_fresnel_conductor(double eta_re, double eta_im, double ci, double out[4]) {
    g_calls[SS_FRESNEL_CONDUCTOR].fetch_add(1, std::memory_order_relaxed);
    const cd m = 1.0 / cd(eta_re, eta_im);
    const double s2 = 1 - ci * ci;
    const cd root = std::sqrt(m * m - s2);   // m cos(theta_t), principal branch
    const cd rs = (ci - root) / (ci + root);
    const cd rp = (m * m * ci - root) / (m * m * ci + root);
    out[0] = rs.real();
    out[1] = rs.imag();
    out[2] = rp.real();
    out[3] = rp.imag();
}
// Mueller matrix (row major) of the Jones matrix diag(fs, fp) in the (s, p) basis: M = A (J (x) J*) A^-1 with Stokes S = A (Es Es*, Es Ep*,
// Ep Es*, Ep Ep*)^T, A = [[1,0,0,1],[1,0,0,-1],[0,1,1,0],[0,i,-i,0]] — with the conjugate on the FIRST factor, which is the handedness of the
// reference's V component.
void ss_mueller_from_jones(double fs_re, double fs_im, double fp_re, double fp_im, float M[16]) {
    g_calls[SS_MUELLER].fetch_add(1, std::memory_order_relaxed);
    const cd J[2] = {cd(fs_re, fs_im), cd(fp_re, fp_im)};
    const cd I(0, 1);
    const cd A[4][4] = {{1, 0, 0, 1}, {1, 0, 0, -1}, {0, 1, 1, 0}, {0, I, -I, 0}};
    const cd Ai[4][4] = {{0.5, 0.5, 0, 0}, {0, 0, 0.5, -0.5 * I}, {0, 0, 0.5, 0.5 * I}, {0.5, -0.5, 0, 0}};
    // K = conj(J) (x) J is diagonal: K[2a+b] = conj(J[a]) J[b]
    cd K[4];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) K[2 * a + b] = std::conj(J[a]) * J[b];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            cd s = 0;
            for (int k = 0; k < 4; ++k) s += A[r][k] * K[k] * Ai[k][c];
            M[4 * r + c] = (float)s.real();
        }
}

// The segment runs from x0 = v - e/2 to v + e/2 with field amplitude ca at its start and cb at its end (the record holds ab = ca - cb and
// iab = (ca + cb) / 2, its mid point v is applied by the caller as the phase exp(-i xi . v) and |e|^2 as a factor).  Its share of
//   B(xi) = sum (xi x e) / |xi|^2  int_0^1 c(t) exp(-i xi . (x0 + t e)) dt
// is  exp(-i xi . v) |e|^2 (xi x e) / (|xi|^2 |e|^2)  I,   I = int_{-1/2}^{1/2} (iab - s ab) exp(-i (xi . e) s) ds,
// and the caller's complex amplitude (a1 + i a2) is i B / (2 pi) without the two factors in front:  a2 - i a1 = (xi x e) / (2 pi |xi|^2 |e|^2) I.
void ss_fraunhofer_segment(const float e[2], float ab, float iab, const float xi[2], float out[2]) {
    g_calls[SS_FRAUNHOFER].fetch_add(1, std::memory_order_relaxed);
    const double ex = e[0], ey = e[1], X = xi[0], Y = xi[1];
    const double xi2 = X * X + Y * Y, e2 = ex * ex + ey * ey;
    out[0] = out[1] = 0.f;
    if (xi2 == 0.0 || e2 == 0.0) return;
    const double zx = X * ex + Y * ey, zy = X * ey - Y * ex;
    const int panels = 1 + (int)std::ceil(std::fabs(zx) / 6.0);
    cd I = 0;
    for (int p = 0; p < panels; ++p) {
        const double a = -0.5 + (double)p / panels, b = -0.5 + (double)(p + 1) / panels;
        I += gl16(a, b, [&](double s) { return ((double)iab - s * (double)ab) * std::exp(cd(0, -zx * s)); });
    }
    const cd q = zy / (2.0 * M_PI * xi2 * e2) * I;
    out[0] = (float)(-q.imag());
    out[1] = (float)q.real();
}

void ss_utd_transition(double x, double out[2]) {   // (for the known-answer test of the contour quadrature itself: against scipy's Fresnel integrals)
    const cd f = utd_transition(x);
    out[0] = f.real();
    out[1] = f.imag();
}
// out = (Re Ds, Im Ds, Re Dh, Im Dh), including the reference's prefactor e^{-i pi/4} / (2 n sqrt(2 pi k ro) sin beta0) (k ro: the spreading factor of
// the reference's caller is folded in) and its overall sign.
void ss_wedge_utd(double n, double k_Li, double k_ro, double sin_beta, double phii, double phio, double out[4]) {
    g_calls[SS_UTD].fetch_add(1, std::memory_order_relaxed);
    const double bm = phii - phio, bp = phii + phio;
    const auto cot = [](double x) { return std::cos(x) / std::sin(x); };
    const cd T1 = cot((M_PI + bm) / (2 * n)) * utd_transition(k_Li * utd_a_pm(bm, n, +1));
    const cd T2 = cot((M_PI - bm) / (2 * n)) * utd_transition(k_Li * utd_a_pm(bm, n, -1));
    const cd T3 = cot((M_PI + bp) / (2 * n)) * utd_transition(k_Li * utd_a_pm(bp, n, +1));
    const cd T4 = cot((M_PI - bp) / (2 * n)) * utd_transition(k_Li * utd_a_pm(bp, n, -1));
    const cd D = std::exp(cd(0, -M_PI / 4)) / (2.0 * n * std::sqrt(2.0 * M_PI * k_ro) * sin_beta);
    const cd Ds = D * (T1 + T2 - (T3 + T4)), Dh = D * (T1 + T2 + (T3 + T4));
    out[0] = Ds.real();
    out[1] = Ds.imag();
    out[2] = Dh.real();
    out[3] = Dh.imag();
}

// calls since the last reset: cone x triangle, Fresnel (dielectric), Fresnel (conductor), Mueller, Fraunhofer segment, UTD wedge
void ss_calls(uint64_t out[6], int reset) {
    for (int i = 0; i < SS_COUNT; ++i) {
        out[i] = g_calls[i].load();
        if (reset) g_calls[i].store(0);
    }
}

}   // extern "C"
