// wave_tracer_amd — Fresnel equations, Stokes vectors and Mueller operators.
//
// Reference: include/wt/interaction/fresnel.hpp:19-144,
//            include/wt/interaction/polarimetric/stokes.hpp:146-165 (reorient),
//            include/wt/interaction/polarimetric/mueller.hpp:26-415.
//
// Mueller matrices are stored row-major in the mathematical convention (S' = M S).  The reference stores glm
// column-major matrices; the translations below were derived element by element (DESIGN.md "Mueller
// conventions"): rotation(t1,t2) = [[1,0,0,0],[0,c,s,0],[0,-s,c,0],[0,0,0,1]] with (c,s) = (cos,sin) of
// twice the signed angle from t1 to t2.
#pragma once
#include "core.h"

namespace wt {

// ---- Fresnel (fresnel.hpp) -----------------------------------------------------------------------
WT_HD vec3 reflect_z(vec3 w) { return vec3{-w.x, -w.y, w.z}; }   // reflect(w, n=(0,0,1)) = 2(w.n)n - w
WT_HD vec3 reflect(vec3 w, vec3 n) { return 2.f * dot(w, n) * n - w; }

struct refract_t {
    vec3 t;
    float cost, eta_12;
    bool TIR;
};
WT_HD refract_t refract(float eta_12, vec3 w, vec3 n) {
    const float wn = dot(w, n);
    eta_12 = wn > 0.f ? eta_12 : 1.f / eta_12;
    const float cost2 = 1.f - sqr(eta_12) * (1.f - sqr(wn));
    if (cost2 >= 0.f) {
        const float cost = sqrtf(cost2);
        const vec3 t = eta_12 * (wn * n - w) - cost * (wn >= 0.f ? n : -n);
        return refract_t{normalize(t), cost, eta_12, false};
    }
    return refract_t{vec3{0, 0, 1}, 0.f, eta_12, true};
}

#if defined(WT_SECOND_SOURCE) && !defined(__HIP_DEVICE_COMPILE__)
// (oracle/indep/prims2.cpp: see the note at WT_SS_ACTIVE in wt/cone.h)
extern "C" void ss_fresnel_dielectric(double eta, double ci, double out[5]);
extern "C" void ss_fresnel_conductor(double eta_re, double eta_im, double ci, double out[4]);
extern "C" void ss_mueller_from_jones(double fs_re, double fs_im, double fp_re, double fp_im, float M[16]);
#define WT_SS_POLAR 1
#else
#define WT_SS_POLAR 0
#endif
struct fresnel_t {
    vec3 t;
    cplx eta_12;
    float Z;
    cplx rs, rp, ts, tp;
    float Ts, Tp;
};
// fresnel.hpp:74-117 — the refraction uses Re(eta) and the coefficients are then computed with the *real*
// relative index returned by refract() (eta_12 is overwritten by refr.eta_12 in the reference).
WT_HD fresnel_t fresnel(cplx eta_12, vec3 w, vec3 n) {
    if (eta_12.re == 1.f && eta_12.im == 0.f)
        return fresnel_t{-w, eta_12, 1.f, {0, 0}, {0, 0}, {1, 0}, {1, 0}, 1.f, 1.f};
    const float abs_cosi = fabsf(dot(w, n));
    const refract_t refr = refract(eta_12.re, w, n);
    if (abs_cosi == 0.f || refr.TIR)
        return fresnel_t{vec3{0, 0, 1}, cplx{refr.eta_12, 0.f}, 1.f, {1, 0}, {1, 0}, {0, 0}, {0, 0}, 0.f, 0.f};
    const float cost = refr.cost;
    const float eta = refr.eta_12;
#if WT_SS_POLAR
    {
        double c[5];
        ss_fresnel_dielectric(eta, abs_cosi, c);
        const float rs2 = (float)c[0], rp2 = (float)c[1], ts2 = (float)c[2], tp2 = (float)c[3], Z2 = (float)c[4];
        return fresnel_t{refr.t, cplx{eta, 0.f}, Z2, {rs2, 0}, {rp2, 0}, {ts2, 0}, {tp2, 0}, fminf_(1.f, Z2 * ts2 * ts2), fminf_(1.f, Z2 * tp2 * tp2)};
    }
#endif
    const float rs = (eta * abs_cosi - cost) / (eta * abs_cosi + cost);
    const float rp = (abs_cosi - eta * cost) / (abs_cosi + eta * cost);
    const float ts = rs + 1.f;
    const float tp = (rp + 1.f) * eta;
    const float Z = fabsf(cost / (eta * abs_cosi));
    return fresnel_t{refr.t, cplx{eta, 0.f}, Z, {rs, 0}, {rp, 0}, {ts, 0}, {tp, 0}, fminf_(1.f, Z * ts * ts), fminf_(1.f, Z * tp * tp)};
}
struct fresnel_conductor_t {
    cplx rs, rp;
};
// fresnel.hpp:128-144
WT_HD fresnel_conductor_t fresnel_reflection(cplx eta_12, vec3 w, vec3 n) {
    const float wn = dot(w, n);
    if ((eta_12.re == 1.f && eta_12.im == 0.f) || wn < 0.f) return {{0, 0}, {0, 0}};
#if WT_SS_POLAR
    {
        double c[4];
        ss_fresnel_conductor(eta_12.re, eta_12.im, wn, c);
        return {{(float)c[0], (float)c[1]}, {(float)c[2], (float)c[3]}};
    }
#endif
    const cplx t2 = cplx{1.f, 0.f} - (1.f - sqr(wn)) * (eta_12 * eta_12);
    const cplx t = csqrt(t2);
    const cplx i{wn, 0.f};
    return {(eta_12 * i - t) / (eta_12 * i + t), (i - eta_12 * t) / (i + eta_12 * t)};
}

// ---- Stokes / Mueller ------------------------------------------------------------------------------
struct stokes_t {
    float s[4];
};
struct mueller_t {
    float m[16];   // row-major, S' = M S
};
WT_HD stokes_t stokes_unpolarized(float I) { return {{I, 0.f, 0.f, 0.f}}; }
WT_HD stokes_t stokes_zero() { return {{0.f, 0.f, 0.f, 0.f}}; }
WT_HD bool stokes_is_unpolarized(const stokes_t& S) { return S.s[1] == 0.f && S.s[2] == 0.f && S.s[3] == 0.f; }
WT_HD stokes_t operator*(const stokes_t& S, float f) { return {{S.s[0] * f, S.s[1] * f, S.s[2] * f, S.s[3] * f}}; }
WT_HD stokes_t operator+(const stokes_t& a, const stokes_t& b) { return {{a.s[0] + b.s[0], a.s[1] + b.s[1], a.s[2] + b.s[2], a.s[3] + b.s[3]}}; }
WT_HD bool stokes_finite(const stokes_t& S) { return finitef(S.s[0]) && finitef(S.s[1]) && finitef(S.s[2]) && finitef(S.s[3]); }

// stokes.hpp:146-165
WT_HD stokes_t stokes_reorient(const stokes_t& S, const frame_t& cur, const frame_t& nw) {
    const vec3 tl = to_local(cur, nw.t), bl = to_local(cur, nw.b);
    const vec2 tou{tl.x, tl.y}, tov{bl.x, bl.y};
    const mat2 R = rotation_matrix2(vec2{1.f, 0.f}, tou);
    const vec2 S12 = mul(R, mul(R, vec2{S.s[1], S.s[2]}));
    stokes_t r{{S.s[0], S12.x, S12.y, S.s[3]}};
    const vec2 v = mul(R, vec2{0.f, 1.f});
    if (dot(v, tov) < 0.f) {
        r.s[2] = -r.s[2];
        r.s[3] = -r.s[3];
    }
    return r;
}

WT_HD mueller_t mueller_zero() {
    mueller_t M;
    for (int i = 0; i < 16; ++i) M.m[i] = 0.f;
    return M;
}
WT_HD mueller_t mueller_identity() {
    mueller_t M = mueller_zero();
    M.m[0] = M.m[5] = M.m[10] = M.m[15] = 1.f;
    return M;
}
WT_HD mueller_t mueller_handness_flip() {
    mueller_t M = mueller_zero();
    M.m[0] = M.m[5] = 1.f;
    M.m[10] = M.m[15] = -1.f;
    return M;
}
WT_HD mueller_t mueller_depolarizer(float s) {
    mueller_t M = mueller_zero();
    M.m[0] = s;
    return M;
}
WT_HD float mueller_mean_intensity(const mueller_t& M) { return M.m[0]; }
WT_HD mueller_t operator*(const mueller_t& A, float s) {
    mueller_t R;
    for (int i = 0; i < 16; ++i) R.m[i] = A.m[i] * s;
    return R;
}
WT_HD mueller_t operator*(float s, const mueller_t& A) { return A * s; }
WT_HD mueller_t operator+(const mueller_t& A, const mueller_t& B) {
    mueller_t R;
    for (int i = 0; i < 16; ++i) R.m[i] = A.m[i] + B.m[i];
    return R;
}
WT_HD mueller_t operator*(const mueller_t& A, const mueller_t& B) {
    mueller_t R;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += A.m[r * 4 + k] * B.m[k * 4 + c];
            R.m[r * 4 + c] = s;
        }
    return R;
}
WT_HD stokes_t operator*(const mueller_t& M, const stokes_t& S) {
    stokes_t R;
    for (int r = 0; r < 4; ++r) R.s[r] = M.m[r * 4 + 0] * S.s[0] + M.m[r * 4 + 1] * S.s[1] + M.m[r * 4 + 2] * S.s[2] + M.m[r * 4 + 3] * S.s[3];
    return R;
}
WT_HD bool mueller_finite(const mueller_t& M) {
    for (int i = 0; i < 16; ++i)
        if (!finitef(M.m[i])) return false;
    return true;
}

// mueller.hpp:208-222
WT_HD mueller_t mueller_rotation(vec2 t1, vec2 t2) {
    mat2 R = rotation_matrix2(t1, t2);
    R = mul(R, R);
    // math-convention entries of R*R: [[c,-s],[s,c]] with c=R.c0x, s=R.c0y
    const float c = R.c0x, s = R.c0y;
    mueller_t M = mueller_zero();
    M.m[0] = M.m[15] = 1.f;
    M.m[5] = c;
    M.m[6] = s;
    M.m[9] = R.c1x;   // = -s
    M.m[10] = R.c1y;  // = c
    return M;
}
// mueller.hpp:244-259
WT_HD mueller_t mueller_fresnel(cplx fs, cplx fp) {
#if WT_SS_POLAR
    {
        mueller_t M2;
        ss_mueller_from_jones(fs.re, fs.im, fp.re, fp.im, M2.m);
        return M2;
    }
#endif
    const float Rs = cnorm(fs), Rp = cnorm(fp);
    const float m00 = (Rs + Rp) / 2.f, m01 = (Rs - Rp) / 2.f;
    const cplx x = fp * conj(fs);
    mueller_t M = mueller_zero();
    M.m[0] = m00;
    M.m[1] = m01;
    M.m[4] = m01;
    M.m[5] = m00;
    M.m[10] = x.re;
    M.m[11] = x.im;
    M.m[14] = -x.im;
    M.m[15] = x.re;
    return M;
}
WT_HD mueller_t mueller_fresnel_reflection(cplx eta_12, vec3 w, vec3 n) {
    const fresnel_conductor_t f = fresnel_reflection(eta_12, w, n);
    return mueller_fresnel(f.rs, f.rp);
}
WT_HD mueller_t mueller_fresnel_transmission(cplx eta_12, vec3 w, vec3 n) {
    const fresnel_t f = fresnel(eta_12, w, n);
    return f.Z * mueller_fresnel(f.ts, f.tp);
}
WT_HD mueller_t mueller_fresnel_rt(cplx eta_12, bool reflection, vec3 w, vec3 n) {
    return reflection ? mueller_fresnel_reflection(eta_12, w, n) : mueller_fresnel_transmission(eta_12, w, n);
}

// M(S, Sin, Min): mueller.hpp:134-144
WT_HD stokes_t mueller_apply(const mueller_t& M, const stokes_t& S, const frame_t& Sin, const frame_t& Min) {
    if (stokes_is_unpolarized(S)) return M * S;
    return M * stokes_reorient(S, Sin, Min);
}
// M(S, Sin, Min, Sout, Mout): mueller.hpp:153-163
WT_HD stokes_t mueller_apply(const mueller_t& M, const stokes_t& S, const frame_t& Sin, const frame_t& Min, const frame_t& Sout,
                             const frame_t& Mout) {
    const stokes_t r = M * stokes_reorient(S, Sin, Min);
    return stokes_reorient(r, Mout, Sout);
}
// compose (mueller.hpp:402-415)
WT_HD mueller_t mueller_compose(const mueller_t& M1, const mueller_t& M2, const frame_t& M1in, const frame_t& M2out) {
    const vec3 tl = to_local(M1in, M2out.t);
    mueller_t R = mueller_rotation(vec2{tl.x, tl.y}, vec2{1.f, 0.f});
    if (handness(M1in) != handness(M2out)) R = mueller_handness_flip() * R;
    return M1 * R * M2;
}
// change_incident_frame (mueller.hpp:168-179)
WT_HD mueller_t mueller_change_incident_frame(const mueller_t& M, const frame_t& old_frame, const frame_t& new_frame) {
    const vec3 tl = to_local(old_frame, new_frame.t);
    mueller_t R = mueller_rotation(vec2{tl.x, tl.y}, vec2{1.f, 0.f});
    if (handness(old_frame) != handness(new_frame)) R = R * mueller_handness_flip();
    return M * R;
}

}   // namespace wt
