// wave_tracer_amd — core math for the per-sample wave-optical integrator path.
//
// Plain-float restatement of the pieces of wave_tracer's L0 math layer that the hot path
// touches (glm vectors, frames, error-free transforms, 2x2 QR/SVD).  Units are carried by
// convention instead of mp-units strong types:
//     lengths  : metres          wavenumber k : 1/mm          wavelength : metres
// so the dimensionless product k*length is  k * length * 1000  (see k_times_len()).
//
// Reference: include/wt/math/frame.hpp, include/wt/math/eft/eft.hpp:118-185,
//            include/wt/math/linalg.hpp, include/wt/math/rotation.hpp
//
// Everything here is `WT_HD` (host+device) so the very same functions are compiled by hipcc for
// gfx950 kernels and by g++ for host-side scene baking and for the CPU checker in oracle/.
#pragma once

#include <cmath>
#include <cstdint>
#include <cfloat>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
// always_inline: a real call on AMDGPU passes the big by-reference PODs (walk_t, vertex_t, ...) through scratch memory
#define WT_HD __host__ __device__ inline __attribute__((always_inline))
#define WT_D __device__ inline __attribute__((always_inline))   // (coop_gather, called from four kernels, was left as a CALL by one build of round 5: k_flux_tasks at 262 VGPRs, one wavefront per SIMD)
#else
#define WT_HD inline
#endif

namespace wt {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kInvTwoPi = 0.15915494309189533577f;
constexpr float kPi2 = 1.57079632679489661923f;
constexpr float kPi4 = 0.78539816339744830962f;
constexpr float kSqrtPi = 1.77245385090551602730f;
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kSqrt2 = 1.41421356237309504880f;
#define WT_INF (__builtin_huge_valf())

WT_HD float sqr(float x) { return x * x; }
// Square root for CONSERVATIVE culling tests only (never for a value that reaches a result): on the device the hardware's v_sqrt_f32 (1 ulp,
// one instruction) instead of the correctly rounded sequence (~15 instructions, eight times per BVH node visit in cone_box_outside); the
// tests that use it keep margins four orders of magnitude above the difference.
WT_HD float cull_sqrtf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return __builtin_sqrtf(x);
#endif
}
// std::min / std::max semantics exactly (matters when an operand is NaN, e.g. inf*0 in cone axes):
//   min(a,b) = (b<a) ? b : a      max(a,b) = (a<b) ? b : a
WT_HD float fminf_(float a, float b) { return (b < a) ? b : a; }
WT_HD float fmaxf_(float a, float b) { return (a < b) ? b : a; }
WT_HD float clampf(float x, float a, float b) { return x < a ? a : (x > b ? b : x); }
WT_HD float clamp01(float x) { return clampf(x, 0.f, 1.f); }
WT_HD float signf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }   // glm::sign
WT_HD float mixf(float a, float b, float t) { return a * (1.f - t) + b * t; }   // glm::mix
// bit test (an arithmetic test such as x-x==0 is broken by fma contraction: a*b - a*b becomes the rounding residual)
WT_HD bool finitef(float x) {
    uint32_t u;
    __builtin_memcpy(&u, &x, 4);
    return (u & 0x7f800000u) != 0x7f800000u;
}
WT_HD float fractf(float x) { return x - floorf(x); }

// ---- error-free transforms (include/wt/math/eft/eft.hpp) ------------------------------------
// a*b - c*d with one fma-based correction (Kahan).
WT_HD float diff_prod(float a, float b, float c, float d) {
    const float cd = c * d;
    const float ret = fmaf(a, b, -cd);
    return ret + fmaf(-c, d, cd);
}
WT_HD float sum_prod(float a, float b, float c, float d) { return diff_prod(a, b, -c, d); }
WT_HD float two_prod(float& err, float a, float b) {
    const float p = a * b;
    err = fmaf(a, b, -p);
    return p;
}
WT_HD float two_sum(float& err, float a, float b) {
    const float s = a + b;
    const float bb = s - a;
    err = (a - (s - bb)) + (b - bb);
    return s;
}

// ---- vectors --------------------------------------------------------------------------------
struct vec2 {
    float x, y;
};
struct vec3 {
    float x, y, z;
};
WT_HD vec2 mk2(float x, float y) { return vec2{x, y}; }
WT_HD vec3 mk3(float x, float y, float z) { return vec3{x, y, z}; }
WT_HD vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
WT_HD vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
WT_HD vec2 operator-(vec2 a) { return {-a.x, -a.y}; }
WT_HD vec2 operator*(vec2 a, float s) { return {a.x * s, a.y * s}; }
WT_HD vec2 operator*(float s, vec2 a) { return {a.x * s, a.y * s}; }
WT_HD vec2 operator*(vec2 a, vec2 b) { return {a.x * b.x, a.y * b.y}; }
WT_HD vec2 operator/(vec2 a, float s) { return {a.x / s, a.y / s}; }
WT_HD vec2 operator/(vec2 a, vec2 b) { return {a.x / b.x, a.y / b.y}; }
WT_HD float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
WT_HD float length2(vec2 a) { return dot(a, a); }
WT_HD float length(vec2 a) { return sqrtf(dot(a, a)); }
WT_HD vec2 normalize(vec2 a) { return a / length(a); }

WT_HD vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
WT_HD vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
WT_HD vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
WT_HD vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
WT_HD vec3 operator*(float s, vec3 a) { return {a.x * s, a.y * s, a.z * s}; }
WT_HD vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
WT_HD vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
WT_HD vec3& operator+=(vec3& a, vec3 b) {
    a = a + b;
    return a;
}
WT_HD float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
WT_HD vec3 cross(vec3 a, vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
WT_HD float length2(vec3 a) { return dot(a, a); }
WT_HD float length(vec3 a) { return sqrtf(dot(a, a)); }
WT_HD vec3 normalize(vec3 a) { return a / length(a); }
WT_HD vec3 vabs(vec3 a) { return {fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
WT_HD float max_element(vec3 a) { return fmaxf_(a.x, fmaxf_(a.y, a.z)); }
WT_HD float min_element(vec3 a) { return fminf_(a.x, fminf_(a.y, a.z)); }
WT_HD vec3 vmin(vec3 a, vec3 b) { return {fminf_(a.x, b.x), fminf_(a.y, b.y), fminf_(a.z, b.z)}; }
WT_HD vec3 vmax(vec3 a, vec3 b) { return {fmaxf_(a.x, b.x), fmaxf_(a.y, b.y), fmaxf_(a.z, b.z)}; }
WT_HD bool veq(vec3 a, vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
WT_HD bool vfinite(vec3 a) { return finitef(a.x) && finitef(a.y) && finitef(a.z); }
WT_HD float comp(const vec3& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
WT_HD vec3 mix3(vec3 a, vec3 b, float t) { return a * (1.f - t) + b * t; }
WT_HD vec2 mix2(vec2 a, vec2 b, float t) { return a * (1.f - t) + b * t; }

// eft::dot (compensated dot product; eft.hpp "Vector math with some error-free transformation")
WT_HD float eft_dot(vec3 a, vec3 b) {
    float d = 0.f, err = 0.f, e1, e2;
    float t = two_prod(e1, a.x, b.x);
    d = two_sum(e2, d, t);
    err = err + e1 + e2;
    t = two_prod(e1, a.y, b.y);
    d = two_sum(e2, d, t);
    err = err + e1 + e2;
    t = two_prod(e1, a.z, b.z);
    d = two_sum(e2, d, t);
    err = err + e1 + e2;
    return d + err;
}

// ---- complex --------------------------------------------------------------------------------
struct cplx {
    float re, im;
};
WT_HD cplx mkc(float re, float im = 0.f) { return cplx{re, im}; }
WT_HD cplx operator+(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
WT_HD cplx operator-(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
WT_HD cplx operator-(cplx a) { return {-a.re, -a.im}; }
WT_HD cplx operator*(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
WT_HD cplx operator*(cplx a, float s) { return {a.re * s, a.im * s}; }
WT_HD cplx operator*(float s, cplx a) { return {a.re * s, a.im * s}; }
WT_HD cplx operator/(cplx a, float s) { return {a.re / s, a.im / s}; }
WT_HD cplx conj(cplx a) { return {a.re, -a.im}; }
WT_HD float cnorm(cplx a) { return a.re * a.re + a.im * a.im; }   // std::norm = |a|^2
WT_HD float cabs(cplx a) { return sqrtf(cnorm(a)); }
WT_HD cplx operator/(cplx a, cplx b) {
    const float d = cnorm(b);
    return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
WT_HD cplx csqrt(cplx z) {   // principal branch
    const float r = cabs(z);
    if (r == 0.f) return {0.f, 0.f};
    float re = sqrtf(fmaxf_(0.f, 0.5f * (r + z.re)));
    float im = sqrtf(fmaxf_(0.f, 0.5f * (r - z.re)));
    if (z.im < 0.f) im = -im;
    return {re, im};
}
WT_HD cplx cpolar(float r, float theta) { return {r * cosf(theta), r * sinf(theta)}; }
WT_HD bool ceq(cplx a, cplx b) { return a.re == b.re && a.im == b.im; }

// ---- 2x2 matrix with glm semantics: m[col][row] ----------------------------------------------
struct mat2 {
    float c0x, c0y, c1x, c1y;   // column 0 = (c0x,c0y), column 1 = (c1x,c1y)
};
WT_HD mat2 mkmat2(vec2 c0, vec2 c1) { return mat2{c0.x, c0.y, c1.x, c1.y}; }
WT_HD vec2 mul(const mat2& m, vec2 v) { return {m.c0x * v.x + m.c1x * v.y, m.c0y * v.x + m.c1y * v.y}; }   // M*v
WT_HD vec2 mul(vec2 v, const mat2& m) { return {v.x * m.c0x + v.y * m.c0y, v.x * m.c1x + v.y * m.c1y}; }   // v*M
WT_HD mat2 mul(const mat2& a, const mat2& b) {
    const vec2 c0 = mul(a, vec2{b.c0x, b.c0y});
    const vec2 c1 = mul(a, vec2{b.c1x, b.c1y});
    return mkmat2(c0, c1);
}
WT_HD mat2 transpose(const mat2& m) { return mat2{m.c0x, m.c1x, m.c0y, m.c1y}; }
WT_HD float determinant(const mat2& m) { return m.c0x * m.c1y - m.c1x * m.c0y; }
WT_HD mat2 inverse(const mat2& m) {
    const float id = 1.f / determinant(m);
    return mat2{m.c1y * id, -m.c0y * id, -m.c1x * id, m.c0x * id};
}

// ---- frames (include/wt/math/frame.hpp) -------------------------------------------------------
struct frame_t {
    vec3 t, b, n;
};
WT_HD vec3 to_local(const frame_t& f, vec3 v) { return {dot(v, f.t), dot(v, f.b), dot(v, f.n)}; }
WT_HD vec3 to_world(const frame_t& f, vec3 v) { return f.t * v.x + f.b * v.y + f.n * v.z; }
WT_HD vec3 to_world(const frame_t& f, vec2 v) { return f.t * v.x + f.b * v.y; }
// frame_t::to_local(vec2) uses only the xy components of t and b (frame.hpp:22-27)
WT_HD vec2 to_local2(const frame_t& f, vec2 v) { return {v.x * f.t.x + v.y * f.t.y, v.x * f.b.x + v.y * f.b.y}; }
WT_HD float handness(const frame_t& f) { return dot(cross(f.n, f.t), f.b) > 0.f ? 1.f : -1.f; }
WT_HD frame_t flip(const frame_t& f) { return {-f.t, -f.b, -f.n}; }

// frame.hpp:156-172 — branch on |n.x|>|n.y| (must be identical on CPU and device)
WT_HD frame_t build_orthogonal_frame(vec3 n) {
    vec3 b;
    if (fabsf(n.x) > fabsf(n.y)) {
        const float x = 1.f / sqrtf(sqr(n.x) + sqr(n.z));
        b = vec3{x * n.z, 0.f, -x * n.x};
    } else {
        const float x = 1.f / sqrtf(sqr(n.y) + sqr(n.z));
        b = vec3{0.f, x * n.z, -x * n.y};
    }
    return frame_t{cross(b, n), b, n};
}
// frame.hpp:140-152
WT_HD frame_t build_shading_frame(vec3 n, vec3 dpdu) {
    if (dpdu.x == 0.f && dpdu.y == 0.f && dpdu.z == 0.f) return build_orthogonal_frame(n);
    const vec3 t = normalize(dpdu - n * dot(n, dpdu));
    const vec3 b = normalize(cross(n, t));
    return frame_t{cross(b, n), b, n};
}

// ---- 2-D rotation from unit vector `from` to unit vector `to` (rotation.hpp:55-66) -----------
// glm column-major: R[0]=(X, xa*yb-xb*ya), R[1]=(xb*ya-xa*yb, X)
WT_HD mat2 rotation_matrix2(vec2 from, vec2 to) {
    const float X = sum_prod(from.x, to.x, from.y, to.y);
    return mat2{X, diff_prod(from.x, to.y, to.x, from.y), diff_prod(to.x, from.y, from.x, to.y), X};
}

// ---- 2x2 QR / SVD (include/wt/math/linalg.hpp:18-127) -----------------------------------------
struct svd_t {
    float Ucos, Usin, Vcos, Vsin, sigma1, sigma2;
};
WT_HD svd_t svd2(const mat2& A) {
    // QR
    float a = A.c0x, b = A.c1x, c = A.c0y, d = A.c1y;
    float x, y, z, Qc, Qs;
    if (c == 0.f) {
        x = a;
        y = b;
        z = d;
        Qc = 1.f;
        Qs = 0.f;
    } else {
        const float mm = fmaxf_(fabsf(c), fabsf(d));
        const float rm = 1.f / mm;
        c *= rm;
        d *= rm;
        const float r = sqrtf(c * c + d * d);
        const float l = 1.f / r;
        x = diff_prod(a, d, b, c) * l;
        y = sum_prod(a, c, b, d) * l;
        z = mm * r;
        Qs = -c * l;
        Qc = d * l;
    }
    float c2 = Qc, s2 = Qs;
    const float n = fmaxf_(fabsf(x), fabsf(y));
    if (n == 0.f) return svd_t{1.f, 0.f, c2, s2, A.c0x, A.c1y};
    const float numer = (z - x) * (z + x) + sqr(y);
    const float tt = numer != 0.f ? numer / (n * x * y) : 0.f;
    float t = 2.f * (tt >= 0.f ? 1.f : -1.f) / (fabsf(tt) + sqrtf(sqr(tt) + 4.f));
    const float c1 = 1.f / sqrtf(1.f + sqr(t));
    const float s1 = c1 * t;
    const float usa = diff_prod(c1, x, s1, y);
    const float usb = sum_prod(s1, x, c1, y);
    const float usc = -s1 * z;
    const float usd = c1 * z;
    t = sum_prod(c1, c2, s1, s2);
    s2 = diff_prod(c2, s1, c1, s2);
    c2 = t;
    float sigma1 = sqrtf(sqr(usa) + sqr(usc));
    float sigma2 = sqrtf(sqr(usb) + sqr(usd));
    float dmax = fmaxf_(sigma1, sigma2);
    const float usmax1 = sigma2 > sigma1 ? usd : usa;
    const float usmax2 = sigma2 > sigma1 ? usb : -usc;
    const float signsigma1 = (x * z > 0.f) ? 1.f : -1.f;
    dmax *= sigma2 > sigma1 ? signsigma1 : 1.f;
    sigma2 *= signsigma1;
    const float r = 1.f / dmax;
    return svd_t{dmax != 0.f ? usmax1 * r : 1.f, dmax != 0.f ? usmax2 * r : 0.f, c2, s2, sigma1, sigma2};
}

// ---- unit helpers ---------------------------------------------------------------------------
// k [1/mm] times length [m] -> dimensionless  (mp-units does this implicitly in the reference)
WT_HD float k_times_len(float k_mm, float len_m) { return k_mm * len_m * 1000.f; }
// wavelength in metres from k in 1/mm
WT_HD float wavenum_to_wavelen_m(float k_mm) { return kTwoPi / (k_mm * 1000.f); }

// sinc as used by fraunhofer/fsd.hpp (boost-style, include/wt/math/common.hpp:416)
WT_HD float sincf_(float x) {
    if (fabsf(x) >= 0.018581361171917516f) return sinf(x) / x;
    float result = 1.f;
    if (fabsf(x) >= FLT_EPSILON) {
        const float x2 = x * x;
        result -= x2 / 6.f;
        if (fabsf(x) >= 0.00034526698300124390f) result += (x2 * x2) / 120.f;
    }
    return result;
}

// generic "struct as array of 32-bit words" accessors of the per-walk state arrays, RECORD-MAJOR: word i of element idx lives at
// base[idx*stride + i], stride = words per record.  (Rounds 1-2 kept the state word-interleaved — base[i*n + idx] — which coalesces
// only while the walks of a wavefront are consecutive: true in the first round, but after the first queue compaction a wavefront's 64
// walks are scattered over the batch and every word of every lane touched its own 64-byte line to use 4 bytes of it.  With one
// contiguous record per walk a lane reads whole lines of its own record whatever the order of the walks: measured HBM-side traffic
// of a pass DESIGN.md §4.)  The CPU checker keeps one record per array (idx = 0).
// (WT_NT_RECORDS, device only, OFF: marks the record accesses non-temporal — the records are a stream of gigabytes per round, each word used
// once per kernel, next to BVH nodes and triangles that the traversal kernels keep re-reading from L2.  Measured in round 4: 14 % SLOWER per
// pass (21.6 / 22.0 vs 25.3 / 25.1 Msamples/s): what one kernel of a round writes the next one reads, and much of that still comes from the
// L2 / the memory-side cache when it is allowed to stay there.)
#if defined(__HIP_DEVICE_COMPILE__) && defined(WT_NT_RECORDS)
#define WT_REC_LOAD(p) __builtin_nontemporal_load(p)
#define WT_REC_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define WT_REC_LOAD(p) (*(p))
#define WT_REC_STORE(v, p) (*(p) = (v))
#endif
template <class T>
WT_HD void soa_store(uint32_t* base, size_t stride, size_t idx, const T& v) {
    static_assert(sizeof(T) % 4 == 0, "POD of 32-bit words expected");
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) WT_REC_STORE(w[i], base + idx * stride + i);
}
template <class T>
WT_HD void soa_load(const uint32_t* base, size_t stride, size_t idx, T& v) {
    static_assert(sizeof(T) % 4 == 0, "POD of 32-bit words expected");
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) w[i] = WT_REC_LOAD(base + idx * stride + i);
}

}   // namespace wt
