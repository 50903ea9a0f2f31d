// wave_tracer_amd — counter-based RNG (Philox-4x32-10) and the sampler warps of the reference.
//
// The reference's path sampler is a thread_local mt19937_64 seeded from random_device/time
// (include/wt/util/seeded_mt19937_64.hpp:33-51) and is not reproducible.  We replace it with a
// counter-based generator: key = run seed, counter = (sample id lo, sample id hi, stream id, draw index/4), so
// that every (pixel, sample, stream) owns an independent, order-free sequence; the CPU checker and the HIP
// kernels therefore consume *identical* random numbers regardless of scheduling.
//
// Warps: include/wt/sampler/sampler.hpp:25-312 (concentric disk, cosine hemisphere, uniform cone, Box-Muller,
// uniform triangle, discrete).
#pragma once
#include "core.h"

namespace wt {

struct philox_t {
    uint32_t c[4];
    uint32_t k[2];
};
WT_HD void philox_mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    const uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}
WT_HD void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        philox_mulhilo(0xD2511F53u, c0, hi0, lo0);
        philox_mulhilo(0xCD9E8D57u, c2, hi1, lo1);
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

// streams of one sample
enum rng_stream_e : uint32_t {
    STREAM_SCENE = 0,         // scene sampler: emitter/spectrum/emitter beam/sensor sample
    STREAM_SENSOR_WALK = 1,   // sensor subpath
    STREAM_EMITTER_WALK = 2,  // emitter subpath
    STREAM_CONNECT = 16,      // + t*32 + s : one stream per (s,t) connection
};

struct sampler_t {
    uint64_t sample_id;
    uint32_t stream;
    uint32_t seed_lo, seed_hi;
    uint32_t draws;     // number of floats consumed so far
    uint32_t buf[4];
};
WT_HD sampler_t make_sampler(uint64_t seed, uint64_t sample_id, uint32_t stream, uint32_t draws = 0) {
    sampler_t s;
    s.sample_id = sample_id;
    s.stream = stream;
    s.seed_lo = (uint32_t)seed;
    s.seed_hi = (uint32_t)(seed >> 32);
    s.draws = draws;
    if (draws & 3u) {
        const uint32_t ctr[4] = {(uint32_t)sample_id, (uint32_t)(sample_id >> 32), stream, draws >> 2};
        const uint32_t key[2] = {s.seed_lo, s.seed_hi};
        philox4x32_10(ctr, key, s.buf);
    }
    return s;
}
// Repositions a sampler at draw number `draws` of its stream (counter-based: any position is reachable in O(1)).
WT_HD void sampler_seek(sampler_t& s, uint32_t draws) {
    s.draws = draws;
    if (draws & 3u) {
        const uint32_t ctr[4] = {(uint32_t)s.sample_id, (uint32_t)(s.sample_id >> 32), s.stream, draws >> 2};
        const uint32_t key[2] = {s.seed_lo, s.seed_hi};
        philox4x32_10(ctr, key, s.buf);
    }
}
WT_HD sampler_t sampler_at(const sampler_t& s, uint32_t draws) {
    sampler_t r = s;
    sampler_seek(r, draws);
    return r;
}
// uniform float in [0,1): 24 random bits
WT_HD float sampler_r(sampler_t& s) {
    const uint32_t lane = s.draws & 3u;
    if (lane == 0) {
        const uint32_t ctr[4] = {(uint32_t)s.sample_id, (uint32_t)(s.sample_id >> 32), s.stream, s.draws >> 2};
        const uint32_t key[2] = {s.seed_lo, s.seed_hi};
        philox4x32_10(ctr, key, s.buf);
    }
    s.draws++;
    return float(s.buf[lane] >> 8) * (1.f / 16777216.f);
}
WT_HD vec2 sampler_r2(sampler_t& s) {
    const float a = sampler_r(s);
    const float b = sampler_r(s);
    return {a, b};
}
WT_HD vec3 sampler_r3(sampler_t& s) {
    const float a = sampler_r(s);
    const float b = sampler_r(s);
    const float c = sampler_r(s);
    return {a, b, c};
}

// ---- warps (sampler.hpp) --------------------------------------------------------------------------
WT_HD vec2 concentric_disk(vec2 u) {
    const vec2 offset = 2.f * u - vec2{1.f, 1.f};
    float rr, theta;
    if (offset.x == 0.f && offset.y == 0.f) {
        rr = 0.f;
        theta = 0.f;
    } else if (fabsf(offset.x) > fabsf(offset.y)) {
        rr = offset.x;
        theta = kPi4 * (offset.y / offset.x);
    } else {
        rr = offset.y;
        theta = kPi2 - kPi4 * (offset.x / offset.y);
    }
    return rr * vec2{cosf(theta), sinf(theta)};
}
WT_HD vec3 cosine_hemisphere(vec2 u) {
    const vec2 d = concentric_disk(u);
    const float z = sqrtf(fmaxf_(0.f, 1.f - sqr(d.x) - sqr(d.y)));
    return {d.x, d.y, z};
}
WT_HD float cosine_hemisphere_pdf(float cosine) { return kInvPi * cosine; }
WT_HD vec3 uniform_cone(float solid_angle, vec2 u) {
    const float cos_theta_max = 1.f - kInvTwoPi * solid_angle;
    const float cos_theta = 1.f + u.x * (cos_theta_max - 1.f);
    const float sin_theta = sqrtf(fmaxf_(0.f, 1.f - sqr(cos_theta)));
    const float phi = kTwoPi * u.y;
    return {cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta};
}
WT_HD float uniform_cone_pdf(float solid_angle) { return 1.f / solid_angle; }
WT_HD vec2 normal2d(vec2 u) {
    const float r = sqrtf(-2.f * logf(1.f - u.x));
    const float theta = kTwoPi * u.y;
    return {r * cosf(theta), r * sinf(theta)};
}
WT_HD vec2 uniform_triangle(vec2 u) {
    if (u.x + u.y > 1.f) u = vec2{1.f, 1.f} - u;
    return u;
}

// icdf of a tabulated cdf with n+1 entries (cdf[0]=0, cdf[n]=1): returns index i with cdf[i] <= u < cdf[i+1]
WT_HD uint32_t cdf_icdf(const float* cdf, uint32_t n, float u) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] <= u)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

}   // namespace wt
