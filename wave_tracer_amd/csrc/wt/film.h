// wave_tracer_amd — film splatting.
//
// Reference: include/wt/sensor/film/film.hpp:75,130 (radius), 214-286 (splat / splat_direct), 310-342 (weights),
//            include/wt/sensor/film/film_storage.hpp:196-252 (write_block, write_light_splat), 256-287 (develop),
//            include/wt/math/distribution/gaussian1d.hpp:100-106 (filter integral).
//
// Layout: value[H][W][C][S] (sum of w*v), weight[H][W] (sum of w; the reference stores the same weight once per
// channel), light[H][W][C][S] (sum of light-image splats); S = 1 (intensity) or, for polarimetric sensors, the 4 Stokes
// components (FilmSampleT = vec4, film.hpp:214-286; one developed image per component, src/main.cpp:405-430).  All f64.  Accumulation is atomic: on gfx950
// atomicAdd(double) is a single global_atomic_add_f64.
#pragma once
#include "sources.h"

namespace wt {

struct film_t {
    double* value;
    double* weight;
    double* light;
    uint32_t width, height, channels;
};
WT_HD uint32_t film_stokes(const sensor_t& s) { return s.polarimetric ? 4u : 1u; }
WT_HD uint32_t film_planes(const sensor_t& s) { return s.channels * film_stokes(s); }

WT_HD void film_add(double* p, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsafeAtomicAdd(p, v);
#else
    uint64_t* ip = reinterpret_cast<uint64_t*>(p);
    uint64_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
    for (;;) {
        double d;
        __builtin_memcpy(&d, &old, 8);
        d += v;
        uint64_t nw;
        __builtin_memcpy(&nw, &d, 8);
        if (__atomic_compare_exchange_n(ip, &old, nw, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
    }
#endif
}

WT_HD float erf_clamped(float x) { return fabsf(x) >= 3.5f ? signf(x) : erff(x); }   // erf_lut saturates at 3.5

// compute_rfilter_weights (film.hpp:310-342) for a 2-D film, radius r<=2
struct rfilter_weights_t {
    float wx[5], wy[5];
    float recp_total;
};
WT_HD rfilter_weights_t film_rfilter_weights(const sensor_t& s, vec2 o) {
    rfilter_weights_t w;
    const int r = s.rf_radius;
    const float n = s.rfilter_sigma > 0.f ? kInvSqrt2 / s.rfilter_sigma : 0.f;
    for (int x = -r; x <= r; ++x) {
        if (s.rfilter_sigma > 0.f) {
            w.wx[x + r] = (erf_clamped((x + o.x + .5f) * n) - erf_clamped((x + o.x - .5f) * n)) / 2.f;
            w.wy[x + r] = (erf_clamped((x + o.y + .5f) * n) - erf_clamped((x + o.y - .5f) * n)) / 2.f;
        } else {
            w.wx[x + r] = (x + o.x - .5f <= 0.f && 0.f <= x + o.x + .5f) ? 1.f : 0.f;
            w.wy[x + r] = (x + o.y - .5f <= 0.f && 0.f <= x + o.y + .5f) ? 1.f : 0.f;
        }
    }
    float tw = 0.f;
    for (int y = -r; y <= r; ++y)
        for (int x = -r; x <= r; ++x) tw += fmaxf_(0.f, w.wx[x + r] * w.wy[y + r]);
    w.recp_total = tw > 0.f ? 1.f / tw : 0.f;
    return w;
}

// film_t::splat + film_storage_t::write_block: primary image (value, weight)
WT_HD void film_splat(const scene_t& sc, const film_t& film, const sensor_element_t& el, const stokes_t& sample, float k) {
    const sensor_t& s = sc.sensor;
    const int r = s.rf_radius;
    const uint32_t S = film_stokes(s), P = s.channels * S;
    const rfilter_weights_t rw = film_rfilter_weights(s, el.offset);
    float val[16];
    for (uint32_t c = 0; c < s.channels; ++c) {
        const float f = spectrum_f(sc, s.response_spec[c], k);
        // avoid NaNs, infs and negatives (of the intensity); cannot bail out on zero: the weights must be summed
        bool ok = true;
        for (uint32_t q = 0; q < S; ++q) {
            val[c * S + q] = sample.s[q] * f;
            ok = ok && finitef(val[c * S + q]);
        }
        ok = ok && val[c * S] >= 0.f;
        if (!ok)
            for (uint32_t q = 0; q < S; ++q) val[c * S + q] = 0.f;
    }
    for (int dy = -r; dy <= r; ++dy) {
        const int y = (int)el.y + dy;
        if (y < 0 || y >= (int)film.height) continue;
        for (int dx = -r; dx <= r; ++dx) {
            const int x = (int)el.x + dx;
            if (x < 0 || x >= (int)film.width) continue;
            const float w = fmaxf_(0.f, rw.wx[dx + r] * rw.wy[dy + r]) * rw.recp_total;
            const size_t pix = (size_t)y * film.width + x;
            film_add(&film.weight[pix], (double)w);
            for (uint32_t c = 0; c < P; ++c) film_add(&film.value[pix * P + c], (double)(w * val[c]));
        }
    }
}
// film_t::splat_direct + film_storage_t::write_light_splat: light image
WT_HD void film_splat_direct(const scene_t& sc, const film_t& film, const sensor_element_t& el, const stokes_t& sample, float k) {
    const sensor_t& s = sc.sensor;
    const int r = s.rf_radius;
    const uint32_t S = film_stokes(s), P = s.channels * S;
    const rfilter_weights_t rw = film_rfilter_weights(s, el.offset);
    for (uint32_t c = 0; c < s.channels; ++c) {
        const float f = spectrum_f(sc, s.response_spec[c], k);
        float val[4];
        bool ok = true;
        for (uint32_t q = 0; q < S; ++q) {
            val[q] = sample.s[q] * f;
            ok = ok && finitef(val[q]);
        }
        if (!ok || val[0] <= 0.f) continue;
        for (int dy = -r; dy <= r; ++dy) {
            const int y = (int)el.y + dy;
            if (y < 0 || y >= (int)film.height) continue;
            for (int dx = -r; dx <= r; ++dx) {
                const int x = (int)el.x + dx;
                if (x < 0 || x >= (int)film.width) continue;
                const float w = fmaxf_(0.f, rw.wx[dx + r] * rw.wy[dy + r]) * rw.recp_total;
                for (uint32_t q = 0; q < S; ++q) film_add(&film.light[((size_t)y * film.width + x) * P + c * S + q], (double)(w * val[q]));
            }
        }
    }
}

}   // namespace wt
