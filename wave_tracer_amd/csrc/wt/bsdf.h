// wave_tracer_amd — flattened BSDF evaluation.
//
// Reference: include/wt/bsdf/bsdf.hpp:32-134, include/wt/bsdf/common.hpp:27-90,
//            src/bsdf/diffuse.cpp:23-71, src/bsdf/dielectric.cpp:26-72 + include/wt/bsdf/dielectric.hpp,
//            src/bsdf/surface_spm.cpp:28-201 + include/wt/bsdf/surface_spm.hpp,
//            src/bsdf/two_sided.cpp:20-55, include/wt/bsdf/scale.hpp,
//            include/wt/interaction/surface_profile/fractal.hpp:26-238, src/interaction/surface_profile/fractal.cpp:27-69,
//            include/wt/interaction/surface_profile/dirac.hpp, include/wt/sampler/density.hpp (pdf tagging).
//
// The reference's virtual wrapper chain (two_sided -> scale -> leaf) is flattened into one material record
// (scene.h: material_t); textures are constants in this round.
#pragma once
#include "beam.h"
#include "rng.h"

namespace wt {

// ---- tagged probability densities (sampler/density.hpp): discrete mass stored as -mass ---------------
WT_HD float pd_discrete(float mass) { return -mass; }
WT_HD bool pd_is_discrete(float p) { return __builtin_signbit(p); }
WT_HD float pd_density_or_zero(float p) { return pd_is_discrete(p) ? 0.f : p; }
WT_HD float pd_mass(float p) { return -p; }

constexpr uint32_t LOBE_SPECULAR = 0, LOBE_SCATTERED = 1;

struct bsdf_sample_t {
    vec3 wo;
    float dpd;   // tagged
    float eta;   // real part of the relative IOR crossed
    mueller_t M;   // weighted bsdf (bsdf/pdf)
    bool valid;
};

// ---- fractal surface profile ---------------------------------------------------------------------
struct fractal_params_t {
    float T;   // [mm^2]
    float sigma2_norm;
    float alpha;
};
WT_HD fractal_params_t fractal_params(const material_t& m, float k) {
    const float meank = kTwoPi / 550e-6f;   // wavelen_to_wavenum(550nm) in 1/mm
    const float max_GGX_alpha = .75f, maxT = sqr(70.f);
    const float alpha2 = sqr(clampf(m.roughness, 0.f, max_GGX_alpha));
    const float T = fminf_(maxT, (1.f - alpha2) / (4.f * sqr(meank) * alpha2));
    const float x = 1.f + k * k * T;
    const float p = m.gamma == 3.f ? x : powf(x, (m.gamma - 1.f) / 2.f);
    return {T, 1.f / (1.f - 1.f / p), sqr(m.roughness / 9.f)};
}
// z: spatial frequency [1/mm]
WT_HD float fractal_psd(const material_t& m, const fractal_params_t& pr, vec2 z, float k) {
    const float x = 1.f + pr.T * dot(z, z);
    const float p = m.gamma == 3.f ? x * x : powf(x, (m.gamma + 1.f) / 2.f);
    return pr.sigma2_norm * (kInvTwoPi * k * k * (m.gamma - 1.f) * pr.T / p);
}
// ---- gaussian profile (interaction/surface_profile/gaussian.hpp:25-255) -----------------------------------------------------
struct gaussian_params_t {
    float sigma2;   // [1/mm^2]
    float sigma2_norm;
    float alpha;
};
// gaussian_t::gaussian_params (gaussian.hpp:94-117)
WT_HD gaussian_params_t gaussian_params(const material_t& m, float k) {
    gaussian_params_t r;
    if (m.gauss_sigma > 0.f) {
        r.sigma2 = sqr(m.gauss_sigma);
        r.alpha = r.sigma2;
    } else {
        const float meank = kTwoPi / 550e-6f;
        const float max_GGX_alpha = .75f, maxT = sqr(70.f);
        const float alpha2 = sqr(clampf(m.roughness, 0.f, max_GGX_alpha));
        const float T = fminf_(maxT, (1.f - alpha2) / (4.f * sqr(meank) * alpha2));
        r.sigma2 = 1.f / T;
        r.alpha = sqr(m.roughness / 9.f);
    }
    r.sigma2_norm = 1.f / (1.f - expf(-(k * k / 2.f / r.sigma2)));
    return r;
}
WT_HD float gaussian_psd(const gaussian_params_t& pr, vec2 z, float k) {
    const float e = expf(-(dot(z, z) / 2.f / pr.sigma2));
    return e <= FLT_EPSILON ? 0.f : pr.sigma2_norm * (kInvTwoPi / pr.sigma2 * k * k * e);
}
WT_HD float gaussian_max_phi(float r, float l) {
    return (r < FLT_EPSILON || l < FLT_EPSILON) ? kPi : fmaxf_(1e-2f, acosf(clampf((sqr(r) + sqr(l) - 1.f) / (2.f * r * l), -1.f, 1.f)));
}
// detail::boxmueller_truncated_pdf (gaussian.hpp:55-75)
WT_HD float boxmueller_truncated_pdf(vec2 wo, vec2 mean, float sigma2) {
    const float l = sqrtf(fminf_(1.f, dot(mean, mean)));
    const float coso = sqrtf(fmaxf_(0.f, 1.f - dot(mean, mean)));
    wo = wo - mean;
    const float r2 = dot(wo, wo);
    const float x = expf(-.5f * r2 / sigma2);
    const float max_phi = gaussian_max_phi(sqrtf(r2), l);
    return .5f * x / (max_phi * sigma2) * coso;
}

WT_HD float profile_alpha(const material_t& m, vec3 wi, vec3 wo, float k) {
    if (m.profile == PROFILE_DIRAC) return 1.f;
    if (m.profile == PROFILE_GAUSSIAN) return expf(-(sqr((fabsf(wi.z) + fabsf(wo.z)) * k) * gaussian_params(m, k).alpha));
    const fractal_params_t pr = fractal_params(m, k);
    const float a = sqr((fabsf(wi.z) + fabsf(wo.z)) * k) * pr.alpha;
    return expf(-a);
}
WT_HD bool profile_is_delta_only(const material_t& m) {
    if (m.profile == PROFILE_GAUSSIAN && m.gauss_sigma > 0.f) return false;
    // (a textured roughness decides through its MEAN value, which a bitmap / function texture does not have: fractal.hpp:169-177 -> false;
    // constant textures are folded into `roughness` by the host)
    if (m.rough_tex) return m.profile == PROFILE_DIRAC;
    return m.profile == PROFILE_DIRAC || m.roughness == 0.f;
}
WT_HD float profile_psd(const material_t& m, vec3 wi, vec3 wo, float k) {
    if (m.profile == PROFILE_DIRAC) return 0.f;
    if (m.profile == PROFILE_GAUSSIAN) return gaussian_psd(gaussian_params(m, k), k * (vec2{wi.x, wi.y} + vec2{wo.x, wo.y}), k);
    const fractal_params_t pr = fractal_params(m, k);
    const vec2 z = k * (vec2{wi.x, wi.y} + vec2{wo.x, wo.y});
    return fractal_psd(m, pr, z, k);
}
WT_HD float profile_pdf(const material_t& m, vec3 wi, vec3 wo, float k) {
    if (m.profile == PROFILE_DIRAC) return 0.f;
    if (m.profile == PROFILE_GAUSSIAN) return boxmueller_truncated_pdf(vec2{wo.x, wo.y}, -vec2{wi.x, wi.y}, gaussian_params(m, k).sigma2 / (k * k));
    const fractal_params_t pr = fractal_params(m, k);
    const vec2 zeta_k = vec2{wi.x, wi.y} + vec2{wo.x, wo.y};
    const float f_k = length(zeta_k);
    const float s = sqrtf(fmaxf_(0.f, 1.f - sqr(wi.z)));
    const float phi_max = (f_k == 0.f || s == 0.f) ? kPi : acosf(clampf((sqr(f_k) + sqr(s) - 1.f) / (2.f * f_k * s), -1.f, 1.f));
    const float psd = fractal_psd(m, pr, zeta_k * k, k);
    const float w = kInvPi * phi_max;
    return w > 1e-2f ? 1.f / w * fabsf(wo.z) * psd : 0.f;
}
struct profile_sample_t {
    vec3 wo;
    float pdf, psd, weight;
};
// fractal.cpp:27-69 (Holzschuch & Pacanowski importance sampling)
WT_HD profile_sample_t profile_sample(const material_t& m, vec3 wi, float k, sampler_t& sampler) {
    if (m.profile == PROFILE_DIRAC) return {{0, 0, 1}, 0.f, 0.f, 0.f};
    if (m.profile == PROFILE_GAUSSIAN) {
        // gaussian_t::sample (gaussian.hpp:205-228) + detail::sample_boxmueller_truncated (gaussian.hpp:26-53)
        const gaussian_params_t gp = gaussian_params(m, k);
        const float s2 = gp.sigma2 / (k * k);
        const vec2 mean = -vec2{wi.x, wi.y};
        const vec2 u = sampler_r2(sampler);
        const float l = sqrtf(fminf_(1.f, dot(mean, mean)));
        const float coso = sqrtf(fmaxf_(0.f, 1.f - dot(mean, mean)));
        const float phi_i = (mean.x != 0.f || mean.y != 0.f) ? atan2f(mean.y, mean.x) : 0.f;
        const float sm = expf(-.5f * sqr(1.f + l) / s2);
        const float x = (1.f - sm) * fmaxf_(FLT_EPSILON, u.x) + sm;
        const float r = sqrtf(-2.f * s2 * logf(x));
        const float max_phi = gaussian_max_phi(r, l);
        const float phi = phi_i + kPi + max_phi * (2.f * u.y - 1.f);
        const vec2 wo2 = r * vec2{cosf(phi), sinf(phi)} + mean;
        const float pdf = .5f * x / (max_phi * s2) * coso;
        const float psd = gaussian_psd(gp, k * (wo2 - mean), k);
        const float z = sqrtf(fmaxf_(0.f, 1.f - dot(wo2, wo2)));
        return {vec3{wo2.x, wo2.y, wi.z >= 0.f ? z : -z}, pdf, psd, psd / pdf};
    }
    const fractal_params_t pr = fractal_params(m, k);
    const float s = sqrtf(fmaxf_(0.f, 1.f - sqr(wi.z)));
    const float phi_i = s > 0.f ? atan2f(wi.y, wi.x) : 0.f;
    const float sqrtT = sqrtf(pr.T);
    const vec2 u2 = sampler_r2(sampler);
    const float k2T = sqr(k) * pr.T;
    const float g = m.gamma;
    const float M = 1.f - powf(1.f + k2T * sqr(1.f + s), -(g - 1.f) / 2.f);
    const float f = sqrtf(powf(1.f - M * u2.x, -2.f / (g - 1.f)) - 1.f) / sqrtT;   // [1/mm]
    const float f_k = f / k;
    const float phi_max = (f == 0.f || s == 0.f) ? kPi : acosf(clampf((sqr(f_k) + sqr(s) - 1.f) / (2.f * f_k * s), -1.f, 1.f));
    const float phi_f = phi_i + (2.f * u2.y - 1.f) * phi_max;
    const vec2 zeta = f * vec2{cosf(phi_f), sinf(phi_f)};
    const vec2 zeta_k = zeta / k;
    const vec2 wo = zeta_k - vec2{wi.x, wi.y};
    const float z = sqrtf(fmaxf_(0.f, 1.f - dot(wo, wo)));
    const float psd = fractal_psd(m, pr, zeta, k);
    const float w = kInvPi * phi_max;
    const float pdf = w > 1e-2f ? z * psd / w : 0.f;
    return {vec3{wo.x, wo.y, wi.z >= 0.f ? z : -z}, pdf, psd, w};
}

// ---- leaf BSDFs -----------------------------------------------------------------------------------
WT_HD vec3 two_sided_flip(vec3 w, float z) { return z >= 0.f ? w : vec3{w.x, w.y, -w.z}; }

WT_HD cplx material_IOR(const scene_t& sc, const material_t& m, float k) {
    const cplx eta_1 = spectrum_value(sc, m.ext_ior_spec, k);
    const cplx eta_2 = spectrum_value(sc, m.ior_spec, k);
    return eta_1 / eta_2;
}
// surface_spm.cpp:28-38
WT_HD vec3 spm_flip_wo(vec3 wo, float eta) {
    const float scale = wo.z > 0.f ? eta : 1.f / eta;
    const vec2 xy = vec2{wo.x, wo.y} * scale;
    const float l2 = dot(xy, xy);
    return l2 > 1.f ? vec3{1, 0, 0} : vec3{xy.x, xy.y, (wo.z > 0.f ? -1.f : 1.f) * sqrtf(fmaxf_(0.f, 1.f - l2))};
}
WT_HD bool IOR_has_transmission(cplx IOR) { return sqr(fabsf(IOR.im)) / cnorm(IOR) <= 1e-2f; }

constexpr int kMaxCompositeBins = (int)(sizeof(material_t::bin_child) / sizeof(int32_t));
// Follows the dispatching wrappers down to a leaf BSDF at wavenumber k:
//   composite (bsdf/composite.hpp:84-135: the bin whose left-inclusive range [kmin,kmax) holds k; none -> no BSDF: f = 0, no sample,
//   pdf = 0, is_delta_only = true), mask (src/bsdf/mask.cpp: the nested BSDF; `alpha` accumulates the mask opacity).
// two_sided / scale of a wrapper apply to what it wraps.  FALSE: no BSDF at this wavenumber.
struct material_wrap_t {
    float alpha;      // mask opacity (1: none)
    bool masked;      // a mask wrapper was passed
    bool mask_two;    // ... under a two_sided wrapper: the mask sees the flipped directions
};
// the non-constant factor of a scale wrapper (bsdf/scale.hpp:78-97: scale->f(tquery).x): a spectrum and / or a texture
WT_HD float material_scale_factor(const scene_t& sc, const material_t& m, float k, vec2 uv) {
    float f = 1.f;
    if (m.scale_spec) f *= spectrum_f(sc, (int)m.scale_spec - 1, k);
    if (m.scale_tex) f *= texture_spectral(sc, (int)m.scale_tex - 1, uv, k);
    return f;
}
// (LEAF: the caller knows that `mat` is an unwrapped BSDF — the wrappers' code is not compiled in)
template <bool LEAF = false>
WT_HD bool material_resolve(const scene_t& sc, int mat, float k, vec2 uv, material_t& m, material_wrap_t& wr) {
    m = sc.materials[mat];
    wr.alpha = 1.f;
    wr.masked = wr.mask_two = false;
    if (LEAF || m.type < MAT_COMPOSITE) {   // a leaf BSDF: the common case (only the fields the caller goes on to use are loaded)
        if (m.scale_spec | m.scale_tex) m.scale *= material_scale_factor(sc, m, k, uv);
        if (m.rough_tex) m.roughness = texture_spectral(sc, (int)m.rough_tex - 1, uv, k);   // fractal.hpp:83-92: roughness_tex->f(query).x
        return true;
    }
    uint32_t two = 0;
    float scale = 1.f;
    int cur = mat;
    for (int depth = 0; depth < 4 && m.type >= MAT_COMPOSITE; ++depth) {
        two |= m.two_sided;
        scale *= m.scale;
        if (m.scale_spec | m.scale_tex) scale *= material_scale_factor(sc, m, k, uv);
        int child = -1;
        if (m.type == MAT_MASK) {
            wr.alpha *= clamp01(m.mask_tex ? texture_spectral(sc, (int)m.mask_tex - 1, uv, k) : m.mask_alpha);   // mask.cpp:27, 50: clamp01(mask->f(tquery).x)
            wr.masked = true;
            wr.mask_two = two != 0;
            child = m.nested;
        } else {
            // (the bins are read from the scene's record, not from the local copy: a dynamically indexed local array lives in scratch memory on the device)
            const material_t& gm = sc.materials[cur];
            for (uint32_t i = 0; i < m.n_bins && i < (uint32_t)kMaxCompositeBins; ++i)
                if (gm.bin_kmin[i] <= k && k < gm.bin_kmax[i]) {
                    child = gm.bin_child[i];
                    break;
                }
        }
        if (child < 0) return false;
        cur = child;
        m = sc.materials[child];
    }
    if (m.type >= MAT_COMPOSITE) return false;   // nesting deeper than the flattener produces
    m.two_sided |= two;
    m.scale *= scale;
    if (m.scale_spec | m.scale_tex) m.scale *= material_scale_factor(sc, m, k, uv);
    if (m.rough_tex) m.roughness = texture_spectral(sc, (int)m.rough_tex - 1, uv, k);
    return true;
}

WT_HD bool material_is_delta_only(const scene_t& sc, int mat, float k) {
    material_t m;
    material_wrap_t wr;
    if (!material_resolve(sc, mat, k, vec2{0.f, 0.f}, m, wr)) return true;   // (is_delta_only does not depend on the mask's value)
    if (m.type == MAT_DIFFUSE) return false;
    if (m.type == MAT_DIELECTRIC) return true;
    return profile_is_delta_only(m);
}

// bsdf_t::f — includes the cosine foreshortening; only non-delta lobes
// `uv`: the surface's texture coordinates (intersection_surface_t::texture_query), used by textured reflectances and masks
WT_HD mueller_t material_f(const scene_t& sc, int mat, vec3 wi, vec3 wo, float k, uint32_t transport, vec2 uv = vec2{0.f, 0.f}) {
    material_t m;
    material_wrap_t wr;
    if (!material_resolve(sc, mat, k, uv, m, wr) || wr.alpha == 0.f) return mueller_zero();
    if (m.two_sided) {
        const float z = wi.z;
        wi = two_sided_flip(wi, z);
        wo = two_sided_flip(wo, z);
    }
    mueller_t M = mueller_zero();
    if (m.type == MAT_DIFFUSE) {
        const float refl = clamp01(spectrum_f(sc, m.refl_spec, k) * m.refl_tex_scale * (m.refl_tex ? texture_spectral(sc, (int)m.refl_tex - 1, uv, k) : 1.f));
        M = mueller_depolarizer((wi.z > 0.f && wo.z > 0.f) ? wo.z * kInvPi * refl : 0.f);
    } else if (m.type == MAT_SURFACE_SPM) {
        const bool is_scatter = !profile_is_delta_only(m);
        const bool is_reflection = wi.z * wo.z >= 0.f;
        const cplx eta_12 = material_IOR(sc, m, k);
        const bool has_transmission = IOR_has_transmission(eta_12);
        if (!(wi.z == 0.f || wo.z == 0.f || !is_scatter || (!is_reflection && !has_transmission))) {
            const vec3 abs_wo = is_reflection ? wo : spm_flip_wo(wo, eta_12.re);
            const float alpha = profile_alpha(m, wi, abs_wo, k);
            float J = 1.f;
            if (!is_reflection && transport == TRANSPORT_BACKWARD) J = sqr(wi.z < 0.f ? 1.f / eta_12.re : eta_12.re);
            const float scale = is_reflection ? m.refl_scale : m.trans_scale;
            const vec3 h = wi + abs_wo;
            const vec3 mm = normalize(wi.z < 0.f ? -h : h);
            // NB: f() evaluates Fresnel with Re(eta) only (surface_spm.cpp:66), sample() with the complex eta.
            const mueller_t F = mueller_fresnel_rt(cplx{eta_12.re, 0.f}, is_reflection, wi, mm);
            const float psd = profile_psd(m, wi, abs_wo, k);
            M = ((1.f - alpha) * J * fabsf(wo.z) * psd * scale) * F;
        }
    }
    // dielectric: delta only -> 0
    if (m.scale != 1.f) M = M * m.scale;
    if (wr.masked) M = M * wr.alpha;   // mask.cpp:24-35
    return M;
}

// CLS (here and in material_sample): the leaf type of the material when the caller knows it at compile time (MAT_DIFFUSE / MAT_DIELECTRIC /
// MAT_SURFACE_SPM: the class kernels of the device's material-sorted interaction pass, which are handed only walks on unwrapped materials of
// that type) — the other BSDFs' code is not compiled in; -1: any material (wrappers resolved at run time).  Same arithmetic either way.
template <int CLS = -1>
WT_HD float material_pdf_leaf(const scene_t& sc, const material_t& m, vec3 wi, vec3 wo, float k, uint32_t transport);
template <int CLS = -1>
WT_HD float material_pdf(const scene_t& sc, int mat, vec3 wi, vec3 wo, float k, uint32_t transport, vec2 uv = vec2{0.f, 0.f}) {
    material_t m;
    material_wrap_t wr;
    if (!material_resolve<(CLS >= 0)>(sc, mat, k, uv, m, wr)) return 0.f;
    if (CLS < 0 && wr.masked) {
        // mask.cpp:79-92: no transmission through a masked BSDF; the nested density times the probability of not taking the null lobe
        if (wr.mask_two) {
            const float z = wi.z;
            wi = two_sided_flip(wi, z);
            wo = two_sided_flip(wo, z);
        }
        if (wi.z <= 0.f || wo.z <= 0.f) return 0.f;
        return material_pdf_leaf<CLS>(sc, m, wi, wo, k, transport) * wr.alpha;
    }
    return material_pdf_leaf<CLS>(sc, m, wi, wo, k, transport);
}
template <int CLS>
WT_HD float material_pdf_leaf(const scene_t& sc, const material_t& m, vec3 wi, vec3 wo, float k, uint32_t transport) {
    const int type = CLS >= 0 ? CLS : m.type;
    if (type == MAT_DIELECTRIC) return 0.f;
    if (m.two_sided) {
        const float z = wi.z;
        wi = two_sided_flip(wi, z);
        wo = two_sided_flip(wo, z);
    }
    if (type == MAT_DIFFUSE) return (wi.z > 0.f && wo.z > 0.f) ? cosine_hemisphere_pdf(wo.z) : 0.f;
    const bool is_reflection = wi.z * wo.z >= 0.f;
    const cplx eta_12 = material_IOR(sc, m, k);
    const bool has_transmission = IOR_has_transmission(eta_12);
    if (wi.z == 0.f || wo.z == 0.f || (!is_reflection && !has_transmission)) return 0.f;
    const vec3 abs_wo = is_reflection ? wo : spm_flip_wo(wo, eta_12.re);
    const float alpha = profile_alpha(m, wi, wi, k);
    const float pdf_specular = alpha;
    const fresnel_t f = fresnel(cplx{eta_12.re, 0.f}, wi, vec3{0, 0, 1});
    const float pdf_transmission = (f.Ts + f.Tp) / 2.f;
    return (1.f - pdf_specular) * profile_pdf(m, wi, abs_wo, k) * (is_reflection ? 1.f - pdf_transmission : pdf_transmission);
}

template <int CLS = -1>
WT_HD bsdf_sample_t material_sample(const scene_t& sc, int mat, vec3 wi_in, float k, uint32_t transport, sampler_t& sampler, vec2 uv = vec2{0.f, 0.f}) {
    bsdf_sample_t r;
    r.valid = false;
    r.wo = {0, 0, 1};
    r.dpd = 0.f;
    r.eta = 1.f;
    r.M = mueller_zero();
    material_t m;
    material_wrap_t wr;
    if (!material_resolve<(CLS >= 0)>(sc, mat, k, uv, m, wr)) return r;   // composite: no bin at this wavenumber
    // mask.cpp:37-77 (every lobe is admitted by the integrators' queries: has_null = true): the null lobe passes the beam straight
    // through with probability 1 - alpha (always, from behind), the nested BSDF is sampled otherwise
    float not_null = 1.f;
    const int type = CLS >= 0 ? CLS : m.type;
    if (CLS < 0 && wr.masked) {
        const float pdf_null = (wr.mask_two ? wi_in.z == 0.f : wi_in.z <= 0.f) ? 1.f : 1.f - wr.alpha;
        const bool is_null = pdf_null == 0.f ? false : (pdf_null == 1.f ? true : sampler_r(sampler) < pdf_null);
        if (is_null) {
            r.wo = -wi_in;
            r.dpd = pd_discrete(pdf_null);
            r.M = mueller_identity() * ((1.f - wr.alpha) / pdf_null);
            r.valid = true;
            return r;
        }
        not_null = 1.f - pdf_null;
    }
    const float flipz = wi_in.z;
    const vec3 wi = m.two_sided ? two_sided_flip(wi_in, flipz) : wi_in;

    if (type == MAT_DIFFUSE) {
        if (wi.z <= 0.f) return r;
        const float refl = clamp01(spectrum_f(sc, m.refl_spec, k) * m.refl_tex_scale * (m.refl_tex ? texture_spectral(sc, (int)m.refl_tex - 1, uv, k) : 1.f));
        r.wo = cosine_hemisphere(sampler_r2(sampler));
        r.dpd = cosine_hemisphere_pdf(r.wo.z);
        r.M = mueller_depolarizer(refl);
        r.valid = true;
    } else if (type == MAT_DIELECTRIC) {
        // dielectric.cpp:26-72 — IOR() returns Re(eta_1/eta_2)
        const float eta_12 = material_IOR(sc, m, k).re;
        const fresnel_t f = fresnel(cplx{eta_12, 0.f}, wi, vec3{0, 0, 1});
        const float T = (f.Ts + f.Tp) / 2.f;
        const bool is_reflection = sampler_r(sampler) >= T;
        const vec3 wo = is_reflection ? reflect_z(wi) : f.t;
        const float pdf = is_reflection ? 1.f - T : T;
        const float scale = is_reflection ? m.refl_scale : m.trans_scale;
        if (scale == 0.f) return r;
        mueller_t M;
        if (is_reflection)
            M = scale * mueller_fresnel(f.rs, f.rp);
        else {
            M = (f.Z * scale) * mueller_fresnel(f.ts, f.tp);
            if (transport == TRANSPORT_BACKWARD) M = M * sqr(f.eta_12.re);
        }
        r.wo = wo;
        r.dpd = pd_discrete(1.f);
        r.eta = f.eta_12.re;
        r.M = M * (1.f / pdf);
        r.valid = true;
    } else {
        // surface_spm.cpp:75-157
        const float alpha = profile_alpha(m, wi, wi, k);
        const bool has_specular = alpha > 0.f;
        const bool has_scatter = alpha < 1.f;
        const cplx eta_12 = material_IOR(sc, m, k);
        const bool has_transmission = IOR_has_transmission(eta_12);
        if (wi.z == 0.f || (!has_specular && !has_scatter)) return r;
        float pdf = 1.f;
        bool is_specular = has_specular;
        if (has_specular && has_scatter) {
            const float pdf_specular = alpha;
            is_specular = pdf_specular == 1.f || sampler_r(sampler) < pdf_specular;
            pdf = is_specular ? pdf_specular : 1.f - pdf_specular;
        }
        float J = 1.f;
        const fresnel_t f = fresnel(eta_12, wi, vec3{0, 0, 1});
        const float pdf_transmission = (f.Ts + f.Tp) / 2.f;
        bool is_reflection = true;
        if (has_transmission) {
            is_reflection = sampler_r(sampler) >= pdf_transmission;
            pdf *= is_reflection ? 1.f - pdf_transmission : pdf_transmission;
        }
        if (!is_reflection && transport == TRANSPORT_BACKWARD) J = sqr(f.eta_12.re);
        const float scale = is_reflection ? m.refl_scale : m.trans_scale;
        if (scale == 0.f || (!is_reflection && !has_transmission)) return r;
        if (is_specular) {
            const vec3 wo = is_reflection ? reflect_z(wi) : f.t;
            const mueller_t F = mueller_fresnel_rt(eta_12, is_reflection, wi, vec3{0, 0, 1});
            r.wo = wo;
            r.dpd = pd_discrete(pdf);
            r.eta = is_reflection ? 1.f : f.eta_12.re;
            r.M = F * (alpha * J * scale / pdf);
            r.valid = true;
        } else {
            const profile_sample_t ps = profile_sample(m, wi, k, sampler);
            const vec3 h = wi + ps.wo;
            const vec3 mm = normalize(wi.z < 0.f ? -h : h);
            const mueller_t F = mueller_fresnel_rt(eta_12, is_reflection, wi, mm);
            const vec3 wo = is_reflection ? ps.wo : spm_flip_wo(ps.wo, eta_12.re);
            pdf *= ps.pdf;
            r.wo = wo;
            r.dpd = pdf;
            r.eta = is_reflection ? 1.f : f.eta_12.re;
            r.M = F * ((1.f - alpha) * J * fabsf(wo.z) * ps.psd * scale / pdf);
            r.valid = true;
        }
    }
    if (r.valid) {
        if (m.two_sided) r.wo = two_sided_flip(r.wo, flipz);
        if (m.scale != 1.f) r.M = r.M * m.scale;
        if (CLS < 0 && wr.masked) {   // mask.cpp:72-75
            r.dpd *= not_null;   // (a discrete mass is stored negated: scaling keeps the flag)
            r.M = r.M * (wr.alpha / not_null);
        }
    }
    return r;
}

// integrator/common.hpp:21-33
WT_HD float shading_normals_correction_scale(uint32_t transport, float wig, float wog, float wis, float wos) {
    if (transport == TRANSPORT_FORWARD) return fminf_(fabsf(wis * wog / (wos * wig)), 1e+2f);
    return 1.f;
}

}   // namespace wt
