// oracle/indep/prims.cpp — the primitive layer behind oracle/indep/prims.h: thin wrappers over the restated physics in
// wave_tracer_amd/csrc/wt/*.h, one primitive each (a ray-triangle test, a BSDF sample, a Gaussian triangle integral, an aperture
// construction, ...).  Nothing in this file decides what happens to a beam or composes a path: which triangle lies under the beam axis,
// whether an interaction is a surface / free-space-diffraction / null interaction and everything after that is oracle/indep/indep.cpp.
//                                                                                                      *** TEST INFRASTRUCTURE ***
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../wave_tracer_amd/csrc/wt/bdpt.h"
#include "../../wave_tracer_amd/csrc/wt/path.h"
#include "prims.h"

using namespace wt;

namespace {
static_assert(sizeof(beam_t) <= sizeof(prim_beam) && sizeof(surface_t) <= sizeof(prim_surface) && sizeof(sensor_element_t) <= sizeof(prim_element), "blob sizes");
template <class T, class B>
T get(const B* b) {
    T t;
    std::memcpy(&t, b->b, sizeof(T));
    return t;
}
template <class T, class B>
void put(B* b, const T& t) {
    std::memset(b->b, 0, sizeof(b->b));
    std::memcpy(b->b, &t, sizeof(T));
}
vec3 v3(const float* p) { return {p[0], p[1], p[2]}; }
void o3(float* p, vec3 v) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}
const scene_t& S(const void* sc) { return *static_cast<const scene_t*>(sc); }

// per-thread scratch: traversal stack + unbounded triangle list of the last prim_trace, Fraunhofer apertures of the current sample
struct tls_t {
    stack_entry_t stack[4096];
    std::vector<uint32_t> tris = std::vector<uint32_t>(1u << 18);
    std::vector<float> dists = std::vector<float>(1u << 18);
    std::vector<fsd_aperture_t> hdr = std::vector<fsd_aperture_t>(64);
    std::vector<fsd_edge_t> edges = std::vector<fsd_edge_t>(64 * (size_t)kFsdMaxEdges);
    uint32_t n_ap = 0;
    utd_aperture_t utd_ap{};                                                       // plt_path: the current interaction's UTD aperture
    std::vector<utd_edge_rec_t> utd_edges = std::vector<utd_edge_rec_t>(kUtdMaxEdges);
};
thread_local tls_t tls;
stack_ref_t stack() { return make_flat_stack(tls.stack, 4096); }
}   // namespace

extern "C" {

void prim_info(const void* sc_, int out[14]) {
    const scene_t& sc = S(sc_);
    const int v[14] = {sc.opts.max_depth, (int)sc.opts.MIS, (int)sc.opts.RR, (int)sc.opts.FSD, (int)sc.opts.sensor_direct, (int)sc.opts.emitter_direct,
                       (int)sc.sensor.width, (int)sc.sensor.height, (int)sc.sensor.channels, (int)film_stokes(sc.sensor), (int)sc.opts.integrator,
                       (sensor_is_virtual(sc.sensor) ? 1 : 0) | (sensor_is_delta_direction(sc.sensor) ? 2 : 0) | (sensor_is_delta_position(sc.sensor) ? 4 : 0) |
                           ((sc.sensor.ray_trace_only || sc.opts.force_ray_tracing) ? 8 : 0),
                       (int)sc.opts.debug_only_s, (int)sc.opts.debug_only_t};
    std::memcpy(out, v, sizeof(v));
}
void prim_streams(uint32_t out[4]) {
    out[0] = STREAM_SCENE;
    out[1] = STREAM_SENSOR_WALK;
    out[2] = STREAM_EMITTER_WALK;
    out[3] = STREAM_CONNECT;
}
void prim_pool_reset(void) { tls.n_ap = 0; }
void prim_pool_reserve(uint32_t n_apertures) {   // one aperture per free-space-diffraction vertex of a sample: 2 x (max_depth + 2) at most
    if (tls.hdr.size() < n_apertures) {
        tls.hdr.resize(n_apertures);
        tls.edges.resize((size_t)n_apertures * kFsdMaxEdges);
    }
}

void prim_generate(const void* sc_, uint64_t seed, uint64_t sid, uint32_t px, uint32_t py, prim_gen* out) {
    const scene_t& sc = S(sc_);
    sampler_t smp = make_sampler(seed, sid, STREAM_SCENE);
    const emitter_k_sample_t ek = scene_sample_emitter_and_spectrum(sc, smp);
    const float k = ek.wavenumber.k;
    const emitter_sample_t es = emitter_sample(sc, ek.emitter, k, smp);
    const bool disc = pd_is_discrete(ek.wavenumber.wpd);
    out->k = k;
    out->recp_spectral_pd = disc ? 1.f / pd_mass(ek.wavenumber.wpd) : 1.f / scene_sum_spectral_pdf(sc, k);
    out->k_density = disc ? pd_mass(ek.wavenumber.wpd) : ek.wavenumber.wpd;
    const sensor_sample_t ss = sensor_sample(sc, px, py, k, smp);
    put(&out->element, ss.element);
    put(&out->sbeam, ss.beam);
    out->s_dpd = ss.dpd;
    out->s_ppd = ss.ppd;
    out->s_has_surface = ss.has_surface;
    put(&out->s_surface, ss.surface);
    put(&out->ebeam, es.beam);
    out->e_dpd = es.dpd;
    out->e_ppd = es.ppd;
    out->e_select_pdf = ek.emitter_pdf;
    out->emitter = ek.emitter;
    out->e_has_surface = es.has_surface;
    put(&out->e_surface, es.surface);
}

// integrator::traverse (+ the self-intersection offset of the origin, traversal.hpp:276-288)
void prim_trace(const void* sc_, const prim_beam* beam, uint32_t prev_offset_tuid, const float prev_ng[3], prim_trav* out) {
    const scene_t& sc = S(sc_);
    const beam_t b = get<beam_t>(beam);
    cone_t env = b.env;
    if (prev_offset_tuid != kInvalid) {
        const tri_geo_t g = sc.tri_geo[prev_offset_tuid];
        const vec3 ng = v3(prev_ng);
        const vec3 err = triangle_fp_errors(g.a, g.b, g.c, env.o);
        const vec3 offset = dot(err, vabs(ng)) * ng;
        env.o = env.o + (dot(env.d, offset) >= 0.f ? offset : -offset);
    }
    const uint_list_t tris{tls.tris.data(), 1, (uint32_t)tls.tris.size(), tls.dists.data()};
    const bool rt = sc.sensor.ray_trace_only || sc.opts.force_ray_tracing;
    const trav_result_t tr = traverse(sc, env, wavenum_to_wavelen_m(b.k), WT_INF, rt, stack(), tris);
    out->empty = tr.empty;
    out->ballistic = tr.ballistic;
    out->dist = tr.dist;
    out->region_depth = tr.region_depth;
    out->front_face = tr.front_face;
    out->tuid = tr.tuid;
    out->bx = tr.bx;
    out->by = tr.by;
    out->ntris = tr.ntris;
    o3(out->origin, tr.origin);
}

// ---- the pieces of one interaction (composed in indep.cpp) ----------------------------------------------------------------------------
uint32_t prim_trav_tri(uint32_t i) { return tls.tris[i]; }
int prim_beam_is_ray(const prim_beam* b) { return beam_is_ray(get<beam_t>(b)) ? 1 : 0; }
int prim_axis_hits_tri(const void* sc_, uint32_t tuid, const float o[3], const float d[3], float zmin, float zmax, float* dist, float bary[2]) {
    const scene_t& sc = S(sc_);
    const tri_geo_t g = sc.tri_geo[tuid];
    const vec3 origin = v3(o);
    ray_tri_hit_t h;
    if (!intersect_ray_tri(origin, v3(d), g.a, g.b, g.c, grow(range_t{zmin, zmax}, cone_intersection_tolerance(origin, g.a, g.b, g.c)), h)) return 0;
    *dist = h.dist;
    bary[0] = h.bx;
    bary[1] = h.by;
    return 1;
}
void prim_tri_edges(const void* sc_, uint32_t tuid, uint32_t e[3]) {
    const tri_meta_t m = S(sc_).tri_meta[tuid];
    e[0] = m.edge[0];
    e[1] = m.edge[1];
    e[2] = m.edge[2];
}
void prim_surface_at(const void* sc_, const prim_beam* beam_, uint32_t tuid, const float bary[2], const float wp[3], float beam_dist, prim_surface* out, int* material,
                     int* emitter_of_shape) {
    const scene_t& sc = S(sc_);
    const beam_t beam = get<beam_t>(beam_);
    surface_t srf = make_surface(sc, tuid, sc.tri_geo[tuid].n, vec2{bary[0], bary[1]}, v3(wp));
    srf.footprint = beam_surface_footprint_static(beam, srf, beam_dist);
    put(out, srf);
    const shape_t shp = sc.shapes[srf.shape];
    *material = shp.material;
    *emitter_of_shape = shp.emitter;
}
void prim_surface_to_world(const prim_surface* s, const float v[3], float out[3]) { o3(out, to_world(get<surface_t>(s).shading, v3(v))); }
void prim_material_sample(const void* sc_, int mat, const prim_surface* at, const float wi[3], float k, int transport, uint64_t seed, uint64_t sid, uint32_t stream,
                          uint32_t* draws, prim_bsdf_sample* out) {
    sampler_t smp = make_sampler(seed, sid, stream, *draws);
    const bsdf_sample_t bs = material_sample(S(sc_), mat, v3(wi), k, (uint32_t)transport, smp, get<surface_t>(at).uv);
    *draws = smp.draws;
    out->valid = bs.valid ? 1 : 0;
    o3(out->wo, bs.wo);
    out->dpd = bs.dpd;
    std::memcpy(out->M, bs.M.m, sizeof(out->M));
    out->eta = bs.eta;
}
float prim_region_tri_flux(const void* sc_, const prim_beam* beam_, float beam_dist, float region_depth, uint32_t tuid, int want_front) {
    const beam_t beam = get<beam_t>(beam_);
    const vec3 sd = beam_footprint(beam, beam_dist) / kBeamEnvelope;
    return region_triangle_flux(S(sc_), cone_frame(beam.env), beam.env, range_t{beam_dist, beam_dist + region_depth}, vec2{sd.x, sd.y}, tuid, want_front != 0);
}
int prim_fsd_build(const void* sc_, const prim_beam* beam_, float beam_dist, const uint32_t* edge_ids, uint32_t n, float aperture_power) {
    const scene_t& sc = S(sc_);
    const beam_t beam = get<beam_t>(beam_);
    if (tls.n_ap >= tls.hdr.size()) return -1;
    const uint32_t slot = tls.n_ap;
    const vec3 sd = beam_footprint(beam, beam_dist) / kBeamEnvelope;
    fsd_aperture_t ap;
    ap.edge_offset = slot * kFsdMaxEdges;
    ap.edge_cap = kFsdMaxEdges;
    const fsd_edges_ref_t ed{tls.edges.data() + ap.edge_offset, 1};
    fsd_build_aperture(sc, cone_frame(beam.env), beam.k, aperture_power, beam.env, edge_ids, n, vec2{sd.x, sd.y}, ap, ed);
    if (ap.n_edges == 0) return -2;   // (the slot stays free)
    tls.hdr[slot] = ap;
    ++tls.n_ap;
    return (int)slot;
}
void prim_fsd_sample(const void* sc_, int slot, uint64_t seed, uint64_t sid, uint32_t stream, uint32_t* draws, prim_fsd_sampled* out) {
    sampler_t smp = make_sampler(seed, sid, stream, *draws);
    const fsd_aperture_t& ap = tls.hdr[(uint32_t)slot];
    const fsd_sample_t fs = fsd_sample(S(sc_), ap, fsd_edges_ref_t{tls.edges.data() + ap.edge_offset, 1}, smp);
    *draws = smp.draws;
    o3(out->wo_world, to_world(ap.frame, fs.wo));
    out->dpd = fs.dpd;
    out->weight = fs.weight;
}
// ---- plt_path primitives ------------------------------------------------------------------------------------------------------------
void prim_path_generate(const void* sc_, uint64_t seed, uint64_t sid, uint32_t px, uint32_t py, prim_path_gen* out) {
    const scene_t& sc = S(sc_);
    sampler_t smp = make_sampler(seed, sid, STREAM_SCENE);
    const emitter_k_sample_t ek = scene_sample_emitter_and_spectrum(sc, smp);
    const float k = ek.wavenumber.k;
    out->k = k;
    sensor_element_t el{0, 0, {0.f, 0.f}};
    if (sc.opts.integrator == INTEGRATOR_PATH_FORWARD) {
        const emitter_sample_t es = emitter_sample(sc, ek.emitter, k, smp);
        out->recp_spectral_pd = 1.f / scene_sum_spectral_pdf(sc, k);
        put(&out->beam, es.beam);
    } else {
        const bool disc = pd_is_discrete(ek.wavenumber.wpd);
        out->recp_spectral_pd = disc ? 1.f / pd_mass(ek.wavenumber.wpd) : 1.f / scene_sum_spectral_pdf(sc, k);
        const sensor_sample_t ss = sensor_sample(sc, px, py, k, smp);
        el = ss.element;
        put(&out->beam, ss.beam);
    }
    put(&out->element, el);
}
static path_geo_t geo_of(const prim_geo* g) {
    return path_geo_t{v3(g->wp), (uint32_t)g->kind, v3(g->ng), g->id};
}
int prim_shadow_geo(const void* sc, const prim_geo* a, const prim_geo* b) { return path_shadow(S(sc), geo_of(a), geo_of(b), stack(), nullptr) ? 1 : 0; }
uint32_t prim_utd_build(const void* sc_, const prim_beam* beam_, const float interaction_wp[3], float dist, const uint32_t* edge_ids, uint32_t n) {
    const beam_t beam = get<beam_t>(beam_);
    tls.utd_ap.edge_offset = 0;
    tls.utd_ap.edge_cap = kUtdMaxEdges;
    utd_build_aperture(S(sc_), v3(interaction_wp), cone_frame(beam.env), beam_footprint(beam, dist), -beam.env.d, beam.k, edge_ids, n, tls.utd_ap,
                       utd_edges_ref_t{tls.utd_edges.data(), 1});
    return tls.utd_ap.n_edges;
}
void prim_utd_f_edge(const void* sc_, uint32_t i, const float src[3], const float dst[3], prim_utd_term* out) {
    utd_diffracting_edge_t f;
    out->valid = utd_f_edge(S(sc_), tls.utd_ap, tls.utd_edges[i], v3(src), v3(dst), f) ? 1 : 0;
    if (!out->valid) return;
    out->edge = f.edge;
    o3(out->p, f.p);
    out->ro = f.ro;
    out->ri = f.ri;
    out->Ds[0] = f.utd.Ds.re;
    out->Ds[1] = f.utd.Ds.im;
    out->Dh[0] = f.utd.Dh.re;
    out->Dh[1] = f.utd.Dh.im;
}
void prim_utd_sample(const void* sc_, const float prev_wp[3], uint64_t seed, uint64_t sid, uint32_t stream, uint32_t* draws, float wo[3], float* weight) {
    sampler_t smp = make_sampler(seed, sid, stream, *draws);
    const utd_sample_t us = utd_sample(S(sc_), tls.utd_ap, utd_edges_ref_t{tls.utd_edges.data(), 1}, v3(prev_wp), smp);
    *draws = smp.draws;
    o3(wo, us.wo);
    *weight = us.weight;
}
int prim_cone_contains(const prim_beam* b, const float p[3]) { return cone_contains(get<beam_t>(b).env, v3(p)) ? 1 : 0; }
uint32_t prim_ballistic_region(const void* sc_, const prim_beam* beam_, float dist) {
    const beam_t beam = get<beam_t>(beam_);
    const float zdist = cone_axes(beam.env, dist).x * kMajorAxisToZScale;
    const uint_list_t tris{tls.tris.data(), 1, (uint32_t)tls.tris.size(), tls.dists.data()};
    cone_hit_t ch;
    bvh_traverse_cone(S(sc_), beam.env, range_t{dist - zdist / 2.f, dist + zdist / 2.f}, 1.f, stack(), tris, ch);
    return ch.ntris;
}
void prim_beam_add(prim_beam* b_, const prim_beam* o) {
    beam_t b = get<beam_t>(b_);
    beam_add(b, get<beam_t>(o));
    put(b_, b);
}
float prim_k_times_length(float k, float d) { return k_times_len(k, d); }
float prim_beam_axis_x(const prim_beam* b_, float dist, float footprint[3]) {
    const beam_t b = get<beam_t>(b_);
    o3(footprint, beam_footprint(b, dist));
    return cone_axes(b.env, dist).x;
}
void prim_beam_transform_restart(prim_beam* b_, const float wp[3], float dist) {
    beam_t b = get<beam_t>(b_);
    beam_transform_restart(b, v3(wp), dist);
    put(b_, b);
}
float prim_uniform(uint64_t seed, uint64_t sid, uint32_t stream, uint32_t* draws) {
    sampler_t smp = make_sampler(seed, sid, stream, *draws);
    const float u = sampler_r(smp);
    *draws = smp.draws;
    return u;
}

void prim_beam_info(const prim_beam* b_, float o[3], float d[3], float* k, int* transport, float* intensity) {
    const beam_t b = get<beam_t>(b_);
    o3(o, b.env.o);
    o3(d, b.env.d);
    *k = b.k;
    *transport = (int)b.transport;
    *intensity = beam_intensity(b);
}
void prim_beam_scale(prim_beam* b_, float f) {
    beam_t b = get<beam_t>(b_);
    beam_scale(b, f);
    put(b_, b);
}
void prim_beam_payload(const prim_beam* b_, float rad[16], float frame[9], float* scale) {
    const beam_t b = get<beam_t>(b_);
    std::memcpy(rad, b.rad, sizeof(b.rad));
    o3(frame, b.frame.t);
    o3(frame + 3, b.frame.b);
    o3(frame + 6, b.frame.n);
    *scale = b.scale;
}
void prim_beam_transform_surface(prim_beam* b_, const prim_surface* s, const float wo[3], const float M_[16], float weight) {
    beam_t b = get<beam_t>(b_);
    mueller_t M;
    std::memcpy(M.m, M_, sizeof(M.m));
    beam_transform_surface_interaction(b, get<surface_t>(s), v3(wo), M, weight);
    put(b_, b);
}
void prim_beam_transform_region(prim_beam* b_, const float wp[3], float dist, const float wo[3], float weight) {
    beam_t b = get<beam_t>(b_);
    beam_transform_region_interaction(b, v3(wp), dist, v3(wo), weight);
    put(b_, b);
}
void prim_surface_info(const prim_surface* s_, float wp[3], float ng[3], float ns[3], uint32_t* tuid, uint32_t* shape) {
    const surface_t s = get<surface_t>(s_);
    o3(wp, s.wp);
    o3(ng, s.geo.n);
    o3(ns, s.shading.n);
    *tuid = s.tuid;
    *shape = s.shape;
}
void prim_surface_to_local(const prim_surface* s_, const float v[3], float out[3]) { o3(out, to_local(get<surface_t>(s_).shading, v3(v))); }
void prim_dummy_surface(const float n[3], const float p[3], prim_surface* out) { put(out, make_dummy_surface(v3(n), v3(p))); }
void prim_material_f(const void* sc, int mat, const prim_surface* at, const float wi[3], const float wo[3], float k, int transport, float M[16]) {
    const mueller_t m = material_f(S(sc), mat, v3(wi), v3(wo), k, (uint32_t)transport, get<surface_t>(at).uv);
    std::memcpy(M, m.m, sizeof(m.m));
}
float prim_material_pdf(const void* sc, int mat, const prim_surface* at, const float wi[3], const float wo[3], float k, int transport) {
    return material_pdf(S(sc), mat, v3(wi), v3(wo), k, (uint32_t)transport, get<surface_t>(at).uv);
}
int prim_material_is_delta_only(const void* sc, int mat, float k) { return material_is_delta_only(S(sc), mat, k) ? 1 : 0; }
float prim_fsd_pdf(int slot, const float wo_world[3]) {
    const fsd_aperture_t ap = tls.hdr[(size_t)slot];
    return fsd_pdf(ap, fsd_edges_ref_t{tls.edges.data() + ap.edge_offset, 1}, to_local(ap.frame, v3(wo_world)));
}
int prim_emitter_flags(const void* sc, int ei) {
    const emitter_t& e = S(sc).emitters[ei];
    return (emitter_is_area(e) ? 1 : 0) | (emitter_is_delta_direction(e) ? 2 : 0) | (emitter_is_delta_position(e) ? 4 : 0) | (emitter_is_infinite(e) ? 8 : 0);
}
float prim_emitter_select_pmf(const void* sc, int ei) { return S(sc).emitters[ei].select_pmf; }
float prim_emitter_pdf_position(const void* sc, int ei, const prim_surface* s) {
    surface_t srf;
    if (s) srf = get<surface_t>(s);
    return emitter_pdf_position(S(sc), ei, s ? &srf : nullptr);
}
float prim_emitter_pdf_direction(const void* sc, int ei, const float d[3], const prim_surface* s) {
    surface_t srf;
    if (s) srf = get<surface_t>(s);
    return emitter_pdf_direction(S(sc), ei, v3(d), s ? &srf : nullptr);
}
float prim_directional_pdf_target_position(const void* sc, int ei, const float wp[3]) { return directional_pdf_target_position(S(sc).emitters[ei], v3(wp)); }
void prim_emitter_Li(const void* sc, int ei, const prim_beam* b, const prim_surface* s, float L[4]) {
    const stokes_t r = emitter_Li(S(sc), ei, get<beam_t>(b), get<surface_t>(s));
    std::memcpy(L, r.s, sizeof(r.s));
}
float prim_sensor_pdf_position(const void* sc) { return sensor_pdf_position(S(sc)); }
float prim_sensor_pdf_direction(const void* sc, const float d[3]) { return sensor_pdf_direction(S(sc), v3(d)); }
void prim_sample_emitter_direct(const void* sc, const float wp[3], float k, uint64_t seed, uint64_t sid, uint32_t stream, uint32_t* draws, prim_edirect* out) {
    sampler_t smp = make_sampler(seed, sid, stream, *draws);
    const emitter_direct_sample_t ed = scene_sample_emitter_direct(S(sc), v3(wp), k, smp);
    *draws = smp.draws;
    put(&out->beam, ed.beam);
    out->dpd = ed.dpd;
    out->emitter = ed.emitter;
    out->has_surface = ed.has_surface;
    put(&out->surface, ed.surface);
}
void prim_sensor_sample_direct(const void* sc, const float wp[3], float k, uint64_t seed, uint64_t sid, uint32_t stream, uint32_t* draws, prim_sdirect* out) {
    sampler_t smp = make_sampler(seed, sid, stream, *draws);
    const sensor_direct_sample_t sd = sensor_sample_direct(S(sc), v3(wp), k, smp);
    *draws = smp.draws;
    put(&out->beam, sd.beam);
    out->dpd = sd.dpd;
    put(&out->element, sd.element);
    out->has_surface = sd.has_surface;
    put(&out->surface, sd.surface);
}
void prim_vplane_Si(const void* sc, const prim_beam* b, float dist, prim_si* out) {
    const sensor_direct_connection_t dc = vplane_Si(S(sc), get<beam_t>(b), range_t{0.f, dist});
    out->valid = dc.valid ? 1 : 0;
    if (dc.valid) {
        put(&out->beam, dc.beam);
        put(&out->element, dc.element);
        put(&out->surface, dc.surface);
    }
}
void prim_offset_origin(const void* sc, const prim_surface* s, const float ro[3], const float rd[3], float out[3]) {
    o3(out, s ? surface_offseted_ray_origin(S(sc), get<surface_t>(s), v3(ro), v3(rd)) : v3(ro));
}
int prim_shadow_ray(const void* sc, const float o[3], const float d[3], float dist) { return ads_shadow_ray(S(sc), v3(o), v3(d), range_t{0.f, dist}, stack()) ? 1 : 0; }
void prim_film_splat(const void* sc_, double* value, double* weight, double* light, const prim_element* el, const float L[4], float k, int direct) {
    const scene_t& sc = S(sc_);
    const film_t film{value, weight, light, sc.sensor.width, sc.sensor.height, sc.sensor.channels};
    stokes_t s;
    std::memcpy(s.s, L, sizeof(s.s));
    if (direct)
        film_splat_direct(sc, film, get<sensor_element_t>(el), s, k);
    else
        film_splat(sc, film, get<sensor_element_t>(el), s, k);
}

}   // extern "C"
