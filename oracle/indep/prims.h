/* prims.h — C view of the restated physics PRIMITIVES (BSDFs, emitters, sensors, beams, traversal, film) for the independent composition
 * checker oracle/indep/indep.cpp.                                                             *** TEST INFRASTRUCTURE ***
 * indep.cpp sees ONLY this header: beams / surface records are opaque blobs it hands back to the primitives; everything that composes
 * them into an estimate (vertices, area-measure densities, Russian roulette, the (s,t) strategies, MIS, beam integration) is written
 * there a second time, from the reference, in double precision.  prims.cpp implements these entry points on top of wt/ headers. */
#ifndef WT_INDEP_PRIMS_H
#define WT_INDEP_PRIMS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { unsigned char b[192]; } prim_beam;
typedef struct { unsigned char b[160]; } prim_surface;
typedef struct { unsigned char b[16]; } prim_element;

typedef struct {
    float k, recp_spectral_pd, k_density;
    prim_element element;
    prim_beam sbeam; float s_dpd, s_ppd; int s_has_surface; prim_surface s_surface;   /* sensor sample (sensor_sample_t) */
    prim_beam ebeam; float e_dpd, e_ppd, e_select_pdf; int emitter; int e_has_surface; prim_surface e_surface;   /* emitter sample */
} prim_gen;
typedef struct { int empty, ballistic; float dist, region_depth; int front_face; uint32_t tuid; float bx, by; uint32_t ntris; float origin[3]; } prim_trav;
typedef struct { int valid; float wo[3]; float dpd; float M[16]; float eta; } prim_bsdf_sample;   /* bsdf_t::sample: local wo, tagged density, weighted Mueller matrix */
typedef struct { float wo_world[3]; float dpd, weight; } prim_fsd_sampled;
typedef struct { prim_beam beam; float dpd; int emitter; int has_surface; prim_surface surface; } prim_edirect;
typedef struct { prim_beam beam; float dpd; prim_element element; int has_surface; prim_surface surface; } prim_sdirect;
typedef struct { int valid; prim_beam beam; prim_element element; prim_surface surface; } prim_si;

void prim_info(const void* sc, int out[14]);   /* max_depth MIS RR FSD sensor_direct emitter_direct width height channels stokes integrator sensor_flags only_s+1 only_t+1 */
void prim_streams(uint32_t out[4]);            /* scene, sensor walk, emitter walk, connect base (+ t*32 + s) */
void prim_pool_reset(void);
void prim_pool_reserve(uint32_t n_apertures);
void prim_generate(const void* sc, uint64_t seed, uint64_t sid, uint32_t px, uint32_t py, prim_gen* out);
void prim_trace(const void* sc, const prim_beam* beam, uint32_t prev_offset_tuid, const float prev_ng[3], prim_trav* out);
/* --- the pieces of one interaction; what kind of interaction happens, which triangle lies under the beam axis, the order of the checks and
 * what is handed to the vertex bookkeeping is decided in indep.cpp (plt_bdpt_detail.hpp:192-526) --- */
uint32_t prim_trav_tri(uint32_t i);   /* i-th triangle of the interaction record of this thread's last prim_trace */
int prim_beam_is_ray(const prim_beam* b);
/* intersect_ray_tri against triangle `tuid` within [zmin, zmax] grown by the triangle's cone_intersection_tolerance (find_closest_triangle's test) */
int prim_axis_hits_tri(const void* sc, uint32_t tuid, const float o[3], const float d[3], float zmin, float zmax, float* dist, float bary[2]);
void prim_tri_edges(const void* sc, uint32_t tuid, uint32_t e[3]);   /* classified edges of a triangle (0xFFFFFFFF: none) */
/* intersection_surface_t of triangle `tuid` at `bary` / `wp` with the beam's static footprint at beam_dist; the shape's bsdf and emitter (-1) */
void prim_surface_at(const void* sc, const prim_beam* beam, uint32_t tuid, const float bary[2], const float wp[3], float beam_dist, prim_surface* out, int* material,
                     int* emitter_of_shape);
void prim_surface_to_world(const prim_surface* s, const float v[3], float out[3]);
void prim_material_sample(const void* sc, int mat, const prim_surface* at, const float wi[3], float k, int transport, uint64_t seed, uint64_t sid, uint32_t stream,
                          uint32_t* draws, prim_bsdf_sample* out);
/* the beam power one triangle of the region intercepts (wavefront_t::integrate_triangle over the clipped, projected triangle); 0 when it faces the other way */
float prim_region_tri_flux(const void* sc, const prim_beam* beam, float beam_dist, float region_depth, uint32_t tuid, int want_front);
/* free_space_diffraction_t over the given classified edges: >= 0 the aperture's slot, -1 no slot left, -2 the aperture is empty */
int prim_fsd_build(const void* sc, const prim_beam* beam, float beam_dist, const uint32_t* edge_ids, uint32_t n, float aperture_power);
void prim_fsd_sample(const void* sc, int slot, uint64_t seed, uint64_t sid, uint32_t stream, uint32_t* draws, prim_fsd_sampled* out);
void prim_beam_transform_restart(prim_beam* b, const float wp[3], float dist);

/* --- plt_path (oracle/indep/indep.cpp: indep_render_path) --- */
typedef struct { float k, recp_spectral_pd; prim_beam beam; prim_element element; } prim_path_gen;
/* integrate_forward / integrate_backward up to the first random_walk (plt_path_detail.hpp:772-828): spectral + emitter (forward) or sensor (backward) sample */
void prim_path_generate(const void* sc, uint64_t seed, uint64_t sid, uint32_t px, uint32_t py, prim_path_gen* out);
/* vertex_geo_variant_t as shadow() / offseted_ray_origin() see it: 0 a point, 1 a surface (triangle id: the self-intersection offset), 2 a classified edge */
typedef struct { int kind; float wp[3]; float ng[3]; uint32_t id; } prim_geo;
int prim_shadow_geo(const void* sc, const prim_geo* a, const prim_geo* b);   /* integrator::shadow (traversal.hpp:319-333): 1 = occluded */
/* UTD aperture of the current interaction (free_space_diffraction_t, UTD form: free_space_diffraction.cpp:23-79); returns its wedge count */
uint32_t prim_utd_build(const void* sc, const prim_beam* beam, const float interaction_wp[3], float dist, const uint32_t* edge_ids, uint32_t n);
typedef struct { int valid; uint32_t edge; float p[3]; float ro, ri; float Ds[2], Dh[2]; } prim_utd_term;
void prim_utd_f_edge(const void* sc, uint32_t i, const float src[3], const float dst[3], prim_utd_term* out);   /* one wedge's term of free_space_diffraction_t::f */
void prim_utd_sample(const void* sc, const float prev_wp[3], uint64_t seed, uint64_t sid, uint32_t stream, uint32_t* draws, float wo[3], float* weight);
int prim_cone_contains(const prim_beam* b, const float p[3]);   /* elliptic_cone_t::contains of the beam's envelope */
/* the cone query of a ballistic hit's surroundings (plt_path_detail.hpp:645-650): triangles in this thread's record (prim_trav_tri), their count */
uint32_t prim_ballistic_region(const void* sc, const prim_beam* beam, float dist);
void prim_beam_add(prim_beam* b, const prim_beam* o);   /* beam_t::operator+= */
float prim_k_times_length(float k, float d);              /* the dimensionless product k d in the library's units */
float prim_beam_axis_x(const prim_beam* b, float dist, float footprint[3]);   /* envelope.axes(dist).x; the beam's footprint at dist */
float prim_uniform(uint64_t seed, uint64_t sid, uint32_t stream, uint32_t* draws);
void prim_beam_info(const prim_beam* b, float o[3], float d[3], float* k, int* transport, float* intensity);
void prim_beam_scale(prim_beam* b, float f);
void prim_beam_payload(const prim_beam* b, float rad[16], float frame[9], float* scale);
void prim_beam_transform_surface(prim_beam* b, const prim_surface* s, const float wo[3], const float M[16], float weight);
void prim_beam_transform_region(prim_beam* b, const float wp[3], float dist, const float wo[3], float weight);
void prim_surface_info(const prim_surface* s, float wp[3], float ng[3], float ns[3], uint32_t* tuid, uint32_t* shape);
void prim_surface_to_local(const prim_surface* s, const float v[3], float out[3]);
void prim_dummy_surface(const float n[3], const float p[3], prim_surface* out);
void prim_material_f(const void* sc, int mat, const prim_surface* at, const float wi[3], const float wo[3], float k, int transport, float M[16]);   /* `at`: texture coordinates of the query */
float prim_material_pdf(const void* sc, int mat, const prim_surface* at, const float wi[3], const float wo[3], float k, int transport);
int prim_material_is_delta_only(const void* sc, int mat, float k);
float prim_fsd_pdf(int slot, const float wo_world[3]);
int prim_emitter_flags(const void* sc, int ei);   /* 1 area, 2 delta direction, 4 delta position, 8 infinite */
float prim_emitter_select_pmf(const void* sc, int ei);
float prim_emitter_pdf_position(const void* sc, int ei, const prim_surface* s);   /* s: the point's surface (textured area emitters read their tables at it), or NULL */
float prim_emitter_pdf_direction(const void* sc, int ei, const float d[3], const prim_surface* s);
float prim_directional_pdf_target_position(const void* sc, int ei, const float wp[3]);
void prim_emitter_Li(const void* sc, int ei, const prim_beam* b, const prim_surface* s, float L[4]);
float prim_sensor_pdf_position(const void* sc);
float prim_sensor_pdf_direction(const void* sc, const float d[3]);
void prim_sample_emitter_direct(const void* sc, const float wp[3], float k, uint64_t seed, uint64_t sid, uint32_t stream, uint32_t* draws, prim_edirect* out);
void prim_sensor_sample_direct(const void* sc, const float wp[3], float k, uint64_t seed, uint64_t sid, uint32_t stream, uint32_t* draws, prim_sdirect* out);
void prim_vplane_Si(const void* sc, const prim_beam* b, float dist, prim_si* out);
void prim_offset_origin(const void* sc, const prim_surface* s, const float ro[3], const float rd[3], float out[3]);
int prim_shadow_ray(const void* sc, const float o[3], const float d[3], float dist);
void prim_film_splat(const void* sc, double* value, double* weight, double* light, const prim_element* el, const float L[4], float k, int direct);
#ifdef __cplusplus
}
#endif
#endif
