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
typedef struct {
    int kind;   /* 0: the walk ends here, 1 surface, 2 free-space diffraction, 3 null, 4 restart behind an EMPTY aperture (not a 'null interaction' in the statistics) */
    prim_surface surface; int material, emitter_of_shape, is_delta;
    float dpd, pdf_revr;      /* sampled / reverse solid-angle densities, tagged (negative = discrete mass) */
    float throughput_mult;    /* factor on the walk's throughput (Russian roulette input) */
    int fsd_slot; float wp[3];
    float apply_M[16], apply_w, apply_wo[3], apply_dist;   /* for prim_step_apply (the beam transform happens AFTER the vertex is appended) */
} prim_step;
typedef struct { prim_beam beam; float dpd; int emitter; int has_surface; prim_surface surface; } prim_edirect;
typedef struct { prim_beam beam; float dpd; prim_element element; int has_surface; prim_surface surface; } prim_sdirect;
typedef struct { int valid; prim_beam beam; prim_element element; prim_surface surface; } prim_si;

void prim_info(const void* sc, int out[14]);   /* max_depth MIS RR FSD sensor_direct emitter_direct width height channels stokes integrator sensor_flags only_s+1 only_t+1 */
void prim_streams(uint32_t out[4]);            /* scene, sensor walk, emitter walk, connect base (+ t*32 + s) */
void prim_pool_reset(void);
void prim_generate(const void* sc, uint64_t seed, uint64_t sid, uint32_t px, uint32_t py, prim_gen* out);
void prim_trace(const void* sc, const prim_beam* beam, uint32_t prev_offset_tuid, const float prev_ng[3], prim_trav* out);
void prim_step_sample(const void* sc, const prim_beam* beam, const prim_trav* tr, uint64_t seed, uint64_t sid, uint32_t stream, uint32_t* draws, prim_step* out);
void prim_step_apply(prim_beam* beam, const prim_step* st);
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
float prim_emitter_pdf_position(const void* sc, int ei);
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
