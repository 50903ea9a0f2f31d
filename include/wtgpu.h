/* wtgpu.h — C-ABI of the MI355X-native wave_tracer hot path (libwtgpu.so).
 *
 * Drop-in seam (seam S1): this library replaces the block fan-out of the reference's render loop,
 *     scene_renderer_t::render()                    src/scene/render.cpp:381-579
 *       -> block_renderer_t -> integrator_t::integrate(ctx, block, pixel, spp)     src/scene/render.cpp:99-113,
 *                                                                                  include/wt/integrator/integrator.hpp:44-57
 * i.e. everything between "scene + ADS are built" and "film storage is developed".  The host keeps scene parsing,
 * film development/tonemapping and image I/O; it hands over a flattened scene description and gets back the
 * three linear film accumulators  value = sum(w*v), weight = sum(w), light = sum(light-image splats), exactly the
 * quantities film_storage_t holds (include/wt/sensor/film/film_storage.hpp:196-252), from which
 *     pixel = value/weight + light/spe        (film_storage.hpp:256-287, src/scene/render.cpp:245-291).
 *
 * The integrator is part of the flattened scene: plt_bdpt (src/integrator/plt_bdpt.cpp:43-148) or plt_path in either transport
 * direction (src/integrator/plt_path.cpp:39-50); polarimetric sensors get four Stokes planes per channel.
 *
 * Plain C types only; opaque handles; int status returns (0 = ok); no exceptions cross the boundary.
 * One wtgpu_scene may be uploaded to one device per process (one process per GPU).
 */
#ifndef WTGPU_H
#define WTGPU_H

#include <stddef.h>
#include <stdint.h>

#include "wtgpu_scene.h" /* wtgpu_scene_desc: the flattened scene (plain C structs) */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wtgpu_scene wtgpu_scene;
typedef struct wtgpu_comm wtgpu_comm;

enum { WTGPU_OK = 0, WTGPU_ERR_INVALID = 1, WTGPU_ERR_NO_DEVICE = 2, WTGPU_ERR_HIP = 3, WTGPU_ERR_OOM = 4, WTGPU_ERR_OVERFLOW = 5, WTGPU_CANCELLED = 6,
       WTGPU_ERR_COMM = 7 };

/* Parameters of a bundled scene; negative/zero fields select the scene file's defaults.
 * Mirrors the CLI `-D res=..` defines + integrator attributes the reference reads
 * (src/main.cpp:805-928; src/integrator/plt_bdpt.cpp:169-174). */
typedef struct wtgpu_scene_params {
    uint32_t res;
    int32_t max_depth;
    int32_t fsd;
    int32_t mis;
    int32_t rr;
    int32_t force_ray_tracing; /* --ray-tracing (include/wt/wt_context.hpp:43) */
    int32_t mesh_detail;       /* 0: low-poly stand-ins, 1: full tessellation */
    uint32_t lut_n_theta, lut_m; /* resolution of the regenerated Fraunhofer iCDF LUT (0: default) */
    int32_t polarimetric;      /* >0: polarimetric sensor — the film stores the 4 Stokes components per channel
                                * (sensor `polarimetric` attribute, include/wt/sensor/sensor/perspective.hpp:185-330) */
} wtgpu_scene_params;

typedef struct wtgpu_scene_info {
    uint32_t width, height, channels;
    uint32_t n_tris, n_edges, n_nodes, n_leaves, n_shapes, n_emitters, n_materials;
    int32_t max_depth;
    uint32_t sensor_type; /* 0 perspective, 1 virtual_plane */
    double fsd_lut_power[2]; /* integrals of the regenerated LUT densities (compare: PA1, PA2 of fsd.hpp:59-61) */
    uint64_t bytes_per_sample_state; /* device bytes of per-sample path/vertex state */
    uint32_t stokes;      /* film components per channel: 1 (intensity) or 4 (polarimetric sensor: I, Q, U, V) */
    uint32_t integrator;  /* 0 plt_bdpt, 1 plt_path forward, 2 plt_path backward */
} wtgpu_scene_info;

/* Device counters (the reference's stat collectors: include/wt/integrator/stats.hpp:27-83, include/wt/ads/ads_stats.hpp),
 * the inputs of the algorithmic-bytes formula of SURVEY.md §8(d). */
typedef struct wtgpu_counters {
    uint64_t samples;
    uint64_t segments, ray_queries, cone_queries, vertices, connections, shadow_rays;
    uint64_t cone_tri_overflow, edge_overflow, fsd_edge_overflow, fsd_pool_overflow, fsd_interactions, null_interactions;
    uint64_t surface_interactions, light_splats;
    uint64_t walk_iteration_cap_hits;
    uint64_t traversal_stack_dropped;   /* children a full wave-cooperative traversal stack (512 entries) could not hold: must be 0 */
} wtgpu_counters;

/* Host-side scene baking (no GPU needed): builds one of the bundled scenes
 * ("double_slits" = scenes/diffraction_simple/double_slits.xml, "cornell_box" = scenes/cornell-box/box.xml stand-in,
 * "furnace" = test scene).  Replaces scene_bootstrap_t<xml_loader_t,bvh8w_constructor_t> (src/main.cpp:634-648). */
int wtgpu_scene_create_named(const char* name, const wtgpu_scene_params* params, wtgpu_scene** out);

/* Wraps an already flattened scene (include/wtgpu_scene.h; host pointers that must outlive the handle).  This is the entry point a
 * port of the reference's own loader calls. */
int wtgpu_scene_create_from_desc(const wtgpu_scene_desc* scene_desc_host, wtgpu_scene** out);

/* Reader of the reference's XML scene format (src/scene/loader/): 14 of the 15 scene files the reference ships load
 * as they are (tests/test_xml_scene.py::test_which_of_the_shipped_scene_files_load) — <default> defines and "$name" substitution, expressions
 * with units, <include>, enabled=..., shared elements and <ref>s (bsdfs, textures, spectra, transforms); plt_bdpt / plt_path integrators; <sampler> of type
 * independent / uniform / sobolld (all served by the library's counter-based streams); perspective and virtual-plane sensors with array films (RGB / monochromatic response, polarimetric flag); spot, directional, point and area
 * emitters; diffuse, dielectric, surface_spm (dirac / fractal / gaussian profile, constant or textured roughness), twosided, scale (constant,
 * spectrum, texture), mask, normalmap and composite BSDFs; constant, checkerboard, bitmap (PNG, PFM, OpenEXR: scan-line files, NONE / RLE / ZIPS / ZIP, read through half precision), scale, transform, function and mix
 * textures; spectra by constant, rgb, blackbody, discrete, piecewise linear, ITU material, or material / emitter name (baked tables or the
 * database files of data/ior, data/emission); rectangle, cube, sphere, cylinder, prism, lens shapes and PLY / OBJ meshes with general to_world
 * transforms (a Git-LFS pointer in place of an asset: a stand-in, or skipped with -Dwtgpu_missing_assets=skip).  OBJ meshes take a material group with `mtl` (the reference's filter).  Not read: textured area-emitter
 * radiance, JPEG images, tiled / deep / PIZ-compressed EXR files; `sobolld`'s low-discrepancy point set itself is not reproduced (a scene that asks for it
 * renders with the same counter-based streams as every other sampler type).  `defines`: n_defines strings "name=value", the -D
 * defines of the reference's command line (src/main.cpp:805-928).  params (may be NULL): res (becomes the define "res" unless given), max_depth /
 * fsd / mis / rr / force_ray_tracing overrides, lut_* resolution, polarimetric.  Anything outside that vocabulary fails with a message. */
int wtgpu_scene_create_from_xml(const char* path, const char* const* defines, uint32_t n_defines, const wtgpu_scene_params* params, wtgpu_scene** out);

int wtgpu_scene_get_info(const wtgpu_scene* scene, wtgpu_scene_info* info);

/* The flattened description of a handle (for CPU-side checkers and tools). */
const wtgpu_scene_desc* wtgpu_scene_host_desc(const wtgpu_scene* scene);

/* Copies the flattened scene to `device` and allocates the per-sample path state: three slices (one internal stream each), each for a
 * batch of up to `max_batch_samples` samples (0: one sample per sensor element, at most 4 M; at most 16 M), shrunk to fit the memory budget —
 * WTGPU_STATE_GB (default 224) and never more than 85 % of the device's free memory.  A render call is cut into batches of that size; larger
 * batches are faster (every batch pays its own thin tail: DESIGN.md, section 0; bench.py renders ~4 M samples per batch).  Refuses runtime
 * settings that are known to serialise the internal streams (an explicit GPU_MAX_HW_QUEUES < 4 or HSA_KERNARG_POOL_SIZE < 4 MiB;
 * WTGPU_ALLOW_SLOW_RUNTIME=1 overrides).  Fails with WTGPU_ERR_NO_DEVICE when
 * no HIP device is present: there is no CPU fallback. */
int wtgpu_scene_upload(wtgpu_scene* scene, int device, uint64_t max_batch_samples);

/* Renders sample indices [sample_begin, sample_end) of every sensor element (the `spp` loop of
 * integrator_t::integrate for all pixels) and accumulates into the caller-owned DEVICE film buffers
 *     d_value  [height][width][channels][stokes] f64,  d_weight [height][width] f64,  d_light [height][width][channels][stokes] f64.
 * `stream` is a hipStream_t (NULL = default stream).  The work runs on internal streams that start after everything already on
 * `stream`, and `stream` continues after them: synchronise `stream` before reading the films.  (The call returns once the LAST part of
 * the work is enqueued; since round 4 a batch of samples is enqueued in two parts with a look at its round queue in between, so the
 * calling thread waits for most of the work to have run — see wtgpu_join.)  RNG: Philox-4x32-10 keyed by `seed`, counter = (pixel, sample, stream). */
int wtgpu_render(wtgpu_scene* scene, void* stream, double* d_value, double* d_weight, double* d_light, uint64_t sample_begin, uint64_t sample_end,
                 uint64_t seed);

/* The two halves of wtgpu_render.  wtgpu_render_async enqueues the work behind everything already on `stream` but does not make
 * `stream` wait for it, so consecutive calls (more samples into the same accumulators) pipeline on the GPU; wtgpu_join makes
 * `stream` continue after everything enqueued so far.  Nothing may read or overwrite the films between an async render and its
 * join.  A batch is enqueued in two parts: generation and the rounds its walks are expected to need, and — after the host has seen
 * its round queue empty (looking again every 8 rounds otherwise) — the connections.  wtgpu_render_async enqueues the first part of
 * its batches (and the second part of whichever earlier batch still holds the state slice it reuses: it may wait for that one);
 * wtgpu_join WAITS on the host for the first parts of the batches still pending (serving whichever is ready first), enqueues their second parts, and makes `stream`
 * continue after them. */
int wtgpu_render_async(wtgpu_scene* scene, void* stream, double* d_value, double* d_weight, double* d_light, uint64_t sample_begin,
                       uint64_t sample_end, uint64_t seed);
int wtgpu_join(wtgpu_scene* scene, void* stream);

/* The render seam's control surface (scene_renderer_t: progress callback + terminate interrupt polled between jobs,
 * include/wt/scene/scene_renderer.hpp:42-62, src/scene/render.cpp:306-368).  Renders [sample_begin, sample_end) in chunks of
 * `chunk_spp` samples per element (0: 1), BLOCKING: after every chunk `stream` is synchronised, progress(samples_done, samples_total,
 * user) is called (may be NULL) and the interrupts are processed (cancel below; pause / resume / capture intermediate further down); a non-zero return of the callback or a wtgpu_cancel() from any thread
 * ends the render after the current chunk with WTGPU_CANCELLED — the films then hold exactly the completed chunks
 * (*spe_done samples per element; may be NULL), which is what the reference's `capture intermediate` develops. */
typedef int (*wtgpu_progress_cb)(uint64_t samples_done, uint64_t samples_total, void* user);
int wtgpu_render_progressive(wtgpu_scene* scene, void* stream, double* d_value, double* d_weight, double* d_light, uint64_t sample_begin,
                             uint64_t sample_end, uint64_t seed, uint32_t chunk_spp, wtgpu_progress_cb progress, void* user, uint64_t* spe_done);
int wtgpu_cancel(wtgpu_scene* scene);   /* thread-safe; cleared when the next wtgpu_render_progressive starts; also lifts a pause (a full reset) */
/* The renderer's other three interrupts (include/wt/scene/interrupts.hpp; scene_renderer_t::process_interrupts, src/scene/render.cpp:329-368),
 * all thread-safe, all taking effect at the next chunk boundary of a running wtgpu_render_progressive:
 *   wtgpu_pause / wtgpu_resume      the render thread stops launching (it polls every millisecond) until resumed or cancelled; the films hold
 *                                   exactly the completed chunks while it waits.  A pause is STICKY: issued while no render runs it holds the
 *                                   next wtgpu_render_progressive after its first chunk; wtgpu_resume or wtgpu_cancel lift it.
 *   wtgpu_capture_intermediate      `capture intermediate`: the render thread calls capture(samples_per_element_done, user) ONCE at the next chunk
 *                                   boundary (also while paused), with `stream` synchronised — the films then hold exactly the completed chunks and
 *                                   the callback may download / develop them (wtgpu_develop) — and carries on in the state it was in, like the
 *                                   reference, which pauses, develops and restores the paused state.  One request may be pending at a time (a
 *                                   second one replaces it); a request made while no render runs is served by the next render's first boundary. */
int wtgpu_pause(wtgpu_scene* scene);
int wtgpu_resume(wtgpu_scene* scene);
typedef void (*wtgpu_capture_cb)(uint64_t samples_per_element_done, void* user);
int wtgpu_capture_intermediate(wtgpu_scene* scene, wtgpu_capture_cb capture, void* user);

/* Multi-GPU, one process per GPU: samples are sharded by sample index, every rank renders into its own films, and
 * the three linear accumulators are summed onto `root` with one RCCL reduce each over xGMI — film_storage_t's merge of per-worker
 * images (include/wt/sensor/film/film_storage.hpp:276-287,411-413) across devices.  Rank 0 creates the id (WTGPU_COMM_ID_BYTES) and
 * hands it to the other ranks by whatever means the host has (MPI, a file, torch.distributed); every rank then calls wtgpu_comm_create
 * with the device its scene is uploaded to.  n_value = width*height*channels*stokes doubles, n_weight = width*height. */
#define WTGPU_COMM_ID_BYTES 128
int wtgpu_comm_unique_id(void* id_out);
int wtgpu_comm_create(int world_size, int rank, int device, const void* id, wtgpu_comm** out);
int wtgpu_film_reduce(wtgpu_comm* comm, void* stream, double* d_value, double* d_weight, double* d_light, uint64_t n_value, uint64_t n_weight,
                      int root);
void wtgpu_comm_destroy(wtgpu_comm* comm);

/* Per-query ADS entry points (ads_t::intersect(ray) / integrator::traverse(cone), include/wt/ads/ads.hpp:71-113,
 * include/wt/integrator/traversal.hpp:94-172) on device-resident arrays; used by the traversal parity tests.
 * rays:  n x {ox,oy,oz, dx,dy,dz, tmin,tmax}            cones: n x {ox,oy,oz, dx,dy,dz, tan_alpha, x0, ecc, lambda_m} */
int wtgpu_trace_rays(wtgpu_scene* scene, void* stream, const float* d_rays, uint32_t n, float* d_dist, uint32_t* d_tuid, float* d_bary,
                     uint32_t* d_front);
int wtgpu_traverse_cones(wtgpu_scene* scene, void* stream, const float* d_cones, uint32_t n, uint32_t cap, float* d_dist, uint32_t* d_flags,
                         uint32_t* d_ntris, uint32_t* d_tris);

/* Region summaries of n cone queries (same cone layout): the traversal policy's hit (dist, flags: 1 empty, 2 ballistic, 4 front face),
 * the triangle under the beam axis (find_closest_triangle, plt_bdpt_detail.hpp:362-389; 0xFFFFFFFF: none) and — for diffusive hits — the
 * interaction region [dist, dist + 2 x major axis] walked IN FULL, whatever its size (the reference's unbounded record,
 * include/wt/ads/traversal_common.hpp:124-148): triangle count, number + sorted ids (first edge_cap) of its classified edges, and the
 * fraction of a Gaussian beam (sigma = cross-section axes / 3) its triangles facing like the closest one intercept
 * (plt_bdpt_detail.hpp:391-416).  Used by the parity tests of regions beyond the device's 64-triangle fast path. */
int wtgpu_query_regions(wtgpu_scene* scene, void* stream, const float* d_cones, uint32_t n, uint32_t edge_cap, float* d_dist, uint32_t* d_flags,
                        uint32_t* d_primary, uint32_t* d_ntris, uint32_t* d_nedges, uint32_t* d_edges, float* d_flux);

/* Profiling aid: `repeats` launches of a streaming copy of n_dwords 32-bit words (one dword per lane, coalesced: the access width
 * of the SoA path state) with a known byte count, used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE (tools/profile_round.sh). */
int wtgpu_calibrate_copy(uint64_t n_dwords, int repeats);

/* Accumulated device counters since upload / last reset. */
int wtgpu_get_counters(wtgpu_scene* scene, wtgpu_counters* out);
int wtgpu_reset_counters(wtgpu_scene* scene);

/* Device time [ms] per kernel, ACCUMULATED over all wtgpu_render calls since upload / the last wtgpu_reset_counters, measured
 * with hipEvents on the internal streams the kernels are launched on (waits for the in-flight batches):
 * out[0]=generate, out[1]=trace (sum over rounds), out[2]=interact first pass (sum), out[3]=connect (4 kernels), out[4]=rounds with work,
 * out[5]=trace launches with work, out[6]=batches, out[7]=cooperative (heavy) trace (sum), out[8]=region edge sets + second
 * interaction pass (k_edges, k_interact_b), out[9]=region power sums (k_flux_split, k_flux_tasks), out[10]=Fraunhofer sampling pass
 * (k_interact_c), out[11] mean number of rounds launched per batch.  Batches run concurrently on
 * several streams, so the sums may exceed the wall time. */
int wtgpu_last_render_timings(wtgpu_scene* scene, float out[12]);

/* Host-side film development (render_context_t::develop, src/scene/render.cpp:245-291):
 * out[h][w][c] = value/weight (0 if weight==0) + light * (1/spe). */
int wtgpu_develop(const wtgpu_scene* scene, const double* value, const double* weight, const double* light, uint64_t spe, float* out);

void wtgpu_scene_destroy(wtgpu_scene* scene);
const char* wtgpu_last_error(void);
const char* wtgpu_scene_stats_json(const wtgpu_scene* scene);

#ifdef __cplusplus
}
#endif
#endif /* WTGPU_H */
