// oracle/oracle.cpp — CPU checker for the wave_tracer_amd hot path.            *** TEST INFRASTRUCTURE ***
//
// What this is: a scalar, per-sample, per-pixel CPU rendering loop that follows the reference's control flow
// (src/integrator/plt_bdpt.cpp:43-148: for each sample { spectral+emitter sample; sensor sample; sensor subpath;
// emitter subpath; all (s,t) connections with MIS; splat }; for plt_path scenes src/integrator/plt_path.cpp:39-50: one walk per
// sample with next-event estimation and UTD diffraction), parallelised over pixel tiles with std::thread like the
// reference's render loop (src/scene/render.cpp:99-172: 24x24-pixel blocks).  It is used ONLY by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product path.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path and cannot be built here
// (all submodules / LFS assets are absent), so this checker cannot be validated against the
// reference's own outputs.  It shares the low-level physics headers (wave_tracer_amd/csrc/wt/*.h, each function
// citing the reference file:line it restates) with the HIP kernels; what it checks independently is the GPU
// orchestration (wavefront scheduling, SoA state, LDS stacks, queues, atomics) — sample for sample, with identical
// counter-based random numbers.  The physics primitives themselves are pinned by closed-form / numpy / scipy
// known-answer tests in tests/test_kat*.py, the analytic double-slit fringe gate and the closed forms of
// tests/test_path_oracle.py (free-space coverage, white furnace).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <set>
#include <vector>

#include "../wave_tracer_amd/csrc/wt/bdpt.h"
#include "../wave_tracer_amd/csrc/wt/path.h"

using namespace wt;

namespace {

// traversal stack of the checker: never the limit (an unbudgeted query on a full stack stops the process, wt/bvh.h: cq_node_step)
constexpr uint32_t kOracleStack = 4096;
// (kMaxWalkIters — the cap on trace / interact rounds per subpath — is wt/bdpt.h's: one definition for the device driver and this one)

struct sample_scratch_t {
    std::vector<uint32_t> svert, evert;   // vertex stores (stride 1)
    stack_entry_t stack[kOracleStack];
    std::vector<uint32_t> tris;   // unbounded in the reference (std::vector): 2^18 entries here
    std::vector<float> dists;     // ... and their cone-hit distances (uint_list_t::d, wt/bvh.h)
};
constexpr uint32_t kOracleConeTris = 1u << 18;
// 1 (default): interaction records drop the triangles beyond their final slab (traversal_common.hpp:131-135 as written); 0: the
// reference's executed behaviour (the distance is never recorded, the filter never fires) — to measure what that costs
int g_region_filter = 1;
int g_traverse_axis = 0;   // oracle_traverse_cones: run the device form of the traversal policy (wt::traverse_axis) instead of the reference's
// renders (run_walk / run_path_sample): 1 = trace with wt::traverse_axis exactly as the device's per-lane kernel calls it (early exit of
// too-short attempts, the two remembered rejecting triangles), 0 = the reference's form.  The results must be identical.
// 2: without the remembered triangles, 3: without the early exit either, 4: like 1 with a work budget of 12 units per cone query and
// the over-budget queries resumed from their hand-over record (what the device's tiers do).
int g_walk_axis = 0;
// 1 / 2: surface interactions through the device's material-sorted pass A (wt/bdpt.h: bdpt_classify + bdpt_surface_step; 1 = the class-agnostic
// instantiation, 2 = one instantiation per material class, as the class kernels run it) instead of bdpt_walk_step's surface branch.  The
// results must be identical (tests/test_oracle.py::test_split_surface_step_is_the_walk_step).
int g_split_step = 0;
// 1: every (s,t) strategy the way the device's staged connection kernels run it (wt/bdpt.h: bdpt_strategy<true> — flux without the shadow ray, the
// ray, MIS + splat with the temporary vertex formed again).  The results must be identical (test_staged_connections_are_the_connections).
int g_staged_connect = 0;
int g_fine_items = 0;   // oracle_set_fine_items
double g_last_utilisation = 1.0;   // of the worker threads of the last render (oracle_last_utilisation)

void add_counters(bdpt_counters_t& a, const bdpt_counters_t& b) {
    unsigned long long* pa = reinterpret_cast<unsigned long long*>(&a);
    const unsigned long long* pb = reinterpret_cast<const unsigned long long*>(&b);
    for (size_t i = 0; i < sizeof(bdpt_counters_t) / sizeof(unsigned long long); ++i) pa[i] += pb[i];
}

void run_walk(const scene_t& sc, walk_t& w, const vertex_store_t& vs, const fsd_pool_t& pool, uint64_t seed, uint64_t sample_id, uint32_t stream,
              sample_scratch_t& scr, bdpt_counters_t& ctr) {
    const stack_ref_t stack = make_flat_stack(scr.stack, kOracleStack);
    const uint_list_t tris{scr.tris.data(), 1, kOracleConeTris, g_region_filter ? scr.dists.data() : nullptr};
    for (uint32_t it = 0; it < kWalkIterLimit && w.active; ++it) {
        const cone_t env = walk_trace_envelope(sc, w);
        const bool rt = sc.sensor.ray_trace_only || sc.opts.force_ray_tracing;
        trav_result_t tr = g_walk_axis ? traverse_axis(sc, env, wavenum_to_wavelen_m(w.beam.k), WT_INF, rt, stack, tris, nullptr, g_walk_axis == 4 ? 12u : 0xFFFFFFFFu, g_walk_axis != 3, false,
                                                       g_walk_axis == 1 || g_walk_axis == 4 ? w.prev_offset_tuid : kInvalid)
                                       : traverse(sc, env, wavenum_to_wavelen_m(w.beam.k), WT_INF, rt, stack, tris);
        if (g_walk_axis == 4 && tr.aborted == 1) {   // the hand-over of an over-budget query: resumed (second per-lane tier) from the record alone
            const trav_result_t h = tr;
            tr = traverse_axis(sc, env, wavenum_to_wavelen_m(w.beam.k), WT_INF, rt, stack, tris, nullptr, 0xFFFFFFFFu, true, false, w.prev_offset_tuid, &h);
        }
        ctr.segments++;
        ctr.ray_queries += tr.n_ray_queries;
        ctr.cone_queries += tr.n_cone_queries;
        ctr.cone_tri_overflow += tr.overflow;
        if (g_split_step) {
            primary_hit_t ph;
            const uint32_t cls = bdpt_classify(sc, w.beam.env.d, beam_is_ray(w.beam), tr, tris, nullptr, ph);
            if (cls == WCLS_END) {
                w.active = 0;
                continue;
            }
            if (cls != WCLS_NO_PRIMARY) {
                const walk_rec_t wr{reinterpret_cast<uint32_t*>(&w)};
                bool cont;
                if (g_split_step == 2 && cls == WCLS_DIFFUSE)
                    cont = bdpt_surface_step<MAT_DIFFUSE>(sc, wr, tr.origin, tr.dist, ph, vs, seed, sample_id, stream, &ctr);
                else if (g_split_step == 2 && cls == WCLS_DIELECTRIC)
                    cont = bdpt_surface_step<MAT_DIELECTRIC>(sc, wr, tr.origin, tr.dist, ph, vs, seed, sample_id, stream, &ctr);
                else if (g_split_step == 2 && cls == WCLS_SPM)
                    cont = bdpt_surface_step<MAT_SURFACE_SPM>(sc, wr, tr.origin, tr.dist, ph, vs, seed, sample_id, stream, &ctr);
                else
                    cont = bdpt_surface_step<-1>(sc, wr, tr.origin, tr.dist, ph, vs, seed, sample_id, stream, &ctr);
                w.active = cont ? 1u : 0u;
                continue;
            }
        }
        w.active = bdpt_walk_step(sc, w, tr, tris, vs, pool, seed, sample_id, stream, &ctr) ? 1u : 0u;
    }
    w.active = 0;
}


// plt_path: src/integrator/plt_path.cpp:39-50 + plt_path_detail.hpp:772-828 — one walk per sample
void run_path_sample(const scene_t& sc, const film_t& film, uint64_t seed, uint64_t sample_id, uint32_t x, uint32_t y, sample_scratch_t& scr,
                     std::vector<utd_edge_rec_t>& utd, bdpt_counters_t& ctr) {
    const stack_ref_t stack = make_flat_stack(scr.stack, kOracleStack);
    const uint_list_t tris{scr.tris.data(), 1, kOracleConeTris, g_region_filter ? scr.dists.data() : nullptr};
    const uint32_t stream = sc.opts.integrator == INTEGRATOR_PATH_FORWARD ? STREAM_EMITTER_WALK : STREAM_SENSOR_WALK;
    const bool rt = sc.sensor.ray_trace_only || sc.opts.force_ray_tracing;
    path_walk_t pw;
    path_generate(sc, seed, sample_id, x, y, pw);
    for (uint32_t it = 0; it < kWalkIterLimit && pw.w.active; ++it) {
        const cone_t env = walk_trace_envelope(sc, pw.w);
        const trav_result_t tr = g_walk_axis ? traverse_axis(sc, env, wavenum_to_wavelen_m(pw.w.beam.k), WT_INF, rt, stack, tris, nullptr, 0xFFFFFFFFu, g_walk_axis != 3, false, g_walk_axis == 1 ? pw.w.prev_offset_tuid : kInvalid)
                                             : traverse(sc, env, wavenum_to_wavelen_m(pw.w.beam.k), WT_INF, rt, stack, tris);
        ctr.segments++;
        ctr.ray_queries += tr.n_ray_queries;
        ctr.cone_queries += tr.n_cone_queries;
        ctr.cone_tri_overflow += tr.overflow;
        pw.w.active = path_walk_step(sc, pw, tr, tris, utd.data(), utd_pool_t{utd.data(), nullptr, kUtdMaxEdges}, film, seed, sample_id, stream, stack, &ctr) ? 1u : 0u;
    }
    path_finish(sc, film, pw);
}

}   // namespace

extern "C" {

// Renders samples [sample_begin, sample_end) of every pixel into (value, weight, light) (accumulating).
// `scene_host` points to a wt::scene_t whose pointers are host pointers.
// `tile_stride`/`tile_offset`: only the 24x24 blocks with index % stride == offset are rendered (a bounded sample of a
// full-size workload; stride 1 = everything).  Returns the number of samples rendered through *n_samples_out.
static int oracle_render_impl(const void* scene_host, uint64_t sample_begin, uint64_t sample_end, uint64_t seed, double* value, double* weight, double* light,
                              int n_threads, unsigned long long* counters_out, uint32_t tile_stride, uint32_t tile_offset, uint64_t* n_samples_out) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    const uint32_t W = sc.sensor.width, H = sc.sensor.height;
    film_t film{value, weight, light, W, H, sc.sensor.channels};
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    const uint32_t B = 24;   // include/wt/wt_context.hpp:45
    const uint32_t bx = (W + B - 1) / B, by = (H + B - 1) / B;
    std::atomic<uint32_t> next{0};
    std::atomic<uint64_t> n_done{0};
    std::vector<bdpt_counters_t> ctrs(n_threads);
    for (auto& c : ctrs) std::memset(&c, 0, sizeof(c));
    std::vector<double> busy(n_threads, 0.0);
    const uint64_t n_spp = sample_end > sample_begin ? sample_end - sample_begin : 0;
    // (one sample index of a block per work item only on request — bench.py's cpu_baseline: several threads then add to the same pixel in an
    // order that differs from run to run, and the tests compare multi-threaded renders bit for bit)
    const uint32_t items_per_block = (g_fine_items && n_threads > 1 && n_spp > 1 && n_spp <= 4096) ? (uint32_t)n_spp : 1u;
    // FSD aperture pool: per thread, reset per sample (apertures only live for one sample)
    auto worker = [&](int tid) {
        sample_scratch_t scr;
        scr.tris.resize(kOracleConeTris);
    scr.dists.resize(kOracleConeTris);
        scr.dists.resize(kOracleConeTris);
        scr.svert.resize(((size_t)sc.opts.max_depth + 2) * kVertexWords);
        scr.evert.resize(((size_t)sc.opts.max_depth + 2) * kVertexWords);
        std::vector<fsd_aperture_t> hdr(2 * (size_t)kMaxWalkIters + 8);   // one per walk step at most (a step that restarts behind an empty aperture keeps its slot)
        std::vector<fsd_edge_t> edges(hdr.size() * (size_t)kFsdMaxEdges);
        uint32_t pool_counter = 0;
        const fsd_pool_t pool{hdr.data(), edges.data(), &pool_counter, (uint32_t)hdr.size(), nullptr, 0};
        bdpt_counters_t& ctr = ctrs[tid];
        const stack_ref_t stack = make_flat_stack(scr.stack, kOracleStack);
        std::vector<utd_edge_rec_t> utd(kUtdMaxEdges);
        const auto t_begin = std::chrono::steady_clock::now();
        for (;;) {
            const uint32_t item = next.fetch_add(1);
            const uint32_t blk = item / items_per_block;
            if (blk >= bx * by) break;
            if (tile_stride > 1 && blk % tile_stride != tile_offset) continue;
            // the samples of this work item: all of the block's (one thread: the summation order the golden fixtures were made with), or one
            // sample index of it (several threads: 24 x 24 x spp samples per item left a 256-thread host waiting for its slowest blocks)
            const uint64_t s_lo = items_per_block > 1 ? sample_begin + item % items_per_block : sample_begin;
            const uint64_t s_hi = items_per_block > 1 ? s_lo + 1 : sample_end;
            const uint32_t x0 = (blk % bx) * B, y0 = (blk / bx) * B;
            for (uint32_t y = y0; y < std::min(H, y0 + B); ++y)
                for (uint32_t x = x0; x < std::min(W, x0 + B); ++x)
                    for (uint64_t s = s_lo; s < s_hi; ++s) {
                        n_done.fetch_add(1, std::memory_order_relaxed);
                        const uint64_t pix = (uint64_t)y * W + x;
                        const uint64_t sample_id = (pix << 32) | (s & 0xFFFFFFFFull);
                        if (sc.opts.integrator != INTEGRATOR_BDPT) {
                            run_path_sample(sc, film, seed, sample_id, x, y, scr, utd, ctr);
                            continue;
                        }
                        pool_counter = 0;
                        sample_ctx_t ctx;
                        walk_t sw, ew;
                        const vertex_store_t svs{scr.svert.data(), 1, 0}, evs{scr.evert.data(), 1, 0};
                        bdpt_generate(sc, seed, sample_id, x, y, ctx, sw, ew, svs, evs);
                        run_walk(sc, sw, svs, pool, seed, sample_id, STREAM_SENSOR_WALK, scr, ctr);
                        run_walk(sc, ew, evs, pool, seed, sample_id, STREAM_EMITTER_WALK, scr, ctr);
                        if (g_staged_connect)
                            bdpt_connect_all<true>(sc, pool, film, svs, evs, (int)sw.nverts, (int)ew.nverts, ctx, seed, sample_id, stack, &ctr, nullptr);
                        else
                            bdpt_connect_all(sc, pool, film, svs, evs, (int)sw.nverts, (int)ew.nverts, ctx, seed, sample_id, stack, &ctr, nullptr);
                    }
        }
        busy[tid] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(worker, t);
    for (auto& t : th) t.join();
    {   // utilisation of the workers over the render: mean busy time / longest busy time (1 = nobody waited for a straggler)
        double sum = 0, mx = 0;
        for (double b : busy) {
            sum += b;
            mx = std::max(mx, b);
        }
        g_last_utilisation = mx > 0 ? sum / (mx * n_threads) : 1.0;
    }
    if (counters_out) {
        bdpt_counters_t total;
        std::memset(&total, 0, sizeof(total));
        for (auto& c : ctrs) add_counters(total, c);
        std::memcpy(counters_out, &total, sizeof(total));
    }
    if (n_samples_out) *n_samples_out = n_done.load();
    return 0;
}

int oracle_render(const void* scene_host, uint64_t sample_begin, uint64_t sample_end, uint64_t seed, double* value, double* weight, double* light,
                  int n_threads, unsigned long long* counters_out /* sizeof(bdpt_counters_t)/8 entries or NULL */) {
    return oracle_render_impl(scene_host, sample_begin, sample_end, seed, value, weight, light, n_threads, counters_out, 1, 0, nullptr);
}
int oracle_render_tiles(const void* scene_host, uint64_t sample_begin, uint64_t sample_end, uint64_t seed, double* value, double* weight, double* light,
                        int n_threads, unsigned long long* counters_out, uint32_t tile_stride, uint32_t tile_offset, uint64_t* n_samples_out) {
    return oracle_render_impl(scene_host, sample_begin, sample_end, seed, value, weight, light, n_threads, counters_out, tile_stride, tile_offset,
                              n_samples_out);
}

// Work profile of the device's traversal configuration (bounded 64-entry triangle list + pruning, any-hit probe first,
// unlimited per-lane budget) on a tile subset: per traverse() call the BVH work split into ray / cone / probe parts.
// out_calls: n x 8 uint32 {ray_nodes, ray_tris, cone_nodes, cone_tris, probe_nodes, probe_tris, n_ray_q | n_cone_q<<16, flags}
// Returns the number of traverse() calls (<= cap rows are stored).  Diagnostic tool, not part of any parity claim.
uint64_t oracle_profile_traversal(const void* scene_host, uint64_t seed, uint32_t tile_stride, uint32_t* out_calls, uint64_t cap, int unbounded) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    const uint32_t W = sc.sensor.width, H = sc.sensor.height, B = 24;
    const uint32_t bx = (W + B - 1) / B, by = (H + B - 1) / B;
    std::vector<double> value((size_t)W * H * film_planes(sc.sensor)), weight((size_t)W * H), light((size_t)W * H * film_planes(sc.sensor));
    film_t film{value.data(), weight.data(), light.data(), W, H, sc.sensor.channels};
    sample_scratch_t scr;
    scr.tris.resize(kOracleConeTris);
    scr.dists.resize(kOracleConeTris);
    scr.svert.resize(((size_t)sc.opts.max_depth + 2) * kVertexWords);
    scr.evert.resize(((size_t)sc.opts.max_depth + 2) * kVertexWords);
    std::vector<fsd_aperture_t> hdr(2 * (size_t)kMaxWalkIters + 8);   // one per walk step at most (a step that restarts behind an empty aperture keeps its slot)
    std::vector<fsd_edge_t> edges(hdr.size() * (size_t)kFsdMaxEdges);
    uint32_t pool_counter = 0;
    const fsd_pool_t pool{hdr.data(), edges.data(), &pool_counter, (uint32_t)hdr.size(), nullptr, 0};
    bdpt_counters_t ctr;
    std::memset(&ctr, 0, sizeof(ctr));
    const stack_ref_t stack = make_flat_stack(scr.stack, kOracleStack);
    uint64_t n_calls = 0;
    for (uint32_t blk = 0; blk < bx * by; ++blk) {
        if (tile_stride > 1 && blk % tile_stride != 0) continue;
        const uint32_t x0 = (blk % bx) * B, y0 = (blk / bx) * B;
        for (uint32_t y = y0; y < std::min(H, y0 + B); ++y)
            for (uint32_t x = x0; x < std::min(W, x0 + B); ++x) {
                const uint64_t pix = (uint64_t)y * W + x;
                const uint64_t sample_id = (pix << 32);
                pool_counter = 0;
                sample_ctx_t ctx;
                walk_t sw, ew;
                const vertex_store_t svs{scr.svert.data(), 1, 0}, evs{scr.evert.data(), 1, 0};
                bdpt_generate(sc, seed, sample_id, x, y, ctx, sw, ew, svs, evs);
                for (int which = 0; which < 2; ++which) {
                    walk_t& w = which ? ew : sw;
                    const vertex_store_t& vs = which ? evs : svs;
                    const uint_list_t tris{scr.tris.data(), 1, unbounded == 1 ? kOracleConeTris : kMaxConeTris, g_region_filter ? scr.dists.data() : nullptr};   // bounded like the device unless asked otherwise
                    for (uint32_t it = 0; it < kWalkIterLimit && w.active; ++it) {
                        const cone_t env = walk_trace_envelope(sc, w);
                        const bool rt = sc.sensor.ray_trace_only || sc.opts.force_ray_tracing;
                        bvh_counters_t bc;
                        std::memset(&bc, 0, sizeof(bc));
                        const trav_result_t tr = traverse(sc, env, wavenum_to_wavelen_m(w.beam.k), WT_INF, rt, stack, tris, &bc, 0xFFFFFFFFu, unbounded != 1);
                        if (n_calls < cap) {
                            uint32_t* o = out_calls + 8 * n_calls;
                            o[0] = bc.nodes; o[1] = bc.tri_tests; o[2] = bc.cone_nodes; o[3] = bc.cone_tri_tests; o[4] = bc.probe_nodes; o[5] = bc.probe_tri_tests;
                            o[6] = unbounded == 1 ? tr.ntris : (tr.n_ray_queries | (tr.n_cone_queries << 16));
                            o[7] = (tr.empty ? 1u : 0u) | (tr.ballistic ? 2u : 0u) | (which ? 4u : 0u) | (it << 8);
                            if (unbounded == 2) {   // beam-width study: overwrite the ray columns with the envelope's x0 / tan_alpha
                                std::memcpy(&o[0], &env.x0, 4);
                                std::memcpy(&o[1], &env.tan_alpha, 4);
                            }
                        }
                        ++n_calls;
                        w.active = bdpt_walk_step(sc, w, tr, tris, vs, pool, seed, sample_id, which ? STREAM_EMITTER_WALK : STREAM_SENSOR_WALK, &ctr) ? 1u : 0u;
                    }
                }
            }
    }
    return n_calls;
}


// Per traverse() call of a tile subset: what a predictor of "this walk's cone queries will exceed the per-lane budget" could look at, and what the
// queries then cost.  out: n x 8 float {max work units of the call's cone queries (2 per node + 1 per triangle test; device form, unbounded budget),
// cone radius at the axis hit, bounding-sphere radius of the axis-hit triangle, axis-hit distance, tan_alpha, x0, emitter walk (0 / 1), number of cone queries}.
// Diagnostic tool (tools/heavy_predictor.py), not part of any parity claim.
uint64_t oracle_profile_heavy(const void* scene_host, uint64_t seed, uint32_t tile_stride, float* out, uint64_t cap) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    const uint32_t W = sc.sensor.width, H = sc.sensor.height, B = 24;
    const uint32_t bx = (W + B - 1) / B, by = (H + B - 1) / B;
    sample_scratch_t scr;
    scr.tris.resize(kOracleConeTris);
    scr.dists.resize(kOracleConeTris);
    scr.svert.resize(((size_t)sc.opts.max_depth + 2) * kVertexWords);
    scr.evert.resize(((size_t)sc.opts.max_depth + 2) * kVertexWords);
    std::vector<fsd_aperture_t> hdr(2 * (size_t)kMaxWalkIters + 8);
    std::vector<fsd_edge_t> edges(hdr.size() * (size_t)kFsdMaxEdges);
    uint32_t pool_counter = 0;
    const fsd_pool_t pool{hdr.data(), edges.data(), &pool_counter, (uint32_t)hdr.size(), nullptr, 0};
    bdpt_counters_t ctr;
    std::memset(&ctr, 0, sizeof(ctr));
    const stack_ref_t stack = make_flat_stack(scr.stack, kOracleStack);
    uint64_t n_calls = 0;
    for (uint32_t blk = 0; blk < bx * by; ++blk) {
        if (tile_stride > 1 && blk % tile_stride != 0) continue;
        const uint32_t x0 = (blk % bx) * B, y0 = (blk / bx) * B;
        for (uint32_t y = y0; y < std::min(H, y0 + B); ++y)
            for (uint32_t x = x0; x < std::min(W, x0 + B); ++x) {
                const uint64_t pix = (uint64_t)y * W + x;
                const uint64_t sample_id = (pix << 32);
                pool_counter = 0;
                sample_ctx_t ctx;
                walk_t sw, ew;
                const vertex_store_t svs{scr.svert.data(), 1, 0}, evs{scr.evert.data(), 1, 0};
                bdpt_generate(sc, seed, sample_id, x, y, ctx, sw, ew, svs, evs);
                for (int which = 0; which < 2; ++which) {
                    walk_t& w = which ? ew : sw;
                    const vertex_store_t& vs = which ? evs : svs;
                    const uint_list_t tris{scr.tris.data(), 1, kMaxConeTris, scr.dists.data()};
                    for (uint32_t it = 0; it < kWalkIterLimit && w.active; ++it) {
                        const cone_t env = walk_trace_envelope(sc, w);
                        const float lambda_m = wavenum_to_wavelen_m(w.beam.k);
                        const bool rt = sc.sensor.ray_trace_only || sc.opts.force_ray_tracing;
                        // the device form of the policy, query by query
                        float max_units = 0.f, nq = 0.f;
                        ray_hit_t ah;
                        const bool axis_hit = ads_intersect_ray(sc, env.o, env.d, range_t{0.f, WT_INF}, stack, ah);
                        axis_walk_t a;
                        aw_begin(a, lambda_m, WT_INF, axis_hit, ah, 0xFFFFFFFFu, true, false, w.prev_offset_tuid);
                        trav_result_t r;
                        cone_query_t q;
                        for (;;) {
                            const int need = aw_next(sc, env, rt, stack, a, q, r);
                            if (need == AW_FINAL) break;
                            if (need == AW_TEST) {
                                aw_test_done(a, cone_attempt_too_short_by(sc, env, a.cand, a.sr, a.min_df_prog));
                                continue;
                            }
                            bvh_counters_t cc;
                            std::memset(&cc, 0, sizeof(cc));
                            while (cq_running(q)) {
                                while (q.s > 0 && q.leaf == 0) cq_node_step(sc, env, stack, q, &cc);
                                if (q.leaf != 0) cq_leaf_step(sc, env, stack, tris, q, &cc);
                            }
                            cq_end(env, tris, q);
                            max_units = std::max(max_units, float(kNodeBudgetCost * cc.cone_nodes + cc.cone_tri_tests));
                            nq += 1.f;
                            if (aw_query_done(sc, env, a, q.rec, r)) break;
                        }
                        if (n_calls < cap) {
                            float* o = out + 8 * n_calls;
                            float sph[4] = {0, 0, 0, 0};
                            if (axis_hit) tri_bounding_sphere(sc.tri_geo[ah.tuid].a, sc.tri_geo[ah.tuid].b, sc.tri_geo[ah.tuid].c, sph);
                            o[0] = max_units;
                            o[1] = axis_hit ? cone_axes(env, ah.dist).x : -1.f;
                            o[2] = sph[3];
                            o[3] = axis_hit ? ah.dist : -1.f;
                            o[4] = env.tan_alpha;
                            o[5] = env.x0;
                            o[6] = (float)which;
                            o[7] = nq;
                        }
                        ++n_calls;
                        const trav_result_t tr = traverse(sc, env, lambda_m, WT_INF, rt, stack, uint_list_t{scr.tris.data(), 1, kOracleConeTris, scr.dists.data()});
                        w.active = bdpt_walk_step(sc, w, tr, uint_list_t{scr.tris.data(), 1, kOracleConeTris, scr.dists.data()}, vs, pool, seed, sample_id,
                                                  which ? STREAM_EMITTER_WALK : STREAM_SENSOR_WALK, &ctr) ? 1u : 0u;
                    }
                }
            }
    }
    return n_calls;
}

// Work profile of wt::traverse_axis (the device form of the traversal policy) on a tile subset, per CONE QUERY: how the attempts of a
// segment end (too short / accepted / empty), what they cost, and whether the triangle that made an attempt too short — or the
// triangle the beam started from — would also have decided the next one (the "rejecting-triangle cache" of traverse_axis).
// out: n x 8 uint32 {call index, segment, outcome (0 too short, 1 accepted, 2 empty), cone_nodes, cone_tri_tests,
//                    flags (1: the origin triangle alone decides "too short", 2: the previous rejecting triangle does, 4: emitter walk),
//                    ray_nodes | ray_tris << 16 (first query of a call only), walk iteration}
// Diagnostic tool, not part of any parity claim.
uint64_t oracle_profile_axis(const void* scene_host, uint64_t seed, uint32_t tile_stride, uint32_t* out, uint64_t cap) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    const uint32_t W = sc.sensor.width, H = sc.sensor.height, B = 24;
    const uint32_t bx = (W + B - 1) / B, by = (H + B - 1) / B;
    sample_scratch_t scr;
    scr.tris.resize(kOracleConeTris);
    scr.dists.resize(kOracleConeTris);
    scr.svert.resize(((size_t)sc.opts.max_depth + 2) * kVertexWords);
    scr.evert.resize(((size_t)sc.opts.max_depth + 2) * kVertexWords);
    std::vector<fsd_aperture_t> hdr(2 * (size_t)kMaxWalkIters + 8);   // one per walk step at most (a step that restarts behind an empty aperture keeps its slot)
    std::vector<fsd_edge_t> edges(hdr.size() * (size_t)kFsdMaxEdges);
    uint32_t pool_counter = 0;
    const fsd_pool_t pool{hdr.data(), edges.data(), &pool_counter, (uint32_t)hdr.size(), nullptr, 0};
    bdpt_counters_t ctr;
    std::memset(&ctr, 0, sizeof(ctr));
    const stack_ref_t stack = make_flat_stack(scr.stack, kOracleStack);
    uint64_t n_rows = 0, n_calls = 0;
    auto too_short_by = [&](const cone_t& env, uint32_t tuid, const range_t& range, float min_prog) {
        if (tuid == kInvalid) return false;
        const tri_geo_t tri = sc.tri_geo[tuid];
        cone_tri_hit_t h;
        return intersect_cone_tri(env, tri.a, tri.b, tri.c, tri.n, range, h) && !(h.dist > range.max) && h.dist - range.min < min_prog;
    };
    for (uint32_t blk = 0; blk < bx * by; ++blk) {
        if (tile_stride > 1 && blk % tile_stride != 0) continue;
        const uint32_t x0 = (blk % bx) * B, y0 = (blk / bx) * B;
        for (uint32_t y = y0; y < std::min(H, y0 + B); ++y)
            for (uint32_t x = x0; x < std::min(W, x0 + B); ++x) {
                const uint64_t pix = (uint64_t)y * W + x;
                const uint64_t sample_id = (pix << 32);
                pool_counter = 0;
                sample_ctx_t ctx;
                walk_t sw, ew;
                const vertex_store_t svs{scr.svert.data(), 1, 0}, evs{scr.evert.data(), 1, 0};
                bdpt_generate(sc, seed, sample_id, x, y, ctx, sw, ew, svs, evs);
                for (int which = 0; which < 2; ++which) {
                    walk_t& w = which ? ew : sw;
                    const vertex_store_t& vs = which ? evs : svs;
                    const uint_list_t tris{scr.tris.data(), 1, kMaxConeTris, scr.dists.data()};
                    for (uint32_t it = 0; it < kWalkIterLimit && w.active; ++it) {
                        const cone_t env = walk_trace_envelope(sc, w);
                        const float lambda_m = wavenum_to_wavelen_m(w.beam.k);
                        // the policy loop of traverse_axis, instrumented
                        if (!(sc.sensor.ray_trace_only || sc.opts.force_ray_tracing) && !cone_is_ray(env)) {
                            bvh_counters_t bc;
                            std::memset(&bc, 0, sizeof(bc));
                            ray_hit_t ah;
                            const bool axis_hit = ads_intersect_ray(sc, env.o, env.d, range_t{0.f, WT_INF}, stack, ah, &bc);
                            uint32_t first = 1, prev_short = kInvalid;
                            float dist = 0.f;
                            for (uint32_t seg = 0;; ++seg) {
                                const float bd = max_ballistic_distance(lambda_m, seg, 0.f);
                                if (axis_hit && ah.dist <= fminf_(WT_INF, dist + bd * kBallisticScale)) break;
                                dist += bd;
                                if (bd == WT_INF) break;
                                const float min_df_prog = cone_axes(env, dist).x / 2.f;
                                const float cone_max = axis_hit ? cone_axis_bound(env, ah.dist) : WT_INF;
                                const range_t sr{dist, cone_max};
                                bvh_counters_t cc;
                                std::memset(&cc, 0, sizeof(cc));
                                cone_hit_t ch;
                                bvh_traverse_cone(sc, env, sr, kMajorAxisToZScale, stack, tris, ch, &cc, 0xFFFFFFFFu, min_df_prog);
                                const bool df_empty = ch.ntris == 0 && ch.overflow == 0;
                                const bool accepted = !ch.too_short && (df_empty || ch.dist - dist >= min_df_prog);
                                uint32_t fl = which ? 4u : 0u;
                                if (too_short_by(env, w.prev_offset_tuid, sr, min_df_prog)) fl |= 1u;
                                if (too_short_by(env, prev_short, sr, min_df_prog)) fl |= 2u;
                                if (n_rows < cap) {
                                    uint32_t* o = out + 8 * n_rows;
                                    o[0] = (uint32_t)n_calls; o[1] = seg; o[2] = accepted ? (df_empty ? 2u : 1u) : 0u; o[3] = cc.cone_nodes; o[4] = cc.cone_tri_tests; o[5] = fl;
                                    o[6] = first ? (bc.nodes | (bc.tri_tests << 16)) : 0u; o[7] = it;
                                }
                                ++n_rows;
                                first = 0;
                                if (ch.too_short) prev_short = ch.short_tuid;
                                if (accepted) break;
                            }
                        }
                        ++n_calls;
                        const bool rt = sc.sensor.ray_trace_only || sc.opts.force_ray_tracing;
                        const trav_result_t tr = traverse(sc, env, lambda_m, WT_INF, rt, stack, uint_list_t{scr.tris.data(), 1, kOracleConeTris, scr.dists.data()});
                        w.active = bdpt_walk_step(sc, w, tr, uint_list_t{scr.tris.data(), 1, kOracleConeTris, scr.dists.data()}, vs, pool, seed, sample_id,
                                                  which ? STREAM_EMITTER_WALK : STREAM_SENSOR_WALK, &ctr) ? 1u : 0u;
                    }
                }
            }
    }
    return n_rows;
}

#ifdef WT_PROFILE_CONE_TRI
void oracle_fsd_hist(unsigned long long* out) { std::memcpy(out, wt::g_fsd_hist, sizeof(wt::g_fsd_hist)); }
void oracle_cone_tri_exits(unsigned long long* out) { std::memcpy(out, g_cone_tri_exits, sizeof(g_cone_tri_exits)); }
#endif

// 0: dead apertures run the reference's full rejection loop (wt/fsd.h: kFsdDeadRatio); default 1e-10
void oracle_set_fsd_dead_ratio(float r) { wt::g_fsd_dead_ratio = r; }

int oracle_counters_count() { return (int)(sizeof(bdpt_counters_t) / sizeof(unsigned long long)); }

// ---- per-query entry points for the traversal parity tests ---------------------------------------------------
// rays: n x {ox,oy,oz,dx,dy,dz,tmin,tmax}; out: n x {dist, tuid(as float bits), bx, by, front}
int oracle_trace_rays(const void* scene_host, const float* rays, uint32_t n, float* out_dist, uint32_t* out_tuid, float* out_bary, uint32_t* out_front) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    stack_entry_t st[kOracleStack];
    const stack_ref_t stack = make_flat_stack(st, kOracleStack);
    for (uint32_t i = 0; i < n; ++i) {
        const float* r = rays + 8 * i;
        ray_hit_t h;
        ads_intersect_ray(sc, vec3{r[0], r[1], r[2]}, vec3{r[3], r[4], r[5]}, range_t{r[6], r[7]}, stack, h);
        out_dist[i] = h.dist;
        out_tuid[i] = h.tuid;
        out_bary[2 * i] = h.bx;
        out_bary[2 * i + 1] = h.by;
        out_front[i] = h.front_face;
    }
    return 0;
}
// The same queries through the device's 128-BYTE NODES (wt/bvh.h: bvh8_qnode_t, child boxes on a 16-bit grid over the scene, rounded outwards), built here
// from the scene's nodes exactly as wtgpu.hip builds them at upload: what a closest-hit ray query / a cone query finds must not depend on the node source.
struct grid_scene_t {
    std::vector<bvh8_qnode_t> nodes;
    qgrid_t grid;
    bool ok = false;
};
static grid_scene_t make_grid_nodes(const scene_t& sc) {
    grid_scene_t g;
    vec3 mn{WT_INF, WT_INF, WT_INF}, mx{-WT_INF, -WT_INF, -WT_INF};
    for (uint32_t i = 0; i < sc.n_nodes; ++i)
        for (int c = 0; c < 8; ++c)
            if (sc.nodes[i].child[c] != 0) {
                mn = vmin(mn, vec3{sc.nodes[i].minx[c], sc.nodes[i].miny[c], sc.nodes[i].minz[c]});
                mx = vmax(mx, vec3{sc.nodes[i].maxx[c], sc.nodes[i].maxy[c], sc.nodes[i].maxz[c]});
            }
    g.grid = qgrid_make(mn, mx);
    g.nodes.resize(sc.n_nodes);
    g.ok = true;
    for (uint32_t i = 0; i < sc.n_nodes; ++i) g.ok = qnode_make(sc.nodes[i], g.grid, g.nodes[i]) && g.ok;
    return g;
}
int oracle_trace_rays_grid(const void* scene_host, const float* rays, uint32_t n, float* out_dist, uint32_t* out_tuid) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    const grid_scene_t g = make_grid_nodes(sc);
    if (!g.ok) return 1;
    const grid_nodes_t ns{g.nodes.data(), g.grid};
    stack_entry_t st[kOracleStack];
    const stack_ref_t stack = make_flat_stack(st, kOracleStack);
    for (uint32_t i = 0; i < n; ++i) {
        const float* r = rays + 8 * i;
        ray_hit_t h;
        bvh_traverse_ray_ns<false>(ns, sc, vec3{r[0], r[1], r[2]}, vec3{r[3], r[4], r[5]}, range_t{r[6], r[7]}, stack, h);
        out_dist[i] = h.dist;
        out_tuid[i] = h.tuid;
    }
    return 0;
}
// cones: n x {ox,oy,oz, dx,dy,dz, tan_alpha, x0, ecc, lambda_m}: ONE cone query over [0, inf) (closest distance, number of listed triangles) through the
// exact nodes (which = 0) or the grid nodes (1)
int oracle_cone_queries(const void* scene_host, const float* cones, uint32_t n, int which, float* out_dist, uint32_t* out_ntris) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    const grid_scene_t g = make_grid_nodes(sc);
    if (!g.ok) return 1;
    const grid_nodes_t gs{g.nodes.data(), g.grid};
    const wide_nodes_t ws{sc.nodes};
    stack_entry_t st[kOracleStack];
    const stack_ref_t stack = make_flat_stack(st, kOracleStack);
    std::vector<uint32_t> tl(kOracleConeTris);
    std::vector<float> dl(kOracleConeTris);
    for (uint32_t i = 0; i < n; ++i) {
        const float* c = cones + 10 * (size_t)i;
        const vec3 d = normalize(vec3{c[3], c[4], c[5]});
        const cone_t env = make_cone(vec3{c[0], c[1], c[2]}, d, build_orthogonal_frame(d).t, c[6], c[8], c[7]);
        const uint_list_t tris{tl.data(), 1, kOracleConeTris, dl.data()};
        cone_hit_t ch;
        if (which)
            bvh_traverse_cone_ns(gs, sc, env, range_t{0.f, WT_INF}, kMajorAxisToZScale, stack, tris, ch);
        else
            bvh_traverse_cone_ns(ws, sc, env, range_t{0.f, WT_INF}, kMajorAxisToZScale, stack, tris, ch);
        out_dist[i] = ch.dist;
        out_ntris[i] = ch.ntris;
    }
    return 0;
}
// cones: n x {ox,oy,oz, dx,dy,dz, tan_alpha, x0, ecc, lambda_m}; runs the full traverse() policy.
// out: dist, ballistic flag, ntris, sorted tri ids (cap per query)
int oracle_traverse_cones(const void* scene_host, const float* cones, uint32_t n, uint32_t cap, float* out_dist, uint32_t* out_flags, uint32_t* out_ntris,
                          uint32_t* out_tris) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    stack_entry_t st[kOracleStack];
    const stack_ref_t stack = make_flat_stack(st, kOracleStack);
    const uint32_t lcap = cap > kMaxConeTris ? cap : kMaxConeTris;   // the device's bounded list unless a larger one is asked for
    std::vector<uint32_t> tl(lcap);
    std::vector<float> dl(lcap);
    for (uint32_t i = 0; i < n; ++i) {
        const float* c = cones + 10 * i;
        const vec3 d = normalize(vec3{c[3], c[4], c[5]});
        const cone_t env = make_cone(vec3{c[0], c[1], c[2]}, d, build_orthogonal_frame(d).t, c[6], c[8], c[7]);
        const uint_list_t tris{tl.data(), 1, lcap, g_region_filter ? dl.data() : nullptr};
        const trav_result_t tr = g_traverse_axis ? traverse_axis(sc, env, c[9], WT_INF, false, stack, tris) : traverse(sc, env, c[9], WT_INF, false, stack, tris);
        out_dist[i] = tr.dist;
        out_flags[i] = (tr.empty ? 1u : 0u) | (tr.ballistic ? 2u : 0u) | (tr.front_face ? 4u : 0u);
        out_ntris[i] = tr.ballistic ? (tr.empty ? 0 : 1) : tr.ntris;
        for (uint32_t j = 0; j < cap; ++j) out_tris[(size_t)i * cap + j] = kInvalid;
        if (tr.ballistic) {
            if (!tr.empty) out_tris[(size_t)i * cap] = tr.tuid;
        } else {
            std::vector<uint32_t> s(tl.begin(), tl.begin() + tr.ntris);
            std::sort(s.begin(), s.end());
            for (uint32_t j = 0; j < tr.ntris && j < cap; ++j) out_tris[(size_t)i * cap + j] = s[j];
        }
    }
    return 0;
}

void oracle_set_region_filter(int on) { g_region_filter = on; }
void oracle_set_traverse_axis(int on) { g_traverse_axis = on; }
void oracle_set_walk_axis(int on) { g_walk_axis = on; }
void oracle_set_split_step(int mode) { g_split_step = mode; }
double oracle_last_utilisation() { return g_last_utilisation; }
void oracle_set_fine_items(int on) { g_fine_items = on; }
void oracle_set_staged_connect(int on) { g_staged_connect = on; }

// Region summaries of cone queries of any size: what the reference's unbounded intersection record yields (`list`: every triangle
// the sequential traversal met, traversal_common.hpp:124-148) next to a brute-force scan of ALL scene triangles against the final
// slab (`slab`: the membership the device's whole-region walks use).  sigma = cross-section axes / 3.
int oracle_query_regions(const void* scene_host, const float* cones, uint32_t n, uint32_t edge_cap, float* out_dist, uint32_t* out_flags,
                         uint32_t* out_primary, uint32_t* out_ntris /* n x {list, slab} */, uint32_t* out_nedges /* n x {list, slab} */,
                         uint32_t* out_edges_list, uint32_t* out_edges_slab, float* out_flux /* n x {list, slab} */) {
    const scene_t& sc = *static_cast<const scene_t*>(scene_host);
    stack_entry_t st[kOracleStack];
    const stack_ref_t stack = make_flat_stack(st, kOracleStack);
    std::vector<uint32_t> tl(1u << 21);
    std::vector<float> dl(1u << 21);
    for (uint32_t i = 0; i < n; ++i) {
        const float* c = cones + 10 * i;
        const vec3 d = normalize(vec3{c[3], c[4], c[5]});
        const cone_t env = make_cone(vec3{c[0], c[1], c[2]}, d, build_orthogonal_frame(d).t, c[6], c[8], c[7]);
        const uint_list_t tris{tl.data(), 1, (uint32_t)tl.size(), g_region_filter ? dl.data() : nullptr};
        const trav_result_t tr = traverse(sc, env, c[9], WT_INF, false, stack, tris);
        out_dist[i] = tr.dist;
        out_flags[i] = (tr.empty ? 1u : 0u) | (tr.ballistic ? 2u : 0u) | (tr.front_face ? 4u : 0u);
        out_primary[i] = kInvalid;
        out_ntris[2 * i] = out_ntris[2 * i + 1] = out_nedges[2 * i] = out_nedges[2 * i + 1] = 0;
        out_flux[2 * i] = out_flux[2 * i + 1] = 0.f;
        for (uint32_t j = 0; j < edge_cap; ++j) out_edges_list[(size_t)i * edge_cap + j] = out_edges_slab[(size_t)i * edge_cap + j] = kInvalid;
        if (tr.ballistic) {
            out_primary[i] = tr.tuid;
            continue;
        }
        if (tr.empty) continue;
        const range_t izr{tr.dist, tr.dist + tr.region_depth};
        const frame_t fr = cone_frame(env);
        const vec2 ax = cone_axes(env, tr.dist);
        const vec2 sigma{ax.x / kBeamEnvelope, ax.y / kBeamEnvelope};
        // primary: find_closest_triangle's scan of the list (plt_bdpt_detail.hpp:362-389)
        float best = WT_INF;
        std::set<uint32_t> el, es;
        double fl = 0, fs = 0;
        for (uint32_t j = 0; j < tr.ntris; ++j) {
            const uint32_t t = tl[j];
            const tri_geo_t g = sc.tri_geo[t];
            ray_tri_hit_t h;
            if (intersect_ray_tri(env.o, env.d, g.a, g.b, g.c, grow(izr, cone_intersection_tolerance(env.o, g.a, g.b, g.c)), h) && h.dist < best) {
                best = h.dist;
                out_primary[i] = t;
            }
            for (int e = 0; e < 3; ++e)
                if (sc.tri_meta[t].edge[e] != kInvalid) el.insert(sc.tri_meta[t].edge[e]);
            fl += region_triangle_flux(sc, fr, env, izr, sigma, t, tr.front_face != 0);
        }
        uint32_t ns = 0;
        for (uint32_t t = 0; t < sc.n_tris; ++t) {
            const tri_geo_t g = sc.tri_geo[t];
            cone_tri_hit_t h;
            if (!intersect_cone_tri(env, g.a, g.b, g.c, g.n, izr, h) || h.dist > izr.max) continue;
            ++ns;
            for (int e = 0; e < 3; ++e)
                if (sc.tri_meta[t].edge[e] != kInvalid) es.insert(sc.tri_meta[t].edge[e]);
            fs += region_triangle_flux(sc, fr, env, izr, sigma, t, tr.front_face != 0);
        }
        out_ntris[2 * i] = tr.ntris;
        out_ntris[2 * i + 1] = ns;
        out_nedges[2 * i] = (uint32_t)el.size();
        out_nedges[2 * i + 1] = (uint32_t)es.size();
        uint32_t k = 0;
        for (uint32_t e : el)
            if (k < edge_cap) out_edges_list[(size_t)i * edge_cap + k++] = e;
        k = 0;
        for (uint32_t e : es)
            if (k < edge_cap) out_edges_slab[(size_t)i * edge_cap + k++] = e;
        out_flux[2 * i] = (float)fl;
        out_flux[2 * i + 1] = (float)fs;
    }
    return 0;
}

}   // extern "C"
