// wave_tracer_amd — wavefront-cooperative construction of a Fraunhofer aperture (device only).
//
// fsd_build_aperture (wt/fsd.h; src/interaction/fsd/fraunhofer/free_space_diffraction.cpp:22-129) visits the region's scene edges
// one after the other: silhouette test, clipping to the envelope ellipse, subdivision into segments, then the 0-th order power
// (eight evaluations of the scattering amplitude, each a sum over all segments) and the normalisation of the segment
// probabilities.  For a region with 10^2..10^3 edges that is 10^6 clock ticks of ONE lane while the 63 other lanes of its wavefront
// wait (measured: the per-lane pass B spent 35 ms per step of the headline workload that way).  Here the 64 lanes share the work of one
// aperture:
//   * lane = scene edge (64 at a time): segment count -> exclusive prefix sum -> every lane writes its edge's segments at its
//     offset, so the segments are stored in the order of the sequential loop (sorted edge ids, then along the edge);
//   * lane = segment for the eight amplitude sums and the normalisation.
// The sums (total segment probability, amplitudes) are accumulated in f64 and reduced by a butterfly: they agree with the sequential
// f32 sums to rounding (~1e-7 relative); everything else is the same arithmetic on the same operands.
#pragma once
#if defined(__HIPCC__)
#include "bdpt.h"
#include "coop.h"

namespace wt {

WT_D double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
WT_D uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
    return v;
}
// inclusive prefix sum over the 64 lanes
WT_D uint32_t wave_scan_u32(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// All 64 lanes of a 64-thread block call with identical arguments.  `slot`: the aperture's pool slot (allocated by the caller).
// Returns FALSE when the segment pool is exhausted (the aperture is left empty, like fsd_pool_alloc_edges' failure in bdpt_walk_step).
WT_D bool coop_build_aperture(const scene_t& sc, const frame_t& frame, float k, const cone_t& beam, const uint32_t* eids, uint32_t n_ids,
                                           vec2 sigma, const fsd_pool_t& pool, uint32_t slot, fsd_aperture_t& ap) {
    const int lane = threadIdx.x & 63;
    fsd_build_state_t st = fsd_build_begin(frame, k, 1.f, sigma, ap);
    // ---- storage: an upper bound of the segment count (fsd_count_segments), one allocation
    uint32_t need = 0;
    for (uint32_t j = lane; j < n_ids; j += 64) need += fsd_count_segments(sc, frame, beam, st.cse, st.max_edge_length, eids[j]);
    need = wave_sum_u32(need);
    uint32_t off = 0, cap = 0, okw = 1;
    if (lane == 0) {
        fsd_aperture_t tmp = ap;
        okw = fsd_pool_alloc_edges(pool, slot, need, tmp) ? 1u : 0u;
        off = tmp.edge_offset;
        cap = tmp.edge_cap;
    }
    ap.edge_offset = (uint32_t)__shfl((int)off, 0, 64);
    ap.edge_cap = (uint32_t)__shfl((int)cap, 0, 64);
    const bool ok = __shfl((int)okw, 0, 64) != 0;
    const fsd_edges_ref_t ed{pool.edges + (size_t)ap.edge_offset, 1};
    // ---- segments, 64 scene edges at a time, in the sequential order
    uint32_t total = 0;
    double psum = 0.0;
    for (uint32_t j0 = 0; j0 < n_ids; j0 += 64) {
        const uint32_t j = j0 + (uint32_t)lane;
        uint32_t cnt = 0;
        if (j < n_ids) fsd_edge_segments(sc, frame, beam, sigma, st.cse, st.max_edge_length, eids[j], [&](const fsd_edge_t&) { ++cnt; });
        const uint32_t incl = wave_scan_u32(cnt);
        uint32_t pos = total + incl - cnt;
        if (cnt)
            fsd_edge_segments(sc, frame, beam, sigma, st.cse, st.max_edge_length, eids[j], [&](const fsd_edge_t& fe) {
                if (pos < ap.edge_cap) {
                    ed.set(pos, fe);
                    psum += (double)fe.pdf;
                }
                ++pos;
            });
        total += (uint32_t)__shfl((int)incl, 63, 64);
    }
    ap.n_edges = total < ap.edge_cap ? total : ap.edge_cap;
    ap.overflow = total - ap.n_edges;
    double P_total = wave_sum(psum);
    __syncthreads();   // the segments are read back by other lanes
    // ---- fsd_build_finish: power in the 0-th order lobe (8-point average on a circle of radius 3*P0_sigma)
    const float psi0r = 3.f * kFsdP0Sigma;
    const vec2 dirs[8] = {{-kInvSqrt2, -kInvSqrt2}, {-1, 0}, {-kInvSqrt2, kInvSqrt2}, {0, 1}, {kInvSqrt2, kInvSqrt2}, {1, 0}, {kInvSqrt2, -kInvSqrt2}, {0, -1}};
    double are[8], aim[8], inc = 0.0;
#pragma unroll
    for (int d = 0; d < 8; ++d) are[d] = aim[d] = 0.0;
    for (uint32_t i = lane; i < ap.n_edges; i += 64) {
        const fsd_edge_t e = ed.get(i);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const cplx p = fsd_Psi(e, psi0r * dirs[d]);
            are[d] += (double)p.re;
            aim[d] += (double)p.im;
            inc += (double)cnorm(p);
        }
    }
    inc = wave_sum(inc);
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const float re = (float)wave_sum(are[d]), im = (float)wave_sum(aim[d]);
        acc += cnorm(cplx{re, im});
    }
    ap.psi02 = acc / 8.f;
    ap.dead = (ap.n_edges >= 2 && acc < kFsdDeadRatio * (float)inc) ? 1u : 0u;   // (fsd_build_finish)
    ap.P0 = (kTwoPi * sqr(kFsdP0Sigma) * ap.psi02) / sqr(k * 1.f);
    float Pt = (float)P_total;
    Pt += ap.P0;
    if (Pt > 0.f) {
        const float rp = 1.f / Pt;
        ap.P0_pdf = ap.P0 * rp;
        for (uint32_t i = lane; i < ap.n_edges; i += 64) {
            fsd_edge_t e = ed.get(i);
            e.pdf *= rp;
            ed.set(i, e);
        }
    } else {
        ap.P0_pdf = 1.f;
        ap.n_edges = 0;
    }
    __syncthreads();
    return ok;
}

}   // namespace wt
#endif
