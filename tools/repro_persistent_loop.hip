// Minimal form of round 5's compiler finding (wtgpu_kernels.h: wave_grab0).  Compile for the device only and read the assembly:
//     hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S -o repro.s tools/repro_persistent_loop.hip
// k_old: the idiom every wavefront-per-item kernel of rounds 2-5 began its persistent loop with.  A one-wavefront block, lane 0 takes the next
//   queue item and hands it to the other lanes through a __shared__ word between two __syncthreads(); the loop body ends with a store by lane 0.
//   hipcc (ROCm 7.2) emits the atomic and the store of the shared word in the OUTER of two nested loops and the load of the shared word in the header
//   of the INNER one (`Depth=2`), which the lanes that skip the atomic — all but lane 0 — go round on their own: they process the same item for ever.
//   (The two `; wave barrier` comments around the ds_read are what __syncthreads() becomes in a block of one wavefront.)
// k_new: the form kept — a convergent marker IN FRONT of the branch, the lane index from mbcnt, readfirstlane for the hand-over: one loop.
// tests/test_source_rules.py compiles this file and checks k_new; tools/check_persistent_loops.py checks every kernel of the library.
#include <hip/hip_runtime.h>

__device__ float work(const float* in, unsigned w, int lane, unsigned n_e) {
    double acc = 0;
    for (unsigned i = (unsigned)lane; i < n_e; i += 64u) {
        const float v = in[w * 64u + i];
        if (v < 0.f) continue;
        acc += (double)sqrtf(v);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    return (float)acc;
}

extern "C" __global__ void __launch_bounds__(64, 3) k_old(unsigned* ctl, const unsigned* queue, const unsigned* empty, const float* in, float* out, const unsigned* ne) {
    __shared__ unsigned s_item;
    const unsigned n = ctl[1];
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(ctl, 1u);
        __syncthreads();
        const unsigned item = s_item;
        __syncthreads();
        if (item >= n) break;
        const unsigned w = queue[item];
        if (empty[w]) continue;
        const float f = work(in, w, threadIdx.x & 63, ne[w]);
        if (threadIdx.x == 0) out[w] = f;
    }
}

extern "C" __global__ void __launch_bounds__(64, 3) k_new(unsigned* ctl, const unsigned* queue, const unsigned* empty, const float* in, float* out, const unsigned* ne) {
    const unsigned n = ctl[1];
    for (;;) {
        __builtin_amdgcn_wave_barrier();
        unsigned old = 0;
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) old = atomicAdd(ctl, 1u);
        __builtin_amdgcn_wave_barrier();
        const unsigned item = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
        if (item >= n) break;
        const unsigned w = queue[item];
        if (empty[w]) continue;
        const float f = work(in, w, threadIdx.x & 63, ne[w]);
        if (threadIdx.x == 0) out[w] = f;
    }
}
