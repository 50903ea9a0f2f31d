#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* p) { p[threadIdx.x] = 42 + threadIdx.x; }
extern "C" int run(int* d) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("launch: %s\n", hipGetErrorString(e)); return 1; }
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("sync: %s\n", hipGetErrorString(e)); return 2; }
    return 0;
}
