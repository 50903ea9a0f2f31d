/* smoke.c — a C (not C++, not Python) host of the C-ABI, compiled against include/ only: what a wave_tracer maintainer's binding sees.
 * Bakes the bundled furnace scene, uploads it, renders 4 spp through wtgpu_render_progressive (progress callback), downloads and
 * develops the film; then wraps the flattened description (include/wtgpu_scene.h) with wtgpu_scene_create_from_desc and checks that
 * it renders the same film.  Exit code 0 = ok.  Built by `make -C wave_tracer_amd/csrc` as wave_tracer_amd/wtgpu_smoke. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wtgpu.h"

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        int rc_ = (x);                                                                             \
        if (rc_ != WTGPU_OK) {                                                                     \
            fprintf(stderr, "smoke: %s -> %d: %s\n", #x, rc_, wtgpu_last_error());                 \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)
#define HCHECK(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "smoke: %s -> %s\n", #x, hipGetErrorString(e_));                       \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

static int calls = 0;
static int on_progress(uint64_t done, uint64_t total, void* user) {
    (void)user;
    ++calls;
    return done > total;   /* never: keep going */
}

static int render_film(wtgpu_scene* sc, const wtgpu_scene_info* info, uint64_t spp, float* out) {
    const size_t nv = (size_t)info->width * info->height * info->channels * info->stokes, nw = (size_t)info->width * info->height;
    double *dv, *dw, *dl;
    HCHECK(hipMalloc((void**)&dv, nv * 8));
    HCHECK(hipMalloc((void**)&dw, nw * 8));
    HCHECK(hipMalloc((void**)&dl, nv * 8));
    HCHECK(hipMemset(dv, 0, nv * 8));
    HCHECK(hipMemset(dw, 0, nw * 8));
    HCHECK(hipMemset(dl, 0, nv * 8));
    uint64_t done = 0;
    CHECK(wtgpu_render_progressive(sc, NULL, dv, dw, dl, 0, spp, 7, 1, on_progress, NULL, &done));
    if (done != spp) return 1;
    double* hv = (double*)malloc(nv * 8);
    double* hw = (double*)malloc(nw * 8);
    double* hl = (double*)malloc(nv * 8);
    HCHECK(hipMemcpy(hv, dv, nv * 8, hipMemcpyDeviceToHost));
    HCHECK(hipMemcpy(hw, dw, nw * 8, hipMemcpyDeviceToHost));
    HCHECK(hipMemcpy(hl, dl, nv * 8, hipMemcpyDeviceToHost));
    CHECK(wtgpu_develop(sc, hv, hw, hl, spp, out));
    free(hv);
    free(hw);
    free(hl);
    HCHECK(hipFree(dv));
    HCHECK(hipFree(dw));
    HCHECK(hipFree(dl));
    return 0;
}

int main(void) {
    wtgpu_scene_params p;
    memset(&p, 0, sizeof(p));
    p.res = 24;
    p.max_depth = p.fsd = p.mis = p.rr = -1;
    p.mesh_detail = 1;
    p.lut_n_theta = p.lut_m = 32;
    wtgpu_scene *a = NULL, *b = NULL;
    CHECK(wtgpu_scene_create_named("furnace", &p, &a));
    wtgpu_scene_info info;
    CHECK(wtgpu_scene_get_info(a, &info));
    const wtgpu_scene_desc* desc = wtgpu_scene_host_desc(a);
    if (!desc || desc->n_tris != info.n_tris || desc->sensor.width != info.width) {
        fprintf(stderr, "smoke: description does not match the info\n");
        return 1;
    }
    CHECK(wtgpu_scene_upload(a, 0, 0));
    const size_t n = (size_t)info.width * info.height * info.channels * info.stokes;
    float* fa = (float*)malloc(n * 4);
    float* fb = (float*)malloc(n * 4);
    if (render_film(a, &info, 4, fa)) return 1;
    CHECK(wtgpu_scene_create_from_desc(desc, &b));
    CHECK(wtgpu_scene_upload(b, 0, 0));
    const int calls_a = calls;
    if (render_film(b, &info, 4, fb)) return 1;
    double sum = 0, diff = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!isfinite(fa[i]) || fa[i] < 0) {
            fprintf(stderr, "smoke: bad pixel value\n");
            return 1;
        }
        sum += fa[i];
        diff += fabs((double)fa[i] - fb[i]);
    }
    wtgpu_counters c;
    CHECK(wtgpu_get_counters(a, &c));
    printf("smoke.c: furnace %ux%u, 4 spp: mean %.6g, named vs from_desc rel. diff %.2e, %d progress calls, %llu samples, %llu segments\n", info.width,
           info.height, sum / (double)n, diff / sum, calls_a, (unsigned long long)c.samples, (unsigned long long)c.segments);
    wtgpu_scene_destroy(b);
    wtgpu_scene_destroy(a);
    free(fa);
    free(fb);
    return (sum > 0 && diff <= 1e-6 * sum && calls_a == 4 && c.samples == 4ull * info.width * info.height) ? 0 : 1;
}
