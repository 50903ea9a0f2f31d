/* Test hooks of the bundled scenes — NOT part of the public C-ABI (include/wtgpu.h); exported for tests/ only. */
#pragma once
#include "../../include/wtgpu.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct wtgpu_test_hooks {
    uint32_t only_s, only_t; /* 0 = all strategies; v>0 evaluates only s (t) = v-1 with unit MIS weight */
    uint32_t crop_of;        /* perspective sensors: 0 = off; v>0: the res x res film is the central crop of a v x v film (same pixel pitch,
                              * hence the same beam footprints, as the full-size render) */
} wtgpu_test_hooks;
int wtgpu_scene_create_named_hooks(const char* name, const wtgpu_scene_params* params, const wtgpu_test_hooks* hooks, wtgpu_scene** out);
/* field-by-field (byte-for-byte) comparison of two flattened scenes: 0 identical, 1 different (`what`: the first difference) */
int wtgpu_scene_compare(const wtgpu_scene* a, const wtgpu_scene* b, char* what, size_t n_what);
/* the same restricted to one part: "sensor", "opts" or "emitters" (records that do not depend on the geometry) */
int wtgpu_scene_compare_part(const wtgpu_scene* a, const wtgpu_scene* b, const char* part, char* what, size_t n_what);
/* WTGPU_TRACE_AB=n (environment, read at upload): the first n rounds of every batch replay their trace queue through k_trace_refill and k_trace_sm (the
 * pipeline continues from the second one's output).  Accumulated since upload: event-timed milliseconds of each kernel, words of their outputs
 * (traversal records, triangle lists, heavy-queue checksums) that differ, walks and rounds replayed. */
int wtgpu_trace_ab_stats(wtgpu_scene* s, double* ms_refill, double* ms_sm, uint64_t* differing_words, uint64_t* walks, uint64_t* rounds);
#ifdef __cplusplus
}
#endif
