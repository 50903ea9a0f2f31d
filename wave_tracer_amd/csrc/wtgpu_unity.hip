// wave_tracer_amd — the whole device code and the host side as ONE translation unit: the DEFAULT build (`make` = `make unity`), the layout of
// rounds 1-4 and the one the GPU suite ran on.  `make split` compiles one translation unit per kernel group instead (wtgpu_kernels.h, Makefile).
#include "wtgpu.hip"
#include "kernels_trace.hip"
#include "kernels_walk.hip"
#include "kernels_fsd.hip"
#include "kernels_path.hip"
#include "kernels_connect.hip"
