// wave_tracer_amd — the whole device code and the host side as ONE translation unit (`make unity`): the layout of rounds 1-4.  The default build
// compiles one translation unit per kernel group (wtgpu_kernels.h).
#include "wtgpu.hip"
#include "kernels_trace.hip"
#include "kernels_walk.hip"
#include "kernels_fsd.hip"
#include "kernels_path.hip"
#include "kernels_connect.hip"
