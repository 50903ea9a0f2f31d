#!/bin/bash
# Builds wave_tracer_amd/_v/libwtgpu_<name>.so from the working tree's wtgpu.hip with extra compiler flags (A/B variants of the
# device code: register budgets, tuning macros).  Select at run time with WTGPU_LIB=<path>.   usage: build_variant.sh <name> [flags...]
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
C=$R/wave_tracer_amd/csrc
mkdir -p $R/wave_tracer_amd/_v $C/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wno-unused-result "$@" -c -o $C/_build/wtgpu_$NAME.o $C/wtgpu.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/wave_tracer_amd/_v/libwtgpu_$NAME.so $C/_build/wtgpu_$NAME.o $C/_build/scene_builder.o $C/_build/scenes.o \
  $C/_build/xml_scene.o $C/_build/ply_loader.o $C/_build/obj_loader.o $C/_build/spectrum_db.o $C/_build/png_loader.o -L/opt/rocm/lib -lrccl -lz -Wl,-rpath,/opt/rocm/lib
echo built $R/wave_tracer_amd/_v/libwtgpu_$NAME.so
