#!/bin/bash
# Builds wave_tracer_amd/_v/libwtgpu_<name>.so from the working tree's device code with extra compiler flags (A/B variants: register
# budgets, tuning macros).  Select at run time with WTGPU_LIB=<path>.   usage: build_variant.sh <name> [flags...]
# TUS="kernels_trace kernels_walk" restricts the recompilation to those translation units (the others are taken from csrc/_build, i.e. from the
# last `make`); default: all of them.
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
C=$R/wave_tracer_amd/csrc
ALL="wtgpu kernels_trace kernels_walk kernels_fsd kernels_path kernels_connect"
TUS=${TUS:-$ALL}
mkdir -p $R/wave_tracer_amd/_v $C/_build/v_$NAME
OBJS=""
for T in $ALL; do
  if [[ " $TUS " == *" $T "* ]]; then
    ( cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wno-unused-result -Wno-comment "$@" -c -o _build/v_$NAME/$T.o $T.hip ) &
    OBJS="$OBJS $C/_build/v_$NAME/$T.o"
  else
    OBJS="$OBJS $C/_build/$T.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/wave_tracer_amd/_v/libwtgpu_$NAME.so $OBJS $C/_build/scene_builder.o $C/_build/scenes.o \
  $C/_build/xml_scene.o $C/_build/ply_loader.o $C/_build/obj_loader.o $C/_build/spectrum_db.o $C/_build/png_loader.o $C/_build/exr_loader.o -L/opt/rocm/lib -lrccl -lz -Wl,-rpath,/opt/rocm/lib
echo built $R/wave_tracer_amd/_v/libwtgpu_$NAME.so
