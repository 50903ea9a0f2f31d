#!/bin/bash
# CPU only: the physics the GPU runs — wave_tracer_amd/csrc/wt/*.h, compiled here into the CPU checker — and the host scene code under
# AddressSanitizer + UndefinedBehaviorSanitizer (+ float-cast-overflow: a float -> int conversion out of range is undefined, and x86 and gfx950
# resolve it differently), over whole renders of the bundled scenes: both integrators, Fraunhofer and UTD diffraction, the polarimetric film,
# textures, wrappers, the split-step and staged-connection flavours.  An out-of-bounds read in a shared header would be one on the device as well.
# `msan`: the same renders under MemorySanitizer instead (ROCm's clang; only the checker and the wt/ headers are instrumented and heap memory counts as
# initialised — libstdc++ is not instrumented — so what it finds are uninitialised STACK values: a local or a struct member read before it is written,
# the kind of defect that makes two compilers of the same header disagree).
# usage: tools/sanitize_checker.sh [quick] [msan] [only=<substring of a configuration>]        (round 5: 38 renders each way, no report — profiles/r05_sanitized_checker.log)
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/wave_tracer_amd/csrc; D=$(mktemp -d /tmp/wtgpu_san_XXXX)
cat > $D/main.cpp <<'CPP'
#include "host/scene_builder.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
extern "C" int oracle_render(const void* scene_host, uint64_t sample_begin, uint64_t sample_end, uint64_t seed, double* value, double* weight, double* light, int threads, unsigned long long* counters);
extern "C" int oracle_counters_count();
extern "C" void oracle_set_split_step(int);
extern "C" void oracle_set_staged_connect(int);
int main(int argc, char** argv) {   // name res spp [key=value ...]
    wth::scene_params_t p{};
    p.res = (uint32_t)atoi(argv[2]);
    const int spp = atoi(argv[3]);
    p.max_depth = p.fsd = p.mis = p.rr = -1; p.mesh_detail = 0; p.lut_n_theta = 32; p.lut_m = 32; p.polarimetric = -1;
    int flavour = 0;
    std::vector<std::string> defines;   // -Dname=value: the defines of a scene FILE (argv[1] ends in .xml)
    for (int i = 4; i < argc; ++i) {
        if (!strncmp(argv[i], "-D", 2)) { defines.push_back(argv[i] + 2); continue; }
        const char* e = strchr(argv[i], '=');
        const std::string k(argv[i], e - argv[i]);
        const int v = atoi(e + 1);
        if (k == "fsd") p.fsd = v; else if (k == "mesh_detail") p.mesh_detail = v; else if (k == "polarimetric") p.polarimetric = v; else if (k == "max_depth") p.max_depth = v;
        else if (k == "rr") p.rr = v; else if (k == "crop_of") p.crop_of = (uint32_t)v; else if (k == "flavour") flavour = v; else if (k == "lut") p.lut_n_theta = p.lut_m = (uint32_t)v;
    }
    wth::scene_builder_t b;
    const std::string name = argv[1];
    if (name.size() > 4 && name.substr(name.size() - 4) == ".xml") {
        try { wth::build_scene_from_xml(name, defines, p, b); } catch (const std::exception& e) { std::printf("%s: %s\n", argv[1], e.what()); return 2; }
    } else
    if (!wth::build_named_scene(argv[1], p, b)) { std::printf("unknown scene %s\n", argv[1]); return 2; }
    const wt::scene_t sc = b.scene();
    const size_t n = (size_t)sc.sensor.height * sc.sensor.width;
    std::vector<double> v(n * 16, 0.0), w(n, 0.0), l(n * 16, 0.0);
    std::vector<unsigned long long> ctr((size_t)oracle_counters_count(), 0);
    if (flavour & 1) oracle_set_split_step(1);
    if (flavour & 2) oracle_set_staged_connect(1);
    const int rc = oracle_render(&sc, 0, (uint64_t)spp, 7, v.data(), w.data(), l.data(), 4, ctr.data());
    double s = 0;
    for (double x : v) s += x;
    for (double x : l) s += x;
    std::printf("rc %d film sum %.6g segments %llu\n", rc, s, ctr[0]);
    return rc;
}
CPP
SRCS="$D/main.cpp $R/oracle/oracle.cpp host/scene_builder.cpp host/scenes.cpp host/xml_scene.cpp host/ply_loader.cpp host/obj_loader.cpp host/spectrum_db.cpp host/png_loader.cpp host/exr_loader.cpp"
if [[ " $* " == *" msan "* ]]; then
  WHAT="MemorySanitizer (stack values)"
  printf 'src:*host/*\nsrc:*main.cpp\n' > $D/ignore.txt
  ( cd $C && /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=memory -fsanitize-memory-track-origins=1 -fsanitize-ignorelist=$D/ignore.txt -fno-omit-frame-pointer \
      -ffp-contract=off -fno-strict-aliasing -pthread -Wno-everything -DWT_ORACLE_UNBOUNDED -I. -o $D/render $SRCS -lz )
  export MSAN_OPTIONS=poison_in_malloc=0:poison_in_free=0
else
  WHAT="ASan + UBSan + float-cast-overflow"
  ( cd $C && g++ -O1 -g -std=c++17 -fsanitize=address,undefined,float-cast-overflow -fno-sanitize=float-divide-by-zero -ffp-contract=off -fno-strict-aliasing -pthread \
      -Wno-unknown-pragmas -DWT_ORACLE_UNBOUNDED -I. -o $D/render $SRCS -lz )
fi
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 ASAN_OPTIONS=detect_leaks=0
CFGS=("bidir_room 24 2 polarimetric=1" "bidir_room 24 2" "cornell_box 32 4 mesh_detail=1" "cornell_box 32 8 mesh_detail=1 crop_of=1440 lut=64" "cornell_box_path 24 4"
 "double_slits 48 4" "double_slits_overview 32 2" "etoile 32 8" "etoile 24 4 mesh_detail=1" "etoile_bdpt 24 4" "etoile_open 24 4" "etoile_path_backward 24 4"
 "furnace 16 8 max_depth=32 rr=0" "furnace 16 4 fsd=1" "furnace 16 4 fsd=1 flavour=1" "furnace 16 4 fsd=1 flavour=2" "cornell_box 24 4 flavour=3" "furnace_path 16 8" "furnace_spm 16 8"
 "furnace_wall_mask 16 8" "furnace_wall_composite 16 4" "furnace_wall_step_gap 16 4" "lens_a 24 4" "lens_b 24 4" "lens_c 24 4" "sunlit 24 8" "sunlit_path 24 8"
 "white_furnace 16 8" "white_furnace_path 16 8" "tex_checker 24 4" "tex_bitmap 24 4" "tex_normal_tilt 24 4" "tex_mask 24 4" "tex_bilinear_ramp 24 4"
 "$R/tests/data/xml/textured_emitter.xml 24 8" "$R/tests/data/xml/textured_emitter.xml 24 8 -Dfilter=bicubic -Dmscale=2" "$R/tests/data/xml/textured_emitter.xml 24 8 -Dintegrator=plt_path -Ddirection=backward"
 "$R/tests/data/xml/textured_emitter.xml 24 16 -Dintegrator=plt_path -Ddirection=forward -Dplane=true")
# (round 6: the area emitter with a bitmap radiance — per-triangle texel tables indexed from random numbers and from barycentrics)
# (MemorySanitizer: named scenes only — the XML reader lives in the uninstrumented host code with libstdc++'s strings and maps, which MSan reports on falsely)
if [[ " $* " == *" msan "* ]]; then NEW=(); for c in "${CFGS[@]}"; do [[ "$c" == *".xml "* ]] || NEW+=("$c"); done; CFGS=("${NEW[@]}"); fi
if [[ " $* " == *" only="* ]]; then ONLY=$(echo " $* " | sed 's/.* only=\([^ ]*\) .*/\1/'); NEW=(); for c in "${CFGS[@]}"; do [[ "$c" == *"$ONLY"* ]] && NEW+=("$c"); done; CFGS=("${NEW[@]}"); fi
[[ " $* " == *" quick "* ]] || CFGS+=("cornell_box 32 48 mesh_detail=1 crop_of=1440 lut=128" "bidir_room 48 8 polarimetric=1 mesh_detail=1" "etoile 48 16 mesh_detail=2" "double_slits 96 8 lut=128")
BAD=0
for cfg in "${CFGS[@]}"; do
  OUT=$($D/render $cfg 2>&1 | grep -v "^wtgpu:" | tail -12); RC=$?
  if echo "$OUT" | grep -q "^rc 0" && ! echo "$OUT" | grep -q "ERROR\|runtime error\|WARNING: MemorySanitizer"; then printf "%-64s %s\n" "$cfg" "$(echo "$OUT" | tail -1)"; else BAD=$((BAD+1)); echo "$cfg: REPORT"; echo "$OUT"; fi
done
echo "${#CFGS[@]} renders under $WHAT: $BAD with a report"
# The known-answer tests against a sanitized build of the checker library (round 5: 128 tests, no report):
#   g++ -std=c++17 -O1 -g -fsanitize=address,undefined,float-cast-overflow -fno-sanitize=float-divide-by-zero -march=x86-64-v3 -ffp-contract=off -fno-strict-aliasing \
#       -fPIC -shared -pthread -Wno-unknown-pragmas -DWT_ORACLE_UNBOUNDED -o /tmp/liboracle_san.so oracle/oracle.cpp oracle/kat.cpp
#   LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 WT_ORACLE_LIB=/tmp/liboracle_san.so \
#       python -m pytest tests/test_kat*.py tests/test_oracle.py tests/test_polarimetric.py tests/test_wrappers.py tests/test_textures.py tests/test_emitters.py -m "not gpu" -s 2>&1 | grep -c "runtime error"
exit $BAD
