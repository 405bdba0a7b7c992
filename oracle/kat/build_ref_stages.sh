#!/bin/bash
# Compiles the reference's OWN stage shaders (shaders/*.comp and everything they #include) for the CPU, from where they lie under
# /root/reference, into oracle/_ref/libref_stages.so — in the authoring container only (/root/reference is absent on the GPU box).
# glsl2cpp.py rewrites GLSL syntax C++ lacks into a scratch directory (rules in its header; no statement is changed) and
# glsl_cpu.h supplies the language runtime + the driver side (ray query over the oracle's flattened triangles, samplers).
# Used by tests/test_ref_stages.py to hold the oracle's stage-level restatement to the reference source, and by
# tests/golden/make_ref_stage_vectors.py to mint the per-pixel vectors that travel with the repository.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=/root/reference
[ -d "$REF/shaders" ] || { echo "no reference checkout: nothing to build"; exit 0; }
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
python3 "$HERE/glsl2cpp.py" "$REF/shaders" "$TMP/sh"
mkdir -p "$HERE/../_ref"
CXX=${CXX:-g++}
FLAGS="-O1 -std=c++20 -fPIC -mfma -ffp-contract=off -fno-fast-math -w -I$TMP/sh -I$HERE"
build_tu() {  # shader file, entry name, extra defines
  $CXX $FLAGS -DREF_SHADER="\"$1\"" -DREF_ENTRY="$2" $3 -c "$HERE/ref_tu.cpp" -o "$TMP/$2.o"
}
build_tu direct_stage.comp     ref_run_direct_stage     -DREF_HAS_PRD &
build_tu direct_gen.comp       ref_run_direct_gen       -DREF_HAS_PRD &
build_tu direct_reuse.comp     ref_run_direct_reuse     -DREF_HAS_PRD &
build_tu indirect_stage.comp   ref_run_indirect_stage   -DREF_HAS_PRD &
build_tu denoise_direct.comp   ref_run_denoise_direct   "" &
build_tu denoise_indirect.comp ref_run_denoise_indirect "" &
build_tu compose.comp          ref_run_compose          "" &
$CXX $FLAGS -c "$HERE/ref_frame.cpp" -o "$TMP/ref_frame.o" &
$CXX -O2 -std=c++17 -fPIC -mfma -ffp-contract=off -fno-fast-math -w -c "$HERE/../orc_scene.cpp" -o "$TMP/orc_scene.o" &
fail=0
for j in $(jobs -p); do wait "$j" || fail=1; done
[ "$fail" = 0 ] || { echo "build_ref_stages: a translation unit failed to compile"; exit 1; }
$CXX -shared -o "$HERE/../_ref/libref_stages.so" "$TMP"/*.o -pthread
echo "built oracle/_ref/libref_stages.so"
