#!/bin/bash
# Compiles the pieces of the reference that build from their own source files with plain g++ — in the
# authoring container only (/root/reference is absent on the GPU box).  Output: oracle/_ref/kat_ref (git-ignored binary).
# The reference's hot path itself (GLSL 460 + ray query + nvpro_core + Vulkan) is NOT buildable here; these are only the
# integer / host-side helpers that pin the oracle's bit-exact pieces.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=/root/reference
[ -d "$REF" ] || { echo "no reference checkout: nothing to build"; exit 0; }
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
# excerpts of the GLSL, with GLSL parameter qualifiers rewritten to C++ (inout T x -> T& x, in T x -> T x)
rewrite() { sed -E 's/\binout ([a-zA-Z0-9_]+) /\1\& /g; s/\bin ([a-zA-Z0-9_]+) /\1 /g'; }
{ sed -n '34,48p;59,65p;98,102p' "$REF/shaders/random.glsl" | rewrite; } > "$TMP/ref_random.inc"
{ sed -n '98,115p;141,143p' "$REF/shaders/common.glsl" | rewrite; } > "$TMP/ref_common.inc"
mkdir -p "$HERE/../_ref"
g++ -O1 -std=c++17 -ffp-contract=off -I"$TMP" -I"$REF/shaders" -I"$REF/src" "$HERE/kat_main.cpp" -o "$HERE/../_ref/kat_ref"
echo "built oracle/_ref/kat_ref"
# ---- float helpers (BSDF, reservoirs, disk mapping): GLSL excerpts on a vector shim (kat_float.cpp) ----
# out/inout/in qualifiers -> C++ references, unsuffixed float literals get an `f` (GLSL literals are single precision),
# the `.xy` swizzle of a vec3 is spelled out.
rewritef() { sed -E 's/\b(inout|out) ([a-zA-Z0-9_]+) /\2\& /g; s/\bin ([a-zA-Z0-9_]+) /\1 /g; s/\b([a-z]+)\.xy\b/vec2(\1.x, \1.y)/g' \
             | perl -pe 's/(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)(?![\w.])/$1f/g'; }
sed -n '37,41p' "$REF/shaders/globals.glsl" | rewritef > "$TMP/ref_globals.inc"
sed -n '28p;171,180p;194,200p' "$REF/shaders/common.glsl" | rewritef > "$TMP/ref_common2.inc"
sed -n '68,92p' "$REF/shaders/common.glsl" | rewritef > "$TMP/ref_common3.inc"
sed -n '23,25p' "$REF/shaders/denoise_common.glsl" | rewritef > "$TMP/ref_lum.inc"
sed -n '260,284p' "$REF/shaders/host_device.h" | rewritef > "$TMP/ref_structs.inc"
sed -n '6,128p' "$REF/shaders/reservoir.glsl" | rewritef > "$TMP/ref_reservoir.inc"
sed -n '8,173p' "$REF/shaders/pbr_metallicworkflow.glsl" | rewritef > "$TMP/ref_pbr.inc"
sed -n '81,92p' "$REF/shaders/random.glsl" | rewritef > "$TMP/ref_pcg3d.inc"
sed -n '24,25p;29,39p;48,65p' "$REF/shaders/tonemapping.glsl" | rewritef | sed -E 's/srgbIn\.xyz/srgbIn/' > "$TMP/ref_tonemap.inc"
sed -n '336,351p' "$REF/shaders/host_device.h" | rewritef > "$TMP/ref_tmstruct.inc"
sed -n '50,68p' "$REF/shaders/post.frag" | rewritef > "$TMP/ref_post.inc"
sed -n '353,376p' "$REF/shaders/host_device.h" | rewritef > "$TMP/ref_skystruct.inc"
sed -n '25,602p' "$REF/shaders/sun_and_sky.glsl" | rewritef > "$TMP/ref_sky.inc"
g++ -O1 -std=c++17 -ffp-contract=off -I"$TMP" "$HERE/kat_float.cpp" -o "$HERE/../_ref/kat_float"
echo "built oracle/_ref/kat_float"
