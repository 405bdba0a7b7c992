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
