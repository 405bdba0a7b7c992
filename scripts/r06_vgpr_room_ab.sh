#!/bin/bash
# round 6: do the traced kernels leave the filter chain no room?  5 waves x 96 VGPRs = 480 of a SIMD's 512 — a k_denoise_lds wave (50-57) cannot become resident beside them.
# Variants with the traced kernels capped at 88 registers (5 x 88 + 56 = 496): csrc/_vg/librestir_hip_v<direct><indirect>.so, built by the caller
# (restir_amd.build.build_hip(variant=..., extra_flags=["-DRT_DIRECT_VGPR=88", ...])).   usage (gpurun): bash scripts/r06_vgpr_room_ab.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r06_vgpr}
V=$R/cis-565-final-vr-raytracer_amd/csrc/_vg
cd $R
REPS=2 bash scripts/ab_libs2.sh $T/real "base9696|-" "v8888|RESTIR_HIP_LIB=$V/librestir_hip_v8888.so" "v8896|RESTIR_HIP_LIB=$V/librestir_hip_v8896.so" "v9688|RESTIR_HIP_LIB=$V/librestir_hip_v9688.so"
echo "== three frames in flight (the filter chain off the loop)"
REPS=1 BENCH_ARGS="" bash scripts/ab_libs2.sh $T/real3 "base9696_mode3|RESTIR_OVERLAP=3" "v8888_mode3|RESTIR_OVERLAP=3 RESTIR_HIP_LIB=$V/librestir_hip_v8888.so"
echo "== config 3"
REPS=1 BENCH_ARGS="--config 3" bash scripts/ab_libs2.sh $T/c3 "base9696|-" "v8888|RESTIR_HIP_LIB=$V/librestir_hip_v8888.so" "v8896|RESTIR_HIP_LIB=$V/librestir_hip_v8896.so"
echo "== lite"
REPS=1 BENCH_ARGS="--scene-footprint lite" bash scripts/ab_libs2.sh $T/lite "base9696|-" "v8888|RESTIR_HIP_LIB=$V/librestir_hip_v8888.so"
