#!/bin/bash
# same-box A/B: product build vs csrc/_ab/librestir_hip_<variant>.so ; usage: bash scripts/r03_ab.sh <tag> <variant> [repeat]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03ab}; mkdir -p $O
cd $R
V=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_$2.so
for i in $(seq 1 ${3:-2}); do
  bash scripts/ab_libs.sh "product|RESTIR_X=0" "$2|RESTIR_HIP_LIB=$V" | tee -a $O/ab_$2.txt
done
