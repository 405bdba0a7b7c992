#!/bin/bash
# round 5 (verdict item 1): round 4's questions asked again on the `real` footprint exterior scene, one box:
#   (1) 4 vs 5 waves per SIMD of the direct stage (the spill trade), (2) what the alpha test costs (opaque cards = the bound of ANY micro-map refinement)
#   and how the candidates resolve (micro-map vs texture, cycles per phase of a triangle step: measurement build)
TAG=${1:-r05_reask}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export BENCH_ARGS="--scene-footprint real --stream-priorities default"
bash scripts/variants_bench.sh $TAG "lb55|-|-" "lb45|-DRT_DIRECT_LB=4|-" "lb55_again|-|-" "lb54|-DRT_INDIRECT_LB=4|-" "opaque_cards|-|RESTIR_DEBUG_OPAQUE_LEAVES=1" "split_a5|-|RESTIR_BVH_SPLIT=1 RESTIR_BVH_SPLIT_ALPHA=1e-5" "split_a5_lb45|-DRT_DIRECT_LB=4|RESTIR_BVH_SPLIT=1 RESTIR_BVH_SPLIT_ALPHA=1e-5" 2>&1 | tee $O/variants.txt
python -c "import restir_amd; from restir_amd import build; build.build_hip(variant='prof', extra_flags=['-DRT_WAVEPROF=1'])" > /dev/null 2>&1
export RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_prof.so
for kind in PROC_BISTRO_EXT_REAL PROC_BISTRO_EXT; do
  echo "==== $kind" | tee -a $O/wave.txt
  WAVE_PROFILE_KIND=$kind WAVE_PROFILE_LAT=1 timeout 900 python scripts/wave_profile.py 496 528 528 576 > $O/wave_$kind.txt 2>&1
  grep -E "^== |all waves|wave time|slowest wave:|alpha" $O/wave_$kind.txt | tee -a $O/wave.txt
done
