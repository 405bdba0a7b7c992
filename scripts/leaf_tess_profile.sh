#!/bin/bash
# per-wave profile (alpha resolution, cycles per round) of the tessellated-leaf experiment; usage (gpurun): bash scripts/leaf_tess_profile.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r04tess}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
# the measurement build does not travel (csrc/_ab is in .gpurunignore): built here, ~25 s
python -c "import restir_amd; from restir_amd import build; build.build_hip(variant='prof', extra_flags=['-DRT_WAVEPROF=1'])" > /dev/null 2>&1
export RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_prof.so
for V in "base|X=1" "tess4|RESTIR_SCENE_TESS_LEAVES=4" "tess8|RESTIR_SCENE_TESS_LEAVES=8"; do
  L="${V%%|*}"; E="${V#*|}"
  for M in 0 1; do
    echo "==== $L lat=$M"
    env $E WAVE_PROFILE_LAT=$M timeout 600 python scripts/wave_profile.py 496 528 528 576 > $O/wave_${L}_lat$M.txt 2>&1
    grep -E "^== |all waves|wave time|slowest wave:" $O/wave_${L}_lat$M.txt
  done
done
