#!/bin/bash
# Round 6, item 2 (the rest of its "done" list): cycles per traversal round of the latency build and of the throughput build with typed and with generic accessors
# (-DRT_GENERIC_AS=1 = the code of rounds 1-5), measurement builds (-DRT_WAVEPROF=1) made on the box; horizon bands of the benchmark frame, real and lite scene.
R=$GRAFT_REPO_ROOT; T=${1:-r06as}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
python -c "import restir_amd; from restir_amd import build; build.build_hip(variant='prof', extra_flags=['-DRT_WAVEPROF=1']); build.build_hip(variant='profgen', extra_flags=['-DRT_WAVEPROF=1', '-DRT_GENERIC_AS=1'])" > $O/build.log 2>&1
for kind in PROC_BISTRO_EXT_REAL PROC_BISTRO_EXT; do for lat in 1 0; do for v in prof profgen; do
  echo "== $kind  latency build=$lat  $v" | tee -a $O/summary.txt
  WAVE_PROFILE_KIND=$kind WAVE_PROFILE_LAT=$lat RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_$v.so timeout 600 python scripts/wave_profile.py 496 528 528 576 > $O/wave_${kind}_${lat}_$v.txt 2>&1
  grep -i "^####\|^== \|cycles per round\|node step, cycles\|triangle step, cycles" $O/wave_${kind}_${lat}_$v.txt | cut -c1-260 | tee -a $O/summary.txt
done; done; done
