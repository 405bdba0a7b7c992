#!/bin/bash
# per-wave profile of bands under the measurement build; $1 = tag, $2.. = env assignments
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03p}; mkdir -p $O; shift
cd $R
# the measurement build does not travel (csrc/_ab is in .gpurunignore): built here, ~25 s
python -c "import restir_amd; from restir_amd import build; build.build_hip(variant='prof', extra_flags=['-DRT_WAVEPROF=1'])" > /dev/null 2>&1
export RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_prof.so
for e in "$@"; do export $e; done
timeout 600 python scripts/wave_profile.py ${BANDS:-496 528 528 576} > $O/wave_profile.txt 2>&1
grep -v "^   (" $O/wave_profile.txt | tail -40
grep -A6 "slowest w" $O/wave_profile.txt | head -40
