#!/bin/bash
# same-band A/B with the product build + wide-kernel profile with the measurement build; $1 = tag; rest: env
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03h}; mkdir -p $O; shift
cd $R
for e in "$@"; do export $e; done
timeout 900 python scripts/band_ab.py 496 528 528 576 256 368 800 1080 > $O/band_ab.txt 2>&1; grep "rows\|rror" $O/band_ab.txt
export RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_prof.so
RESTIR_LAT=1 timeout 600 python scripts/wave_profile.py 496 528 528 576 > $O/wave_profile.txt 2>&1
grep -A6 "wide build\|mean workgroup\|percentiles" $O/wave_profile.txt | grep -v "^--" | head -60
