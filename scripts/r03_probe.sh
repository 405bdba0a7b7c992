#!/bin/bash
# round 3, first GPU call: per-wave profile of the horizon bands (measurement build) + the emulate-world baseline of the product build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
export RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_prof.so
timeout 600 python scripts/wave_profile.py 496 528 528 576 240 384 > $O/wave_profile_1080p.txt 2>&1
unset RESTIR_HIP_LIB
tail -80 $O/wave_profile_1080p.txt
timeout 600 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/emulate8_1080p.json 2> $O/emulate8.err; cat $O/emulate8_1080p.json
