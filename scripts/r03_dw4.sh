#!/bin/bash
# latency direct kernel capped at 128 VGPRs (two workgroups per CU) vs the product (151 VGPRs, one workgroup per CU)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03dw4}; mkdir -p $O
cd $R
B="496 512 544 560 496 528 528 576 464 528 256 368"
echo "== product"; timeout 900 python scripts/band_ab.py $B > $O/band_prod.txt 2>&1; grep rows $O/band_prod.txt
echo "== 128 VGPRs"; RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_dw4.so timeout 900 python scripts/band_ab.py $B > $O/band_dw4.txt 2>&1; grep rows $O/band_dw4.txt
