#!/bin/bash
# small-launch tile order A/B: product (chunks of 8 tiles for launches under 24 tile rows) vs whole tile rows per XCD
# (build the variant first: build_hip(variant="rows", extra_flags=["-DRT_TILE_SMALL_ROWS=0"]) -> csrc/_ab/librestir_hip_rows.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03v}; mkdir -p $O
cd $R
echo "== chunks of 8 tiles (product)"; timeout 900 python scripts/band_ab.py 496 512 528 544 544 560 496 528 528 576 256 368 > $O/band_chunk.txt 2>&1; grep rows $O/band_chunk.txt
echo "== whole tile rows per XCD"; RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_rows.so timeout 900 python scripts/band_ab.py 496 512 528 544 544 560 496 528 528 576 256 368 > $O/band_rows.txt 2>&1; grep rows $O/band_rows.txt
