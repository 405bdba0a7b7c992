#!/bin/bash
# (ran with tileOfChunk() / RT_TILE_BLOCK_Y in csrc/stage_common.h, removed after the measurement: profiles/r04_tile_blocks_ab.txt)
# XCD tile order: 2-D blocks of ceil(tilesX / 8) x BY tiles per XCD (-DRT_TILE_BLOCK_Y=BY) against whole tile rows (round 4).
# usage (gpurun): bash scripts/tile_blocks_ab.sh <tag> [BY ...]
R=$GRAFT_REPO_ROOT; T=${1:-r04blk}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; shift
V=("rows|-|-")
for by in ${@:-8 16 32}; do V+=("by$by|-DRT_TILE_BLOCK_Y=$by|-"); done
V+=("rows_again|-|-")
LAT=1 bash scripts/variants_bench.sh $T "${V[@]}" | tee $O/ab.txt
by=${1:-8}
RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_by$by.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_digests.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -1 | tee $O/parity.txt
