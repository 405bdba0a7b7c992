#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03d}; mkdir -p $O; shift
cd $R
for e in "$@"; do export $e; done
timeout 900 python scripts/band_ab.py > $O/band_ab.txt 2>&1; grep rows $O/band_ab.txt
