#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03u}; mkdir -p $O
cd $R
timeout 900 python scripts/band_ab.py 496 512 512 528 528 544 544 560 560 576 > $O/band16.txt 2>&1; grep rows $O/band16.txt
