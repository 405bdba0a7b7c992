#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03w}; mkdir -p $O
cd $R
for nw in 8 6 4 3; do
  echo "== RESTIR_LAT_WAVES=$nw"
  RESTIR_LAT_WAVES=$nw timeout 900 python scripts/band_ab.py 544 560 496 528 528 576 464 528 > $O/band_ab_nw$nw.txt 2>&1; grep "rows" $O/band_ab_nw$nw.txt
done
