#!/bin/bash
# round 5 (verdict item 4): order the LDS ray pool of the indirect stage before the waves pull from it.  -DRT_POOL_ORDER bit 0: the work list of the K-tile
# single-bounce pools ordered by direction octant; bit 1: shadow rays before bounce rays in the mixed pool of the multi-bounce tiles.  Measurement builds, one box.
TAG=${1:-r05_pool}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for fp in real lite; do
  echo "==== footprint $fp" | tee -a $O/ab.txt
  BENCH_ARGS="--scene-footprint $fp --stream-priorities default" bash scripts/variants_bench.sh $TAG/$fp "base|-|-" "octant|-DRT_POOL_ORDER=1|-" "shadow_first|-DRT_POOL_ORDER=2|-" "both|-DRT_POOL_ORDER=3|-" "base_again|-|-" 2>&1 | tee -a $O/ab.txt
done
