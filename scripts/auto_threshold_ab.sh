#!/bin/bash
# AUTO threshold of the direct stage (tiles per launch up to which the latency kernel runs) on the 8-rank emulation
R=$GRAFT_REPO_ROOT; T=${1:-r03thr}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
for t in ${TS:-512 1024 1440 2048}; do echo "== RESTIR_LAT_TILES=$t"; bash scripts/emulate8.sh $T RESTIR_LAT_TILES=$t; cp $O/emulate8_1080p.json $O/emulate8_t$t.json; done
