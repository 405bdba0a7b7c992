#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03j}; mkdir -p $O; shift
cd $R
for e in "$@"; do export $e; done
timeout 900 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/emulate8_1080p.json 2> $O/emulate8_1080p.err; cat $O/emulate8_1080p.json; tail -3 $O/emulate8_1080p.err
