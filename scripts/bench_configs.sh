#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03p}; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 10 > $O/bench_c4.json 2> $O/b4.err; tail -c 1500 $O/bench_c4.json; tail -2 $O/b4.err
for c in 2 3 5; do timeout 1200 python bench.py --config $c --steps 30 --warmup 10 > $O/bench_c$c.json 2> $O/b$c.err; tail -c 900 $O/bench_c$c.json; tail -2 $O/b$c.err; done
timeout 900 python bench.py --config 4 --moving-camera --steps 60 --warmup 10 > $O/bench_c4_moving.json 2> $O/b4m.err; tail -c 600 $O/bench_c4_moving.json; tail -2 $O/b4m.err
