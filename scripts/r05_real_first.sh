#!/bin/bash
# round 5, first contact of the `real` footprint scene with the GPU: parity (small + full size), lite vs real bench lines on one box, counter pass of the real line
TAG=${1:-r05a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "real" > $O/pytest_small.log 2>&1; tail -3 $O/pytest_small.log
timeout 1500 python -m pytest tests/test_gpu_fullsize_allstages.py -x -q -k "real" > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log
timeout 900 python bench.py --scene-footprint lite > $O/bench_lite.json 2> $O/bench_lite.err; cut -c1-400 $O/bench_lite.json; tail -2 $O/bench_lite.err
timeout 900 python bench.py --scene-footprint real > $O/bench_real_nopmc.json 2> $O/bench_real.err; cut -c1-400 $O/bench_real_nopmc.json; tail -2 $O/bench_real.err
bash scripts/pmc.sh $TAG/pmc_real config4_real --scene-footprint real > $O/pmc_real.log 2>&1; tail -40 $O/pmc_real.log
bash scripts/pmc.sh $TAG/pmc_lite config4 > $O/pmc_lite.log 2>&1; tail -3 $O/pmc_lite.log
cp profiles/pmc_traffic.json $O/pmc_traffic.json
timeout 900 python bench.py --scene-footprint real > $O/bench_real.json 2> $O/bench_real2.err; cut -c1-300 $O/bench_real.json
timeout 900 python bench.py --scene-footprint lite > $O/bench_lite2.json 2> $O/bench_lite2.err; cut -c1-300 $O/bench_lite2.json
timeout 900 python bench.py --config 3 --scene-footprint real --no-cpu-baseline > $O/bench_c3_real.json 2> $O/bench_c3_real.err; cut -c1-300 $O/bench_c3_real.json
timeout 900 python bench.py --config 3 --scene-footprint lite --no-cpu-baseline > $O/bench_c3_lite.json 2> $O/bench_c3_lite.err; cut -c1-300 $O/bench_c3_lite.json
