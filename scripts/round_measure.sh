#!/bin/bash
# One measurement pass on the GPU box: GPU tests, demo app, bench line, rocprofv3 kernel stats, PMC passes.
# usage (via gpurun): bash scripts/round_measure.sh <tag>
TAG=${1:-r01c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 cis-565-final-vr-raytracer_amd/host/restir_demo -f tests/golden/mini_scene.gltf -w 640 -h 360 -n 8 -o $O/demo_mini > $O/demo.log 2>&1; tail -2 $O/demo.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o stats -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --profile-run > $O/prof_bench.json 2> $O/prof.err
cat $O/prof_bench.json
ls $O/prof
cd $R && timeout 1200 bash scripts/pmc.sh $TAG/pmc > $O/pmc.log 2>&1; tail -40 $O/pmc.log
