#!/bin/bash
# End-of-round artefacts of the build in the tree (no source may change between this run and the commit: pmc_traffic.json is stamped with the library hash):
# bench line at the default and at the driver's flags, rocprofv3 kernel stats, PMC passes.   usage (via gpurun): bash scripts/final_measure.sh <tag>
TAG=${1:-final}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
bash scripts/pmc.sh $TAG/pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log
cp $O/pmc/pmc_traffic.json profiles/pmc_traffic.json      # so that the bench lines below read the counters of THIS build
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench2.err; cut -c1-300 $O/bench_driver_flags.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o stats -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --profile-run > $O/prof_bench.json 2> $O/prof.err
ls $O/prof | head
