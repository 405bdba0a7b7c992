#!/bin/bash
# End-of-round artefacts of the build in the tree (no source may change between this run and the commit: pmc_traffic.json is stamped with the library hash):
# PMC passes of all five bench workloads first (each under its own key), then the bench lines that read them, then the kernel stats.
#   usage (via gpurun): bash scripts/final_measure.sh <tag>           copy gpurun_out/<tag>/{pmc_traffic.json,bench*.json,...} into profiles/ afterwards
TAG=${1:-final}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
rm -f profiles/pmc_traffic.json
# (keys "auto": what bench.py calls the workload — config, camera, scene footprint; the moving-camera passes cover the bench's own frames, scripts/pmc.sh)
bash scripts/pmc.sh $TAG/pmc auto > $O/pmc_config4.log 2>&1; tail -2 $O/pmc_config4.log
bash scripts/pmc.sh $TAG/pmc_lite auto --scene-footprint lite > $O/pmc_config4_lite.log 2>&1
bash scripts/pmc.sh $TAG/pmc_moving auto --moving-camera > $O/pmc_config4_moving.log 2>&1
bash scripts/pmc.sh $TAG/pmc2 auto --config 2 > $O/pmc_config2.log 2>&1
bash scripts/pmc.sh $TAG/pmc3 auto --config 3 > $O/pmc_config3.log 2>&1
bash scripts/pmc.sh $TAG/pmc5 auto --config 5 > $O/pmc_config5.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench2.err; cut -c1-300 $O/bench_driver_flags.json
timeout 900 python bench.py > $O/bench_again.json 2> $O/bench_again.err      # (two back-to-back lines on one box: the cpu_baseline has to repeat within 10 %)
timeout 900 python bench.py --scene-footprint lite > $O/bench_lite.json 2> $O/bench_lite.err; cut -c1-300 $O/bench_lite.json
timeout 600 python bench.py --config 2 > $O/bench_config2.json 2> $O/bench_c2.err
timeout 600 python bench.py --config 3 > $O/bench_config3.json 2> $O/bench_c3.err
timeout 900 python bench.py --config 5 > $O/bench_config5.json 2> $O/bench_c5.err
timeout 600 python bench.py --config 4 --moving-camera > $O/bench_config4_moving_camera.json 2> $O/bench_c4m.err
for f in bench bench_again bench_lite bench_config2 bench_config3 bench_config5 bench_config4_moving_camera; do python - $O/$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], "bound", r.get("bound"), "alg", r.get("frac_algorithmic"), "hbm", r.get("frac_hbm_counter"), "valu", r.get("frac_valu"))
PY
done
cd /tmp && export TMPDIR=/tmp
# (the profiled command decides its stream priorities like the bench line does: at its first frame)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o stats -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --profile-run > $O/prof_bench.json 2> $O/prof.err
ls $O/prof | head
