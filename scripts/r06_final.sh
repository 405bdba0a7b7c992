#!/bin/bash
# round 6, end of round on ONE box: the whole GPU suite, the artefacts of scripts/final_measure.sh (counter passes first, then the bench lines that read them, then the
# kernel stats), the two extra poses of the headline scene with their own counter passes, the glTF path at BASELINE scale, where the seconds of a scene load go, and a
# randomised parity campaign — all on the build in the tree.   usage (gpurun): bash scripts/r06_final.sh <tag> [n1 n2]
TAG=${1:-r06z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu exit $?"; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
bash scripts/final_measure.sh $TAG > $O/final_measure.log 2>&1; tail -12 $O/final_measure.log | cut -c1-250
# the headline scene from the two other poses (bench.py --pose): counters over their own frames, then the lines
for p in 1 2; do
  bash scripts/pmc.sh $TAG/pmc_pose$p auto --pose $p > $O/pmc_pose$p.log 2>&1
  cp profiles/pmc_traffic.json $O/pmc_traffic.json
  timeout 900 python bench.py --pose $p > $O/bench_pose$p.json 2> $O/bench_pose$p.err
  python - $O/bench_pose$p.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], "bound", r.get("bound"), "alg", r.get("frac_algorithmic"), "hbm", r.get("frac_hbm_counter"), "valu", r.get("frac_valu"), d.get("stream_priorities"))
PY
done
RESTIR_BVH_TIMING=1 timeout 1500 python bench.py --via-gltf --no-cpu-baseline > $O/bench_via_gltf.json 2> $O/bench_via_gltf.err
python -c "
import json; d=json.loads(open('$O/bench_via_gltf.json').read().strip().splitlines()[-1]); v=d['via_gltf']; print('via gltf', d['ms_per_step'], {k: v[k] for k in ('files','bytes_on_disk','save_s','load_s','peak_rss_gb_after_load','scene_digest_equal','frames_equal')})"
grep "scene load\|bvh8 build" $O/bench_via_gltf.err | head -30
bash scripts/fuzz_campaign.sh $TAG/fuzz ${2:-300} ${3:-100} 61 2>&1 | cut -c1-200 | tail -6
