#!/bin/bash
# round 5: is the load-time tuner's ranking the ranking of the timed region?  Every candidate setting timed by bench.py itself (100 frames) next to what the tuner measured.
TAG=${1:-r05_tuner}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for w in "real --scene-footprint real" "config3 --config 3" "config5 --config 5" "lite --scene-footprint lite"; do
  set -- $w; name=$1; shift
  for p in "1,0" "1,1" "0,1" "0,-1" "1,-1"; do
    ms=$(timeout 600 python bench.py --no-cpu-baseline --profile-run --stream-priorities=$p "$@" 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$name explicit $p timed $ms" | tee -a $O/check.txt
  done
  timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name tuned: timed', d['ms_per_step'], d['stream_priorities'])
" | tee -a $O/check.txt
done
