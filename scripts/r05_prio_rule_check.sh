#!/bin/bash
# round 5: the first-frame rule against every explicit setting, each in a fresh process (bench.py timed region, 100 frames)
TAG=${1:-r05_rule}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_stream_prio.py -x -q 2>&1 | tail -2 | head -1 | tee -a $O/check.txt
for w in "real --scene-footprint real" "lite --scene-footprint lite" "config3 --config 3" "config5 --config 5" "moving --moving-camera"; do
  set -- $w; name=$1; shift
  line="$name:"
  for p in "1,0" "1,1" "0,1" "0,-1" "1,-1"; do
    ms=$(timeout 600 python bench.py --no-cpu-baseline --profile-run --stream-priorities=$p "$@" 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    line="$line ($p) $ms"
  done
  auto=$(timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stream_priorities']['chosen'], d['stream_priorities']['filter_share'])")
  echo "$line | auto $auto" | tee -a $O/check.txt
done
