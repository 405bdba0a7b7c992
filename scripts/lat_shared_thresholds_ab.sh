#!/bin/bash
# RT_TRAVERSAL_AUTO thresholds for launches that share the chip (frames in flight inside a rank), with gang mode: 8-rank emulation
R=$GRAFT_REPO_ROOT; T=${1:-r04thr}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
for V in "512 640" "768 960" "1024 1280" "1440 1536" "256 320"; do
  set -- $V
  for S in 1080p; do
    RESTIR_LAT_TILES_SHARED=$1 RESTIR_LAT_TILES_IND_SHARED=$2 timeout 1200 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/emu8_$1_$2.json 2> $O/e.err
    python - $O/emu8_$1_$2.json $1 $2 <<'PY'
import json,sys,statistics
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p=d.get("rank_period_ms") or [0]
print("shared thresholds", sys.argv[2], sys.argv[3], "serial slowest", d["slowest_rank_ms"], "| period slowest", d.get("slowest_rank_period_ms"), "median", round(statistics.median(p),3), p)
PY
  done
done
