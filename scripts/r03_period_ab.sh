#!/bin/bash
# A/B of the frames-in-flight period of selected emulated ranks under environment variants: bash scripts/r03_period_ab.sh <tag> "ENV=a ENV=b" "ENV=c" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03l}; mkdir -p $O; shift
cd $R
export RESTIR_EMULATE_RANKS=${RANKS:-1,4,6}
i=0
for v in "$@"; do
  i=$((i+1))
  ( for e in $v; do [ "$e" != "none" ] && export $e; done
    timeout 900 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/v$i.json 2> $O/v$i.err
    python - "$v" $O/v$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "| serial rank_ms", d["rank_ms"], "| period", d.get("rank_period_ms"))
PY
  )
done
