#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03y}; mkdir -p $O
cd $R
export RESTIR_EMULATE_RANKS=0,1
for g in 1 0; do
RESTIR_EMULATE_GATHER=$g timeout 900 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/g$g.json 2> $O/e.err
python - $O/g$g.json $g <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("gather", sys.argv[2], "period ranks 0,1:", d.get("rank_period_ms"), "bands", d["bands_last_frame"][:2])
PY
done
