#!/bin/bash
# A/B of environment switches on the build in the tree: serial + pipelined frame (benchmark scene, 4K interior), each variant twice
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for V in "$@"; do
  echo "== $V"
  for ov in 0 2; do env $V RESTIR_OVERLAP=$ov timeout 600 python scripts/gpu_perf.py bistro 2>&1 | grep '"case"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'], d['wall_ms'], d['stage_ms'])"; done
done; done
