#!/bin/bash
# perf of the build in the tree (already compiled): serial + pipelined frame of the benchmark scene and the 4K interior scene, N times
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-2}); do
  for ov in 0 2; do env RESTIR_OVERLAP=$ov timeout 600 python scripts/gpu_perf.py bistro interior 2>&1 | grep '"case"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'], d['wall_ms'], d['stage_ms'])"; done
done
