#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  echo "== $V"
  env $V timeout 600 python scripts/gpu_perf.py sponza bistro 2>&1 | grep '"case"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'], d['wall_ms'], d['event_ms'], d['stage_ms'], d['Mrays_s'])"
done
