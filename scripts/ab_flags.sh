#!/bin/bash
# A/B of extra hipcc flags (RESTIR_EXTRA_HIPFLAGS): rebuilds librestir_hip.so on the GPU box for every variant.  "none" = no extra flag.
cd $GRAFT_REPO_ROOT
for F in "$@"; do
  echo "== flags: $F"
  if [ "$F" = "none" ]; then export RESTIR_EXTRA_HIPFLAGS=""; else export RESTIR_EXTRA_HIPFLAGS="$F"; fi
  python -c "
import sys; sys.path.insert(0,'.')
from importlib import import_module
import_module('restir_amd.build').build_hip(force=True)" > /dev/null 2>&1 || { echo BUILD FAILED; continue; }
  for ov in 0 2; do env RESTIR_OVERLAP=$ov timeout 600 python scripts/gpu_perf.py sponza bistro 2>&1 | grep '"case"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'], d['wall_ms'], d['stage_ms'])"; done
done
