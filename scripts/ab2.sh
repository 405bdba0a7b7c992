#!/bin/bash
# parity (fused subset + frames-in-flight), then perf under RESTIR_OVERLAP variants
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "${ABK:-fused or back_to_back or in_flight}" 2>&1 | tail -6
for OV in "$@"; do
  echo "== RESTIR_OVERLAP=$OV"
  RESTIR_OVERLAP=$OV timeout 600 python scripts/gpu_perf.py sponza bistro 2>&1 | grep '"case"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'], d['wall_ms'], d['event_ms'], d['stage_ms'], d['Mrays_s'])"
done
