#!/bin/bash
# A/B inside one gpurun call: parity subset on the new lib, then perf of each lib given as argument (default lib = "")
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${ABK:-fused}" 2>&1 | tail -4
for L in "$@"; do
  echo "== $L"
  if [ "$L" = "default" ]; then unset RESTIR_HIP_LIB; else export RESTIR_HIP_LIB=$GRAFT_REPO_ROOT/$L; fi
  timeout 600 python scripts/gpu_perf.py sponza bistro 2>&1 | grep '"case"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'], d['wall_ms'], d['stage_ms'], d['Mrays_s'], d['nodes_per_ray'], d['tris_per_ray'], d['closest'], d['any'])"
done
