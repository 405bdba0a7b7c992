#!/bin/bash
# Experiment (round 4): what would clipping alpha-masked geometry to its opaque part buy?  The leaf quads of the benchmark scene as n x n sub-quads without
# the all-transparent ones (RESTIR_SCENE_TESS_LEAVES, a different scene) against the product scene and against opaque leaves.  usage (gpurun): bash scripts/leaf_tess_ab.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r04tess}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
for V in "base|X=1" "tess4|RESTIR_SCENE_TESS_LEAVES=4" "tess8|RESTIR_SCENE_TESS_LEAVES=8" "opaque|RESTIR_DEBUG_OPAQUE_LEAVES=1"; do
  L="${V%%|*}"; E="${V#*|}"
  echo "==== $L"
  env $E RESTIR_OVERLAP=0 timeout 600 python scripts/gpu_perf.py bistro 2>&1 | grep '"case"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('serial', d['accel'], 'build_s', d['build_s'], 'wall', d['wall_ms'], d['stage_ms'], 'nodes/ray', d['nodes_per_ray'], 'tris/ray', d['tris_per_ray'])"
  env $E timeout 600 python scripts/gpu_perf.py bistro 2>&1 | grep '"case"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('in flight wall', d['wall_ms'], 'Mrays/s', d['Mrays_s'])"
  env $E timeout 900 python scripts/band_ab.py 496 528 528 576 256 368 > $O/band_$L.txt 2>&1; grep rows $O/band_$L.txt
  if [ "$L" == "base" ] || [ "$L" == "tess4" ]; then
    env $E timeout 900 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/emulate8_$L.json 2> $O/e_$L.err
    python - $O/emulate8_$L.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("emu8 one", d["single_gpu_serial_ms"], d["single_gpu_frames_in_flight_ms"], "| serial", d["rank_ms"], "slowest", d["slowest_rank_ms"])
print("emu8 period", d.get("rank_period_ms"), "slowest", d.get("slowest_rank_period_ms"))
PY
  fi
done
