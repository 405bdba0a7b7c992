#!/bin/bash
# round 5: tile rows of the full-frame direct-stage launch in the order of what they cost last frame (RESTIR_ROW_ORDER, csrc/stages.hip k_row_order) against the
# screen order; every workload, stream priorities at the library default, one box.  Also the direct stage launched alone (scripts/bvh_ab.py).
TAG=${1:-r05_row}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for w in "real --scene-footprint real" "lite --scene-footprint lite" "config3 --config 3" "config5 --config 5" "moving --moving-camera"; do
  set -- $w; name=$1; shift
  for on in 0 1 0 1; do
    RESTIR_ROW_ORDER=$on timeout 600 python bench.py --no-cpu-baseline --stream-priorities default "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$name row_order=$on  frame in flight %.4f ms | serial sum %.4f | latency %.4f | serial stages %s' % (d['ms_per_step'], d.get('ms_per_frame_serial', 0), d.get('frame_latency_ms', 0), r.get('serial', {}).get('stage_ms_per_frame')))
" | tee -a $O/ab.txt
  done
done
for on in 0 1; do
  echo "== 8-rank emulation, real, row_order=$on" | tee -a $O/ab.txt
  RESTIR_ROW_ORDER=$on timeout 1200 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/emulate8_row$on.json 2> /dev/null
  python - $O/emulate8_row$on.json <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   emu8 single", d["single_gpu_serial_ms"], d["single_gpu_frames_in_flight_ms"], "| rank serial", d["rank_ms"], "slowest", d["slowest_rank_ms"], "| period", d.get("rank_period_ms"), "slowest", d.get("slowest_rank_period_ms"))
PY
done
