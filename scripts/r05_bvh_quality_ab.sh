#!/bin/bash
# round 5: tree quality A/B (spatial splits, csrc/bvh8_builder.cpp BuilderS) on the lite and the real exterior scene, one box:
#   steps per ray + the two traced stages alone (scripts/bvh_ab.py), the frame in flight (bench.py), then the 8-rank emulation for off vs the best setting
TAG=${1:-r05_bvh}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for kind in PROC_BISTRO_EXT_REAL PROC_BISTRO_EXT; do
  fp=lite; [ $kind = PROC_BISTRO_EXT_REAL ] && fp=real
  for setting in "0 1e-5" "1 1e-4" "1 1e-5" "1 1e-6"; do
    set -- $setting
    export RESTIR_BVH_SPLIT=$1 RESTIR_BVH_SPLIT_ALPHA=$2
    echo "== $kind split=$1 alpha=$2" | tee -a $O/ab.txt
    timeout 600 python scripts/bvh_ab.py $kind 2>/dev/null | tail -1 | tee -a $O/ab.txt
    timeout 600 python bench.py --scene-footprint $fp --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('   frame in flight %.4f ms  %.1f Mrays/s | serial sum %.4f | latency %.4f | serial stages %s' % (d['ms_per_step'], d['value'], d.get('ms_per_frame_serial', 0), d.get('frame_latency_ms', 0), r.get('serial', {}).get('stage_ms_per_frame')))
" | tee -a $O/ab.txt
  done
done
unset RESTIR_BVH_SPLIT RESTIR_BVH_SPLIT_ALPHA
