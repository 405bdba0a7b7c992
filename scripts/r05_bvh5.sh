#!/bin/bash
# round 5, tree quality, fifth batch: the collapse and the leaf / bin parameters around the shipped tree (real scene, one box); then the lite scene for the best
TAG=${1:-r05_bvh5}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
one() { # label, kind, footprint, env assignments...
  local label=$1 kind=$2 fp=$3; shift; shift; shift
  echo "== $label" | tee -a $O/ab.txt
  env "$@" timeout 600 python scripts/bvh_ab.py $kind 2>/dev/null | tail -1 | tee -a $O/ab.txt
  env "$@" timeout 600 python bench.py --scene-footprint $fp --no-cpu-baseline --stream-priorities default 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('   frame in flight %.4f ms  %.1f Mrays/s | serial sum %.4f | latency %.4f | serial stages %s' % (d['ms_per_step'], d['value'], d.get('ms_per_frame_serial', 0), d.get('frame_latency_ms', 0), r.get('serial', {}).get('stage_ms_per_frame')))
" | tee -a $O/ab.txt
}
one "real shipped" PROC_BISTRO_EXT_REAL real X_=1
one "real SAH-optimal collapse (dp)" PROC_BISTRO_EXT_REAL real RESTIR_BVH_COLLAPSE=dp
one "real slot cost 0.125" PROC_BISTRO_EXT_REAL real RESTIR_BVH_SLOTCOST=0.125
one "real slot cost 0.5" PROC_BISTRO_EXT_REAL real RESTIR_BVH_SLOTCOST=0.5
one "real spatial bins 32" PROC_BISTRO_EXT_REAL real RESTIR_BVH_SBINS=32
one "real spatial bins 8" PROC_BISTRO_EXT_REAL real RESTIR_BVH_SBINS=8
one "real object bins 32, spatial 16" PROC_BISTRO_EXT_REAL real RESTIR_BVH_BINS=32 RESTIR_BVH_SBINS=16
one "real object bins 8" PROC_BISTRO_EXT_REAL real RESTIR_BVH_BINS=8 RESTIR_BVH_SBINS=16
one "lite shipped" PROC_BISTRO_EXT lite X_=1
one "lite dp" PROC_BISTRO_EXT lite RESTIR_BVH_COLLAPSE=dp
