#!/bin/bash
# round 5, tree quality, fourth batch: a sweep around the shipped setting (splits alpha 1e-5 + 4 rotation passes) on the real scene, one box
TAG=${1:-r05_bvh4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
one() { # label, env assignments...
  local label=$1; shift
  echo "== $label" | tee -a $O/ab.txt
  env "$@" timeout 600 python scripts/bvh_ab.py PROC_BISTRO_EXT_REAL 2>/dev/null | tail -1 | tee -a $O/ab.txt
  env "$@" timeout 600 python bench.py --no-cpu-baseline --stream-priorities default 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('   frame in flight %.4f ms  %.1f Mrays/s | serial sum %.4f | latency %.4f | serial stages %s' % (d['ms_per_step'], d['value'], d.get('ms_per_frame_serial', 0), d.get('frame_latency_ms', 0), r.get('serial', {}).get('stage_ms_per_frame')))
" | tee -a $O/ab.txt
}
one "shipped (split 1e-5, rotate 4)" X_=1
one "rotate 2" RESTIR_BVH_ROTATE=2
one "rotate 8" RESTIR_BVH_ROTATE=8
one "rotate 4 + grandchild swaps" RESTIR_BVH_ROTATE_GG=1
one "alpha 3e-6" RESTIR_BVH_SPLIT_ALPHA=3e-6
one "alpha 3e-5" RESTIR_BVH_SPLIT_ALPHA=3e-5
one "alpha 1e-6 budget 0.5" RESTIR_BVH_SPLIT_ALPHA=1e-6 RESTIR_BVH_SPLIT_BUDGET=0.5
one "32 bins" RESTIR_BVH_BINS=32
one "shipped again" X_=1
