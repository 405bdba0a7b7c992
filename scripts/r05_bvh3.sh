#!/bin/bash
# round 5, tree quality, third batch: insertion-based optimisation (RESTIR_BVH_REINSERT) with / without spatial splits and rotations, real and lite scene, one box
TAG=${1:-r05_bvh3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
one() { # kind footprint split rotate reinsert alpha
  export RESTIR_BVH_SPLIT=$3 RESTIR_BVH_ROTATE=$4 RESTIR_BVH_REINSERT=$5 RESTIR_BVH_SPLIT_ALPHA=$6
  echo "== $1 split=$3 rotate=$4 reinsert=$5 alpha=$6" | tee -a $O/ab.txt
  timeout 600 python scripts/bvh_ab.py $1 2>/dev/null | tail -1 | tee -a $O/ab.txt
  timeout 600 python bench.py --scene-footprint $2 --no-cpu-baseline --stream-priorities default 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('   frame in flight %.4f ms  %.1f Mrays/s | serial sum %.4f | latency %.4f | serial stages %s' % (d['ms_per_step'], d['value'], d.get('ms_per_frame_serial', 0), d.get('frame_latency_ms', 0), r.get('serial', {}).get('stage_ms_per_frame')))
" | tee -a $O/ab.txt
}
for s in "0 0 0 1e-5" "1 4 0 1e-5" "0 4 8 1e-5" "1 4 8 1e-5" "1 4 8 1e-4" "1 8 16 1e-5" "1 0 8 1e-5"; do one PROC_BISTRO_EXT_REAL real $s; done
for s in "0 0 0 1e-5" "1 4 0 1e-5" "0 4 8 1e-5" "1 4 8 1e-5"; do one PROC_BISTRO_EXT lite $s; done
export RESTIR_BVH_SPLIT=1 RESTIR_BVH_ROTATE=4 RESTIR_BVH_REINSERT=8 RESTIR_BVH_SPLIT_ALPHA=1e-5
for c in 3 5; do
  for on in 0 1; do
    if [ $on == 0 ]; then export RESTIR_BVH_SPLIT=0 RESTIR_BVH_ROTATE=0 RESTIR_BVH_REINSERT=0; else export RESTIR_BVH_SPLIT=1 RESTIR_BVH_ROTATE=4 RESTIR_BVH_REINSERT=8; fi
    echo "== config $c quality passes $on" | tee -a $O/ab.txt
    timeout 600 python bench.py --config $c --no-cpu-baseline --stream-priorities default 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('   frame in flight %.4f ms  %.1f Mrays/s | serial sum %.4f | accel %s' % (d['ms_per_step'], d['value'], d.get('ms_per_frame_serial', 0), d['config']['accel']))
" | tee -a $O/ab.txt
  done
done
export RESTIR_BVH_SPLIT=1 RESTIR_BVH_ROTATE=4 RESTIR_BVH_REINSERT=8 RESTIR_BVH_SPLIT_ALPHA=1e-5
python -c "import restir_amd; from restir_amd import build; build.build_hip(variant='prof', extra_flags=['-DRT_WAVEPROF=1'])" > /dev/null 2>&1
for q in "0 0 0" "1 4 8"; do
  set -- $q
  export RESTIR_BVH_SPLIT=$1 RESTIR_BVH_ROTATE=$2 RESTIR_BVH_REINSERT=$3
  echo "==== wave profile split=$1 rotate=$2 reinsert=$3" | tee -a $O/lanes.txt
  RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_prof.so WAVE_PROFILE_KIND=PROC_BISTRO_EXT_REAL timeout 900 python scripts/wave_profile.py 0 1080 > $O/wave_$1$2$3.txt 2>&1
  grep -E "^== |rounds by|all waves|wave time|^   \(" $O/wave_$1$2$3.txt | tee -a $O/lanes.txt
done
