#!/bin/bash
# round 5, second batch of same-box measurements:
#  (1) live / executing lane histograms of the traced stages, pool order off vs by octant (measurement builds)
#  (2) where the direct stage's time goes on the real scene: full stage vs primary ray + shading only, with 2k^2 and with 128^2 textures, against lite
#  (3) tree quality, second half: rotations; the row-band side (8-rank emulation) of spatial splits
#  (4) counters of the real workload with spatial splits on (does the instruction count follow the steps per ray?)
TAG=${1:-r05_round2}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python -c "import restir_amd; from restir_amd import build; build.build_hip(variant='prof', extra_flags=['-DRT_WAVEPROF=1']); build.build_hip(variant='prof_oct', extra_flags=['-DRT_WAVEPROF=1', '-DRT_POOL_ORDER=1'])" > /dev/null 2>&1
for v in prof prof_oct; do
  echo "==== $v" | tee -a $O/lanes.txt
  RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_$v.so WAVE_PROFILE_KIND=PROC_BISTRO_EXT_REAL timeout 900 python scripts/wave_profile.py 0 1080 > $O/wave_$v.txt 2>&1
  grep -E "^== |rounds by|all waves|wave time" $O/wave_$v.txt | tee -a $O/lanes.txt
done
echo "==== lite, prof" | tee -a $O/lanes.txt
RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_prof.so timeout 900 python scripts/wave_profile.py 0 1080 > $O/wave_prof_lite.txt 2>&1
grep -E "^== |rounds by|all waves|wave time" $O/wave_prof_lite.txt | tee -a $O/lanes.txt
for v in "PROC_BISTRO_EXT_REAL X_=1" "PROC_BISTRO_EXT_REAL RESTIR_SCENE_TEXSIZE=128" "PROC_BISTRO_EXT X_=1" "PROC_BISTRO_EXT_REAL RESTIR_DEBUG_OPAQUE_LEAVES=1"; do
  set -- $v
  env $2 timeout 600 python scripts/gpu_direct_breakdown.py $1 2>/dev/null | grep -E "PROC|mode 3 M 4|mode 0|debug view" | tee -a $O/direct_breakdown.txt
  env $2 timeout 600 python bench.py --no-cpu-baseline --stream-priorities default 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('   frame in flight %.4f ms | serial sum %.4f | serial stages %s | texels %d MB' % (d['ms_per_step'], d.get('ms_per_frame_serial', 0), r.get('serial', {}).get('stage_ms_per_frame'), d['config']['texture_bytes'] // 1000000))
" | tee -a $O/direct_breakdown.txt
done
for setting in "0 0 1e-5" "0 4 1e-5" "1 0 1e-5" "1 4 1e-5" "1 4 1e-4"; do
  set -- $setting
  export RESTIR_BVH_SPLIT=$1 RESTIR_BVH_ROTATE=$2 RESTIR_BVH_SPLIT_ALPHA=$3
  echo "== real split=$1 rotate=$2 alpha=$3" | tee -a $O/bvh2.txt
  timeout 600 python scripts/bvh_ab.py PROC_BISTRO_EXT_REAL 2>/dev/null | tail -1 | tee -a $O/bvh2.txt
  timeout 600 python bench.py --no-cpu-baseline --stream-priorities default 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('   frame in flight %.4f ms  %.1f Mrays/s | serial sum %.4f | latency %.4f | serial stages %s' % (d['ms_per_step'], d['value'], d.get('ms_per_frame_serial', 0), d.get('frame_latency_ms', 0), r.get('serial', {}).get('stage_ms_per_frame')))
" | tee -a $O/bvh2.txt
  if [ "$setting" == "0 0 1e-5" ] || [ "$setting" == "1 4 1e-5" ]; then
    timeout 1200 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/emulate8_split$1_rot$2.json 2> /dev/null
    python - $O/emulate8_split$1_rot$2.json <<'PY' | tee -a $O/bvh2.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   emu8 single", d["single_gpu_serial_ms"], d["single_gpu_frames_in_flight_ms"], "| rank serial", d["rank_ms"], "slowest", d["slowest_rank_ms"], "| period", d.get("rank_period_ms"), "slowest", d.get("slowest_rank_period_ms"))
PY
  fi
done
export RESTIR_BVH_SPLIT=1 RESTIR_BVH_ROTATE=4 RESTIR_BVH_SPLIT_ALPHA=1e-5
bash scripts/pmc.sh $TAG/pmc_split config4_real_split > $O/pmc_split.log 2>&1
unset RESTIR_BVH_SPLIT RESTIR_BVH_ROTATE RESTIR_BVH_SPLIT_ALPHA
bash scripts/pmc.sh $TAG/pmc_base auto > $O/pmc_base.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
