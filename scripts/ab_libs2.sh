#!/bin/bash
# Round 6: same-box A/B of prebuilt libraries / environment settings, alternated REPS times (default 2) so that drift of the box shows.
# usage (gpurun):  [REPS=2] [BENCH_ARGS="--scene-footprint lite"] bash scripts/ab_libs2.sh <tag> "<label>|<env assignments or ->" ...
#   e.g. "r05|RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_prev/librestir_hip_r05.so" "tree|-"
# (csrc/_prev/ holds the library the previous round shipped: git-ignored like every .so, but it travels to the GPU box, unlike csrc/_ab.)
R=$GRAFT_REPO_ROOT; T=${1:-r06ab}; O=$R/gpurun_out/$T; mkdir -p $O; shift
cd $R
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  IFS='|' read -r label envs <<< "$v"
  [ "$envs" == "-" ] && envs="X_=1"
  envs="${envs//\$R/$R}"
  env $envs timeout 900 python bench.py --no-cpu-baseline $BENCH_ARGS > $O/bench_${label}_$rep.json 2> $O/bench_${label}_$rep.err || { echo "$label: bench failed"; tail -3 $O/bench_${label}_$rep.err; continue; }
  python - "$label" $O/bench_${label}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
s = d["roofline"]["serial"]["stage_ms_per_frame"]
print("%-22s in flight %.3f  sustained %.3f  latency %s  serial %.3f | direct %.3f indirect %.3f filters %.3f + %.3f" % (sys.argv[1], d["ms_per_step"], d["sustained"]["ms_per_frame"], d.get("frame_latency_ms"),
      d["ms_per_frame_serial"], s["direct_stage"], s["indirect_stage"], s["denoise_direct"], s["denoise_indirect"]), flush=True)
PY
done; done 2>&1 | tee -a $O/summary.txt
