#!/bin/bash
# Same-box A/B of measurement builds and environment settings on the bench scene (round 4).  usage (gpurun):
#   bash scripts/variants_bench.sh <tag> "<label>|<extra hipcc flags or ->|<env assignments or ->" ...
# A variant with flags is built ON THE BOX into csrc/_ab/librestir_hip_<label>.so (measurement builds do not travel); "-" flags = the product library.
# Prints per variant: frames in flight (bench.py timed region + sustained), serial sum and the serial stage times; LAT=1 adds frame latency, CFG=<n> uses --config n,
# BENCH_ARGS="..." is passed to bench.py (round 5: --scene-footprint real).
R=$GRAFT_REPO_ROOT; T=${1:-r04ab}; O=$R/gpurun_out/$T; mkdir -p $O; shift
cd $R
for v in "$@"; do
  IFS='|' read -r label flags envs <<< "$v"
  unset RESTIR_HIP_LIB
  if [ "$flags" != "-" ] && [ -n "$flags" ]; then
    python -c "import restir_amd; from restir_amd import build; build.build_hip(variant='$label', extra_flags='$flags'.split())" > $O/build_$label.log 2>&1 || { echo "$label: build failed"; tail -5 $O/build_$label.log; continue; }
    export RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_$label.so
  fi
  [ "$envs" == "-" ] && envs="X_=1"
  env $envs timeout 900 python bench.py --no-cpu-baseline ${CFG:+--config $CFG} $BENCH_ARGS > $O/bench_$label.json 2> $O/bench_$label.err || { echo "$label: bench failed"; tail -3 $O/bench_$label.err; continue; }
  python - "$label" $O/bench_$label.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
s = d["roofline"]["serial"]["stage_ms_per_frame"]
print("%-26s in flight %.3f  sustained %.3f  latency %s  serial %.3f | direct %.3f indirect %.3f filters %.3f + %.3f" % (sys.argv[1], d["ms_per_step"], d["sustained"]["ms_per_frame"], d.get("frame_latency_ms"),
      d["ms_per_frame_serial"], s["direct_stage"], s["indirect_stage"], s["denoise_direct"], s["denoise_indirect"]), flush=True)
PY
done
