#!/bin/bash
# band A/B of measurement builds: scripts/variants_ab.sh <tag> <variant> [<variant> ...]   (csrc/_ab/librestir_hip_<variant>.so; "product" = the product build)
R=$GRAFT_REPO_ROOT; T=${1:-r03var}; O=$R/gpurun_out/$T; mkdir -p $O; shift
cd $R
B=${BANDS:-"496 512 544 560 496 528 528 576 464 528"}
for v in "$@"; do
  echo "== $v"
  if [ "$v" == "product" ]; then unset RESTIR_HIP_LIB; else export RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_$v.so; fi
  timeout 900 python scripts/band_ab.py $B > $O/band_$v.txt 2>&1; grep rows $O/band_$v.txt
done
if [ -n "$BENCH" ]; then
for v in "$@"; do
  if [ "$v" == "product" ]; then unset RESTIR_HIP_LIB; else export RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_$v.so; fi
  timeout 600 python bench.py --steps 60 --warmup 20 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python - $O/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench", sys.argv[2], d["ms_per_step"], "serial", d.get("ms_per_frame_serial"), d["roofline"].get("stage_ms_per_frame"))
PY
done
fi
