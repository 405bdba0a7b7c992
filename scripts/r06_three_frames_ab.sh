#!/bin/bash
# round 6: a third frame in flight (rt_set_overlap 3 / RESTIR_OVERLAP=3) against the default two, every cell a fresh process; stream levels by the rule (auto) and forced.
#   usage (gpurun): bash scripts/r06_three_frames_ab.sh <tag>
TAG=${1:-r06_three}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
line() {  # label, env, bench args
  local label="$1" envs="$2"; shift; shift
  env $envs timeout 600 python bench.py --no-cpu-baseline "$@" > $O/cell.json 2> $O/cell.err
  python - "$label" $O/cell.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    s = d["roofline"]["serial"]["stage_ms_per_frame"]
    print("%-44s in flight %.3f  sustained %.3f  latency %s  serial %.3f | prio %s" % (sys.argv[1], d["ms_per_step"], d["sustained"]["ms_per_frame"], d.get("frame_latency_ms", "?"),
          d["ms_per_frame_serial"], json.dumps({k: v for k, v in (d.get("stream_priorities") or {}).items() if k not in ("how",)})[:110]), flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace(".json", ".err")).read()[-400:])
PY
}
for rep in 1 2; do
  for m in 2 3; do
    line "real   overlap $m auto" "RESTIR_OVERLAP=$m"
    line "real   overlap $m ind+ (2)" "RESTIR_OVERLAP=$m RESTIR_PRIO=2"
    line "real   overlap $m both+ (1)" "RESTIR_OVERLAP=$m RESTIR_PRIO=1"
  done
done
for m in 2 3; do
  line "real   overlap $m none (0)" "RESTIR_OVERLAP=$m RESTIR_PRIO=0"
  line "lite   overlap $m auto" "RESTIR_OVERLAP=$m" --scene-footprint lite
  line "lite   overlap $m both+" "RESTIR_OVERLAP=$m RESTIR_PRIO=1" --scene-footprint lite
  line "cfg3   overlap $m auto" "RESTIR_OVERLAP=$m" --config 3
  line "cfg3   overlap $m ind+" "RESTIR_OVERLAP=$m RESTIR_PRIO=2" --config 3
  line "cfg5   overlap $m auto" "RESTIR_OVERLAP=$m" --config 5
  line "cfg5   overlap $m ind+" "RESTIR_OVERLAP=$m RESTIR_PRIO=2" --config 5
  line "moving overlap $m auto" "RESTIR_OVERLAP=$m" --config 4 --moving-camera
  line "pose1  overlap $m auto" "RESTIR_OVERLAP=$m" --pose 1
  line "pose2  overlap $m auto" "RESTIR_OVERLAP=$m" --pose 2
done
