#!/bin/bash
# Round 6, verdict item 6 + advisor (medium): the priority rule on warm probe frames.  Per workload, each cell a fresh process: `auto` (the rule: reports the warm filter
# share of the third probe frame and what it chose), then the two settings the rule chooses between, explicit.  usage (gpurun): bash scripts/r06_prio_rule.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r06prio}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
run() {   # label, bench args
  local label=$1; shift
  for s in auto 1,0 1,1; do
    timeout 600 python bench.py --no-cpu-baseline --stream-priorities $s "$@" > $O/${label}_$s.json 2> $O/${label}_$s.err || { echo "$label $s failed"; tail -2 $O/${label}_$s.err; }
  done
  python - $O $label <<'PY' | tee -a $O/summary.txt
import json, sys
O, label = sys.argv[1], sys.argv[2]
d = {s: json.loads(open(f"{O}/{label}_{s}.json").read().strip().splitlines()[-1]) for s in ("auto", "1,0", "1,1")}
sp = d["auto"]["stream_priorities"]
a, b, c = d["auto"]["ms_per_step"], d["1,0"]["ms_per_step"], d["1,1"]["ms_per_step"]
best = min(b, c)
print("%-22s warm share %.4f (margin %+.0f %%) -> %s | auto %.4f  (1,0) %.4f  (1,1) %.4f | best %s, the rule loses %+.2f %%" % (label, sp["filter_share"], 100 * sp.get("margin_rel", 0), sp["chosen"], a, b, c,
      "(1,0)" if b <= c else "(1,1)", 100 * (a / best - 1)))
PY
}
run real
run lite --scene-footprint lite
run config3 --config 3
run config5 --config 5
run moving --moving-camera
run pose1 --pose 1
run pose2 --pose 2
run config2_sponza_lite --config 3 --scene-footprint lite
