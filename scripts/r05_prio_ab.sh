#!/bin/bash
# round 5 (verdict item 5): stream priorities of the frames-in-flight schedule by workload.  RESTIR_PRIO = three characters over {-, 0, +} for the main (direct
# stage) / indirect / filter stream.  One box, every workload x every setting, ms per frame in flight (bench.py --profile-run: warm-up + K timed frames only).
TAG=${1:-r05_prio}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
run() { # name, bench args...
  local name=$1; shift
  for prio in 0+0 000 +00 00+ 0++ ++0 +0+ 0+- +0- +-0 0-0 00-; do
    ms=$(RESTIR_PRIO=$prio timeout 600 python bench.py --profile-run --no-cpu-baseline --steps 100 --warmup 20 "$@" 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$name prio=$prio ms_per_frame=$ms" | tee -a $O/prio.txt
  done
}
run config4_real --scene-footprint real
run config4_lite --scene-footprint lite
run config3 --config 3
run config5 --config 5
run config4_moving --moving-camera
