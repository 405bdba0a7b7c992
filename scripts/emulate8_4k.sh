#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03x4k}; mkdir -p $O; shift
cd $R
for e in "$@"; do export $e; done
timeout 1800 python bench.py --emulate-world 8 --width 3840 --height 2160 --steps 20 --warmup 10 > $O/emulate8_4k.json 2> $O/e.err
python - $O/emulate8_4k.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("one", d["single_gpu_serial_ms"], d["single_gpu_frames_in_flight_ms"], "| serial", d["rank_ms"], "slowest", d["slowest_rank_ms"])
print("period", d.get("rank_period_ms"), "slowest", d.get("slowest_rank_period_ms"), "bands", d.get("bands_period_balanced"))
PY
