#!/bin/bash
# the emulation set of a round: N = 2, 4, 8 at 1080p and N = 4, 8 at 4K
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03o}; mkdir -p $O
cd $R
for n in 2 4 8; do timeout 900 python bench.py --emulate-world $n --steps 30 --warmup 12 > $O/emulate${n}_1080p.json 2> $O/e.err; done
for n in 4 8; do timeout 1800 python bench.py --emulate-world $n --width 3840 --height 2160 --steps 20 --warmup 10 > $O/emulate${n}_4k.json 2> $O/e.err; done
for f in $O/emulate*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "one", d["single_gpu_serial_ms"], d["single_gpu_frames_in_flight_ms"], "serial slowest", d["slowest_rank_ms"], "period", d.get("rank_period_ms"), "steady MB", [round(x/1e6,1) for x in d.get("rank_bytes_steady",[])], "xgmi", d.get("xgmi_model",{}).get("on_frame_path_ms"))
PY
done
