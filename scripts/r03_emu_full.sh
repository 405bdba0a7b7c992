#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03m}; mkdir -p $O; shift
cd $R
for p in 2 0; do
  RESTIR_MGPU_PRIO=$p timeout 900 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/emulate8_1080p_prio$p.json 2> $O/e1.err
  python - $O/emulate8_1080p_prio$p.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "serial", d["rank_ms"], "period", d.get("rank_period_ms"), "one", d["single_gpu_serial_ms"], d["single_gpu_frames_in_flight_ms"])
PY
done
RESTIR_MGPU_PRIO=2 timeout 1500 python bench.py --emulate-world 8 --width 3840 --height 2160 --steps 20 --warmup 10 > $O/emulate8_4k_prio2.json 2> $O/e2.err; cat $O/emulate8_4k_prio2.json
