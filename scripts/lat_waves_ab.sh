#!/bin/bash
# waves per workgroup of the latency kernels when two traced launches share the chip (frames in flight inside a rank): 8-rank emulation at 1080p and 4K
R=$GRAFT_REPO_ROOT; T=${1:-r04lw}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
for V in "8" "6" "4" "3"; do
  for S in 1080p 4k; do
    A=""; [ $S == 4k ] && A="--width 3840 --height 2160"
    RESTIR_LAT_WAVES_SHARED=$V timeout 1200 python bench.py --emulate-world 8 $A --steps 24 --warmup 10 > $O/emu8_${S}_w$V.json 2> $O/e.err
    python - $O/emu8_${S}_w$V.json $V $S <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("shared waves", sys.argv[2], sys.argv[3], "one", d["single_gpu_frames_in_flight_ms"], "serial slowest", d["slowest_rank_ms"], "period slowest", d.get("slowest_rank_period_ms"), d.get("rank_period_ms"))
PY
  done
done
