#!/bin/bash
# quick perf check of a build: bench line, stages on bands (throughput vs latency kernels), 8-rank emulation
R=$GRAFT_REPO_ROOT; T=${1:-r03quick}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 600 python bench.py --steps 60 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench", d["value"], d["unit"], d["ms_per_step"], "serial", d.get("ms_per_frame_serial"), "latency", d.get("frame_latency_ms"))
print("   serial stages", d["roofline"].get("stage_ms_per_frame"))
PY
timeout 900 python scripts/band_ab.py 496 512 544 560 496 528 528 576 256 368 > $O/band.txt 2>&1; grep rows $O/band.txt
[ "$2" == "noemu" ] || bash scripts/emulate8.sh $T
