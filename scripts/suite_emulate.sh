#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03z}; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_trace_pin.py > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
bash scripts/emulate8.sh $1
timeout 600 python bench.py --steps 60 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench", d["value"], d["unit"], d["ms_per_step"], "serial", d.get("ms_per_frame_serial"), "latency", d.get("frame_latency_ms"))
PY
