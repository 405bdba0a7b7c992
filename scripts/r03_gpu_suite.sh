#!/bin/bash
# full GPU suite + 1-GPU bench line; $1 = tag
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03q}; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -30 $O/pytest_gpu.log
timeout 600 python bench.py --steps 60 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench", d["value"], d["unit"], d["ms_per_step"], "serial", d.get("ms_per_frame_serial"), d["roofline"]["serial"]["stage_ms_per_frame"])
PY
