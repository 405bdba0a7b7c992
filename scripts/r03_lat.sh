#!/bin/bash
# GPU suite + the latency-build A/B on the per-rank metric (bench.py --emulate-world 8) + a 1-GPU sanity line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03b}; mkdir -p $O
cd $R
if [ "$2" != "notests" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
fi
for v in 0 1; do
  RESTIR_LAT=$v timeout 600 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/emulate8_lat$v.json 2> $O/emulate8_lat$v.err; cat $O/emulate8_lat$v.json
done
timeout 600 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/emulate8_auto.json 2> $O/emulate8_auto.err; cat $O/emulate8_auto.json
timeout 600 python bench.py --steps 60 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cat $O/bench.json
