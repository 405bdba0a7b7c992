#!/bin/bash
# the frames-in-flight tests repeated (a schedule race shows up as an intermittent mismatch)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03ms}; mkdir -p $O
cd $R
pass=0; fail=0
for i in $(seq 1 ${2:-12}); do
  if timeout 600 python -m pytest tests/test_gpu_mgpu.py -m gpu -x -q -k "frames_in_flight" > $O/run$i.log 2>&1; then pass=$((pass+1)); else fail=$((fail+1)); grep "AssertionError" $O/run$i.log | head -2; fi
done
echo "frames-in-flight runs: $pass passed, $fail failed"
