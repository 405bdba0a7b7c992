#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
for V in "$@"; do
  echo "== $V"
  env $V timeout 600 python scripts/gpu_band_timing.py 8 2>&1 | grep "^rank" | cut -c1-220
  env $V timeout 600 python scripts/gpu_band_timing.py 4 2>&1 | grep "^rank" | cut -c1-220
done
