#!/bin/bash
# per-rank frame time of an N-way row tiling on ONE GPU (exchanges stubbed): bench.py --emulate-world N --emulate-rank r for every rank
cd $GRAFT_REPO_ROOT
for N in 2 4 8; do
  line="N=$N:"
  for r in $(seq 0 $((N-1))); do
    v=$(python bench.py --emulate-world $N --emulate-rank $r --steps 40 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    line="$line $v"
  done
  echo "$line"
done
