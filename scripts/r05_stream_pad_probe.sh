#!/bin/bash
# round 5: how does a stream's position in the process's creation order change what the frames-in-flight schedule gets out of it?  Config 3, the default priorities and
# filter-high, a / b idle streams created before the filter / indirect stream (RESTIR_STREAM_PAD), one fresh process per point.
TAG=${1:-r05_pad}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for p in "1,0" "1,1" "0,1"; do
  for a in 0 1 2 3 4 5; do
    line="prio $p a=$a:"
    for b in 0 1 2 3 4; do
      ms=$(RESTIR_STREAM_PAD="$a,$b" timeout 300 python bench.py --config 3 --no-cpu-baseline --profile-run --steps 60 --warmup 10 --stream-priorities=$p 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      line="$line b=$b $ms |"
    done
    echo "$line" | tee -a $O/pad.txt
  done
done
