#!/bin/bash
# A/B of the cooperative-tail threshold (RESTIR_COOP = live rays per wave at or under which triangle steps go cooperative)
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  echo "== RESTIR_COOP=$V"
  for ov in 0 2; do env RESTIR_COOP=$V RESTIR_OVERLAP=$ov timeout 600 python scripts/gpu_perf.py sponza bistro 2>&1 | grep '"case"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'], d['wall_ms'], d['stage_ms']['direct'], d['stage_ms']['indirect'])"; done
  env RESTIR_COOP=$V python bench.py --emulate-world 8 --emulate-rank 3 --steps 40 --warmup 6 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=8 rank3', d['ms_per_step'])"
  env RESTIR_COOP=$V python bench.py --emulate-world 2 --emulate-rank 0 --steps 40 --warmup 6 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=2 rank0', d['ms_per_step'])"
done
