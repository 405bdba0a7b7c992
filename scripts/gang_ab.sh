#!/bin/bash
# gang mode of the latency build (idle ray slots of a wave work for its last rays): parity on both builds, bands, 8-rank emulation at 1080p / 4K
R=$GRAFT_REPO_ROOT; T=${1:-r04gang}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_digests.py tests/test_gpu_ref_vectors.py tests/test_gpu_mgpu.py -m gpu -q -x > $O/parity.log 2>&1; tail -2 $O/parity.log
for G in ${GS:-0 4 7 2}; do
  echo "==== RESTIR_GANG=$G"
  RESTIR_GANG=$G timeout 900 python scripts/band_ab.py 496 512 544 560 496 528 528 576 256 368 > $O/band_g$G.txt 2>&1; grep rows $O/band_g$G.txt
  RESTIR_GANG=$G timeout 900 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/emu8_g$G.json 2> $O/e.err
  python - $O/emu8_g$G.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("1080p N=8: one", d["single_gpu_frames_in_flight_ms"], "| serial", d["rank_ms"], "slowest", d["slowest_rank_ms"], "| period", d.get("rank_period_ms"), "slowest", d.get("slowest_rank_period_ms"))
PY
done
