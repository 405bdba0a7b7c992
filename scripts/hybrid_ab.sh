#!/bin/bash
# hybrid direct stage (heavy tiles of a large launch on the latency build): parity under the switch, then frame / latency / 4K 8-rank emulation per setting
R=$GRAFT_REPO_ROOT; T=${1:-r04hyb}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
RESTIR_HYBRID_PCT=6 RESTIR_HYBRID_MIN_TILES=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_golden_digests.py tests/test_gpu_fuzz.py -m gpu -q -x > $O/parity.log 2>&1; tail -3 $O/parity.log
bash scripts/variants_bench.sh $T "pct0|-|-" "pct2|-|RESTIR_HYBRID_PCT=2" "pct4|-|RESTIR_HYBRID_PCT=4" "pct6|-|RESTIR_HYBRID_PCT=6" "pct10|-|RESTIR_HYBRID_PCT=10" "pct0b|-|-"
for P in 0 4 8; do
  echo "== 4K, 8 emulated ranks, RESTIR_HYBRID_PCT=$P"
  RESTIR_HYBRID_PCT=$P timeout 1200 python bench.py --emulate-world 8 --width 3840 --height 2160 --steps 20 --warmup 8 > $O/emu8_4k_$P.json 2> $O/emu8_4k_$P.err
  python - $O/emu8_4k_$P.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("one", d["single_gpu_serial_ms"], d["single_gpu_frames_in_flight_ms"], "| serial slowest", d["slowest_rank_ms"], "| period slowest", d.get("slowest_rank_period_ms"), d.get("rank_period_ms"))
PY
done
