#!/bin/bash
# (ran with `struct alignas(128) Node8` behind -DRT_NODE_PAD in csrc/bvh8.h; not kept)
# BVH8 nodes padded to one 128-byte cache line each (-DRT_NODE_PAD) against the 80-byte stride (round 4): parity subset on the padded build, bench line A/B,
# 8-rank emulation.   usage (gpurun): bash scripts/node_pad_ab.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r04pad}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
LAT=1 bash scripts/variants_bench.sh $T "stride80|-|-" "pad128|-DRT_NODE_PAD|-" "stride80_again|-|-" | tee $O/ab.txt
RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_pad128.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_digests.py -m gpu -q -x 2>&1 | tail -1 | tee $O/parity.txt
for v in "" pad128; do
  [ -n "$v" ] && export RESTIR_HIP_LIB=$R/cis-565-final-vr-raytracer_amd/csrc/_ab/librestir_hip_$v.so || unset RESTIR_HIP_LIB
  timeout 900 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/emu8_${v:-stride80}.json 2> $O/e.err
  python - $O/emu8_${v:-stride80}.json ${v:-stride80} <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-10s 8 ranks: slowest serial %.3f  slowest period %.3f  (single GPU %.3f in flight)" % (sys.argv[2], d["slowest_rank_ms"], d.get("slowest_rank_period_ms", 0), d["single_gpu_frames_in_flight_ms"]))
PY
done | tee -a $O/ab.txt
