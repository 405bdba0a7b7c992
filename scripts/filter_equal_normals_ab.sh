#!/bin/bash
# the equal-normal shortcut of the A-Trous pair weight (a wave whose pairs all have equal normals skips one of three exponentials): parity, then config 4 and 3
R=$GRAFT_REPO_ROOT; T=${1:-r04fe}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_allstages.py tests/test_golden_digests.py tests/test_gpu_ref_vectors.py -m gpu -q -x > $O/parity.log 2>&1; tail -2 $O/parity.log
bash scripts/variants_bench.sh $T "shortcut|-|-" "general|-DRT_NO_EQUAL_NORMAL_SHORTCUT|-" "shortcut2|-|-" "general2|-DRT_NO_EQUAL_NORMAL_SHORTCUT|-"
CFG=3 bash scripts/variants_bench.sh $T/c3 "c3_shortcut|-|-" "c3_general|-DRT_NO_EQUAL_NORMAL_SHORTCUT|-"
