#!/bin/bash
# parity of the two builds + same-band A/B; $1 = tag, $2 = "tests" to run the parity files first
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03e}; mkdir -p $O
cd $R
if [ "$2" == "tests" ]; then
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_vectors.py tests/test_golden_digests.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest exit $?" >> $O/pytest_parity.log
tail -12 $O/pytest_parity.log
fi
timeout 900 python scripts/band_ab.py > $O/band_ab.txt 2>&1; grep "rows\|Error\|error" $O/band_ab.txt
