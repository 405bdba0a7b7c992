#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03i}; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_mgpu.py -m gpu -x -q > $O/pytest_mgpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_mgpu.log
tail -25 $O/pytest_mgpu.log
