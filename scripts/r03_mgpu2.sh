#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03k}; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_mgpu.py -m gpu -x -q > $O/pytest_mgpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_mgpu.log
tail -5 $O/pytest_mgpu.log
timeout 900 python bench.py --emulate-world 8 --steps 30 --warmup 12 > $O/emulate8_1080p.json 2> $O/emulate8_1080p.err; cat $O/emulate8_1080p.json; tail -3 $O/emulate8_1080p.err
