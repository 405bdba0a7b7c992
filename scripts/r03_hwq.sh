#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03hq}; mkdir -p $O
cd $R
for q in 4 8 16 24; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python scripts/mgpu_solo_trace.py 0 30 2>/dev/null | tail -1 | sed "s/^/GPU_MAX_HW_QUEUES=$q /"
  GPU_MAX_HW_QUEUES=$q timeout 600 python scripts/mgpu_solo_trace.py 4 30 2>/dev/null | tail -1 | sed "s/^/GPU_MAX_HW_QUEUES=$q /"
done
