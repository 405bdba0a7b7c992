#!/bin/bash
# round 5, end of round on ONE box: the whole GPU suite, the artefacts of scripts/final_measure.sh, the emulation set, a randomised parity campaign — all on the build in the tree
TAG=${1:-r05z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu exit $?"; tail -3 $O/pytest_gpu.log | head -1
bash scripts/final_measure.sh $TAG > $O/final_measure.log 2>&1; tail -12 $O/final_measure.log | cut -c1-250
bash scripts/emulate_set.sh $TAG > $O/emulate_set.log 2>&1; cat $O/emulate_set.log | cut -c1-300
bash scripts/fuzz_campaign.sh $TAG/fuzz ${2:-600} ${3:-200} 41 2>&1 | cut -c1-200
