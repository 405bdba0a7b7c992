#!/bin/bash
# r06d: GPU suite on the build in the tree; the persistent multi-bounce waves (parity + A/B); the priority rule on warm frames; the builder on the box's host cores
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
(timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; grep -n "passed\|failed" $O/gputests.log | tail -3)
echo "== persistent multi-bounce waves: RESTIR_IND_PERSIST (0 = one wave per tile)" | tee $O/summary.txt
REPS=2 bash scripts/ab_libs2.sh r06d_persist "p0|RESTIR_IND_PERSIST=0" "p2|RESTIR_IND_PERSIST=2" "p3|RESTIR_IND_PERSIST=3" "p4|RESTIR_IND_PERSIST=4" "p6|RESTIR_IND_PERSIST=6"
REPS=1 BENCH_ARGS="--scene-footprint lite" bash scripts/ab_libs2.sh r06d_persist_lite "p0|RESTIR_IND_PERSIST=0" "p2|RESTIR_IND_PERSIST=2" "p3|RESTIR_IND_PERSIST=3" "p4|RESTIR_IND_PERSIST=4"
REPS=1 BENCH_ARGS="--moving-camera" bash scripts/ab_libs2.sh r06d_persist_moving "p0|RESTIR_IND_PERSIST=0" "p2|RESTIR_IND_PERSIST=2" "p3|RESTIR_IND_PERSIST=3"
echo "== priority rule, warm probe frames"
bash scripts/r06_prio_rule.sh r06d_prio
echo "== builder"
bash scripts/r06_bvh_build.sh r06d_bvh 2>&1 | head -30
