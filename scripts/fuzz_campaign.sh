#!/bin/bash
# randomised parity campaign of the build in the tree: single-GPU frame vs oracle (both traversal builds), native multi-GPU frame vs single-GPU frame
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-fuzz}; mkdir -p $O
cd $R
C1=${2:-400}; C2=${3:-120}; SEED=${4:-31}
RESTIR_FUZZ_CASES=$C1 RESTIR_FUZZ_SEED=$SEED timeout ${FUZZ_TIMEOUT:-2400} python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k random_configurations > $O/gpu_fuzz.log 2>&1; echo "gpu fuzz ($C1 cases, seed $SEED) exit $?"; tail -2 $O/gpu_fuzz.log
timeout ${FUZZ_TIMEOUT:-2400} python scripts/mgpu_fuzz.py $C2 $SEED > $O/mgpu_fuzz.log 2>&1; echo "mgpu fuzz exit $?"; grep -c MISMATCH $O/mgpu_fuzz.log; tail -1 $O/mgpu_fuzz.log
# the float64 pin of the ray / triangle decision at the round-3 coverage (100 k rays per scene on both legs; the default suite runs 50 k / 40 k for the driver's clock)
RESTIR_PIN_RAYS=100000 RESTIR_PIN_RAYS_GPU=100000 timeout ${FUZZ_TIMEOUT:-2400} python -m pytest tests/test_trace_pin.py -x -q > $O/trace_pin_100k.log 2>&1; echo "trace pin 100k exit $?"; tail -2 $O/trace_pin_100k.log
