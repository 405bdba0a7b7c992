#!/bin/bash
# round 6: the 88-register traced kernels x frames in flight (2 / 3) on every bench workload — is there a rule (on the probe frames' filter share) that would pick a winner?
R=$GRAFT_REPO_ROOT; T=${1:-r06_grid}; V=$R/cis-565-final-vr-raytracer_amd/csrc/_vg
cd $R
run() { local tag=$1; shift; REPS=1 BENCH_ARGS="$*" bash scripts/ab_libs2.sh $T/$tag "m2_96|-" "m3_96|RESTIR_OVERLAP=3" "m2_88|RESTIR_HIP_LIB=$V/librestir_hip_v8888.so" "m3_88|RESTIR_OVERLAP=3 RESTIR_HIP_LIB=$V/librestir_hip_v8888.so" | sed "s/^/$tag  /"; }
run real
run pose2 --pose 2
run moving --config 4 --moving-camera
run lite --scene-footprint lite
run pose1 --pose 1
run cfg3 --config 3
run cfg5 --config 5
run real_again
