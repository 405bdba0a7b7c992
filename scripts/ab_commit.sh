#!/bin/bash
# timing of the build in the tree against the build of a git revision (default HEAD) on the same box: the caller builds the old revision into csrc/_ab/lib_old.so (git-ignored) first
# out into a scratch copy, builds both, alternates the measurements.  usage: ab_commit.sh [rev]
cd $GRAFT_REPO_ROOT
REV=${1:-HEAD}
echo "(no git on the box: the caller passes the two prebuilt libraries instead)" > /dev/null
for rep in 1 2; do for L in "$GRAFT_REPO_ROOT/cis-565-final-vr-raytracer_amd/csrc/_ab/lib_old.so" ""; do
  echo "== ${L:-tree}"
  for ov in 0 2; do env RESTIR_HIP_LIB=$L RESTIR_OVERLAP=$ov timeout 600 python scripts/gpu_perf.py bistro 2>&1 | grep '"case"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'], d['wall_ms'], d['stage_ms'])"; done
done; done
