#!/bin/bash
# round 6: the builder's own phase profile (RESTIR_BVH_TIMING) in five fresh processes — where the wall time of the BVH2 phase goes when it is 1.2 s and when it is 4 s
R=$GRAFT_REPO_ROOT; T=${1:-r06_build_prof}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
[ -f gpurun_out/r06t/build/one.py ] || { mkdir -p gpurun_out/r06t/build; sed -n "/^cat > \$O\/one.py <<'PY'/,/^PY/p" scripts/r06_build_in_process.sh | sed '1d;$d' > gpurun_out/r06t/build/one.py; }
for v in renderer renderer renderer fresh fresh; do
  echo "== $v  (loadavg $(cut -d' ' -f1-3 /proc/loadavg))"; python gpurun_out/r06t/build/one.py $v 2>&1 | grep "bvh8 build\|load_scene\|hash" | cut -c1-220
done
