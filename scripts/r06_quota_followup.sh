#!/bin/bash
# round 6: after the quota finding (16 of 256 CPUs) — the builder and the oracle with thread counts from rt_cpu_budget(), the bench line's new cpu_baseline, the GPU suite's time
R=$GRAFT_REPO_ROOT; T=${1:-r06q}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
bash scripts/r06_build_profile.sh $T/build 2>&1 | grep "==\|sequential\|BVH2\|load_scene\|fresh:" | cut -c1-200
echo "== RESTIR_CPUS=32"; RESTIR_CPUS=32 python gpurun_out/r06t/build/one.py renderer 2>&1 | grep "sequential\|BVH2\|load_scene" | cut -c1-200
echo "== RESTIR_CPUS=64"; RESTIR_CPUS=64 python gpurun_out/r06t/build/one.py renderer 2>&1 | grep "sequential\|BVH2\|load_scene" | cut -c1-200
timeout 800 python scripts/r06_oracle_scaling.py 2>&1 | grep threads | cut -c1-520
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["cpu_baseline"]
print("bench", d["ms_per_step"], d["value"], "bvh8_build_s", d["config"].get("bvh8_build_s"), "| cpu", {k: c[k] for k in ("value", "cores", "host_cpus", "cpu_quota", "threads_used", "cpus_busy", "per_core", "single_thread", "pinned")})
print([(p["threads"], p["mrays_s"], p["cpus_busy"], p["seconds_best"]) for p in c["scaling"]]); print(c["sample"])
PY
( time timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; tail -2 $O/pytest_gpu.log
