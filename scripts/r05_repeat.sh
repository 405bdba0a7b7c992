#!/bin/bash
# round 5: does a bench line repeat?  (stream cache instead of stream re-creation; thread-local counters in the oracle)  Three back-to-back default lines + the slot-assignment experiment
TAG=${1:-r05_repeat}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_stream_prio.py tests/test_gpu_fullsize.py tests/test_gpu_bench_cli.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log | head -1
for k in 1 2 3; do
  timeout 900 python bench.py > $O/bench_$k.json 2> $O/bench_$k.err
  python - $O/bench_$k.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); cb=d["cpu_baseline"]
print("bench", d["ms_per_step"], d["sustained"]["ms_per_frame"], d["stream_priorities"]["chosen"], {k: v for k, v in d["stream_priorities"]["ms_per_frame"].items()}, "| cpu", cb["value"], cb["per_thread"], cb["parallel_efficiency_all_vs_1"], [(p["threads"], p["mrays_s"], p["cpu_seconds"]) for p in cb["scaling"]], cb["host_loadavg_before_after"])
PY
done
for c in "--config 3" "--config 5" "--moving-camera" "--scene-footprint lite"; do
  timeout 900 python bench.py --no-cpu-baseline $c 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$c', d['ms_per_step'], d['sustained']['ms_per_frame'], d['stream_priorities'])
"
done
for kind in "PROC_BISTRO_EXT_REAL real" "PROC_BISTRO_EXT lite"; do
  set -- $kind
  for slots in greedy opt greedy opt; do
    echo "== $1 slots=$slots"
    RESTIR_BVH_SLOTS=$slots timeout 600 python scripts/bvh_ab.py $1 2>/dev/null | tail -1 | cut -c1-420
    RESTIR_BVH_SLOTS=$slots timeout 600 python bench.py --scene-footprint $2 --no-cpu-baseline --stream-priorities default 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('   frame in flight %.4f ms | serial sum %.4f | serial stages %s' % (d['ms_per_step'], d.get('ms_per_frame_serial', 0), r.get('serial', {}).get('stage_ms_per_frame')))
"
  done
done
