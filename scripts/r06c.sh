#!/bin/bash
# r06c: BVH build on the box + host period with the pipelined RCCL host forced + emulate8 per RESTIR_MGPU_PRIO setting (in process)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R
bash scripts/r06_bvh_build.sh r06c 2>&1 | tee $O/bvh.txt
echo "== host period, pipelined RCCL host forced at world 1" | tee $O/summary.txt
timeout 1200 python scripts/r06_host_period.py --reps 2 2> $O/host_period.err | tee $O/host_period.txt | grep -v '^{' | tee -a $O/summary.txt
for prio in 000 +00 0+0 000 +00 0+0; do
  RESTIR_MGPU_PRIO=$prio timeout 900 python bench.py --emulate-world 8 --no-cpu-baseline > $O/emulate8_$prio.json 2> $O/emulate8_$prio.err || tail -3 $O/emulate8_$prio.err
  python - $O/emulate8_$prio.json $prio <<'PY' | tee -a $O/summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("prio %s: single %.3f / %.3f; slowest rank serial %.3f period %.3f (x%.2f); periods %s" % (sys.argv[2], d["single_gpu_serial_ms"], d["single_gpu_frames_in_flight_ms"], d["slowest_rank_ms"], d["slowest_rank_period_ms"],
      d["projected_speedup_period_vs_single_gpu_frames_in_flight"], d["rank_period_ms"]))
PY
done
