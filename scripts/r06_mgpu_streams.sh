#!/bin/bash
# Round 6, verdict item 1: the hardware-queue finding of round 5 carried into both multi-GPU hosts and into the emulation.  usage (gpurun): bash scripts/r06_mgpu_streams.sh <tag> [N]
#  1. one band through every host, each in a fresh process (scripts/r06_host_period.py): context streams vs torch's pool, created before / after the process group
#  2. bench.py --emulate-world N (default 8) with lazy per-rank streams, --solo-fresh 2: the two slowest ranks again in processes of their own
#  3. RESTIR_MGPU_PRIO sweep: the slowest and a middle rank of that partition, one fresh process per (rank, setting)
R=$GRAFT_REPO_ROOT; T=${1:-r06m}; N=${2:-8}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
echo "== 1. one band per host (ms/frame, fresh process each)" | tee $O/summary.txt
timeout 1500 python scripts/r06_host_period.py --reps 2 2> $O/host_period.err | tee $O/host_period.txt | grep -v '^{' | tee -a $O/summary.txt
echo "== 2. emulate $N ranks, 1080p, lazy streams, --solo-fresh 2" | tee -a $O/summary.txt
timeout 1500 python bench.py --emulate-world $N --solo-fresh 2 --no-cpu-baseline > $O/emulate$N.json 2> $O/emulate$N.err || tail -3 $O/emulate$N.err
python - $O/emulate$N.json $N <<'PY' | tee -a $O/summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("single GPU serial / in flight %.3f / %.3f ms; slowest rank serial %.3f, period %.3f (x%.2f of in flight); periods %s" % (d["single_gpu_serial_ms"], d["single_gpu_frames_in_flight_ms"],
      d["slowest_rank_ms"], d["slowest_rank_period_ms"], d["projected_speedup_period_vs_single_gpu_frames_in_flight"], d["rank_period_ms"]))
print("bands", d.get("bands_period_balanced"))
for s in d.get("solo_fresh", []):
    print("rank %d: in process %.3f ms, fresh process %s ms (passes %s), streams %s" % (s["rank"], s["period_in_process_ms"], s.get("period_fresh_process_ms"), s.get("passes_ms"), json.dumps(s.get("stream_layout", s.get("error")))))
print("in-process stream layout:", json.dumps(d.get("stream_layout_in_process")))
b = d.get("bands_period_balanced") or d["bands_last_frame"]
per = d["rank_period_ms"]
order = sorted(range(len(per)), key=lambda q: -per[q])
open(sys.argv[1] + ".part", "w").write(",".join(str(x[0]) for x in b) + "," + str(b[-1][1]) + "\n%d %d\n" % (order[0], order[len(order) // 2]))
PY
PART=$(head -1 $O/emulate$N.json.part); RANKS=$(tail -1 $O/emulate$N.json.part)
echo "== 3. RESTIR_MGPU_PRIO sweep (main / indirect / filter), ranks $RANKS of partition $PART, fresh process each" | tee -a $O/summary.txt
for rep in 1 2; do for prio in +00 0+0 0++ 000 ++0; do for q in $RANKS; do
  RESTIR_MGPU_PRIO=$prio timeout 600 python bench.py --emulate-world $N --emulate-child "$q:$PART" --no-cpu-baseline 2> $O/child.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('prio $prio rank $q rep $rep: %.3f ms  passes %s  created %d' % (d['period_fresh_process_ms'], d['passes_ms'], d['stream_layout']['library_streams_created']))" | tee -a $O/summary.txt
done; done; done
