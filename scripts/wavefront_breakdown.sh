#!/bin/bash
# Per-kernel durations of the wavefront organisation (serial schedule): how the direct / indirect stages split into trace and shade work
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-wfb}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
RESTIR_OVERLAP=0 RESTIR_PIPELINE=wavefront rocprofv3 --kernel-trace --stats --output-format csv -d $O -o wf -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --profile-run > $O/bench.json 2> $O/err.log
python - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rt::" in r["Name"] and "_cnt" not in r["Name"]]
frames = 11
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print("%-60s calls/frame %5.1f  ms/frame %.3f  avg us %.1f" % (r["Name"].split("(")[0][:60], int(r["Calls"]) / frames, float(r["TotalDurationNs"]) / frames / 1e6, float(r["AverageNs"]) / 1e3))
PY
