#!/bin/bash
# VALU instructions per kernel of the wavefront organisation (serial): how the stages' instruction counts split into shading and traversal
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-wfvalu}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
RESTIR_OVERLAP=0 RESTIR_PIPELINE=wavefront rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O -o wf -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --profile-run > $O/bench.json 2> $O/err.log
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$O/wf*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "rt::" not in k or "_cnt" in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
frames = 5
tot = sum(d["SQ_INSTS_VALU"] for d in agg.values())
for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
    print("%-44s launches/frame %5.1f  VALU M/frame %8.1f (%4.1f %%)  lane util %.2f" % (k[-44:], n[(k, "SQ_INSTS_VALU")] / frames, d["SQ_INSTS_VALU"] / frames / 1e6, 100 * d["SQ_INSTS_VALU"] / tot,
          d["SQ_THREAD_CYCLES_VALU"] / max(1.0, 64 * d["SQ_INSTS_VALU"])))
PY
