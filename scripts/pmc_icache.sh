#!/bin/bash
# Instruction-cache behaviour of the frame kernels, pipelined (frames in flight) and serial: scripts/pmc_icache.sh <tag>
TAG=${1:-icache}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in 2 0; do
  RESTIR_OVERLAP=$mode rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -o ic$mode -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-run > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
for mode in ("2", "0"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for f in glob.glob("$OUT/ic%s*counter_collection.csv" % mode):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if "_cnt::" in k or "rt::" not in k: continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[(k, row["Counter_Name"])] += 1
    print("overlap mode", mode)
    for k, d in sorted(agg.items()):
        req, hit, miss = d.get("SQC_ICACHE_REQ", 0), d.get("SQC_ICACHE_HITS", 0), d.get("SQC_ICACHE_MISSES", 0)
        n = max(1, calls[(k, "SQC_ICACHE_REQ")])
        print("  %-44s icache req %12.0f  miss %11.0f (%.2f %%)  dup %10.0f | wait_inst_any/wave_cycles %.3f" % (k.replace("void ", "")[:44], req / n, miss / n, 100 * miss / max(1, req),
              d.get("SQC_ICACHE_MISSES_DUPLICATE", 0) / n, d.get("SQ_WAIT_INST_ANY", 0) / max(1, d.get("SQ_WAVE_CYCLES", 1))))
PY
