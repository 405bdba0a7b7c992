#!/bin/bash
# Round 6, item 3: why the persistent multi-bounce waves do not pay — wave-level VALU instructions and lane utilisation of k_indirect_stage per launch (one rocprofv3
# --pmc pass per setting, the bench's own frames) for RESTIR_IND_PERSIST = 0 (one wave per tile), 1, 2, 3.  usage (gpurun): bash scripts/r06_persist_pmc.sh <tag>
TAG=${1:-r06pp}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for p in 0 1 2 3; do
  RESTIR_IND_PERSIST=$p timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT -o p$p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-run > /dev/null 2>&1
done
python - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, os, sys
out = sys.argv[1]
print("%-8s %-28s %14s %14s %10s %12s %12s" % ("persist", "kernel", "INSTS_VALU", "THREAD_CYC", "lanes %", "waves", "wait_any %"))
for p in (0, 1, 2, 3):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/**/p{p}_counter_collection.csv", recursive=True) + glob.glob(f"{out}/p{p}_counter_collection.csv"):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            per[(k, row["Counter_Name"], int(row.get("Dispatch_Id", 0) or 0))] += float(row["Counter_Value"])
        for (k, c, d), v in per.items():
            agg[k][c].append(v)
    for k in sorted(agg):
        if "k_indirect_stage" not in k and "k_direct_stage" not in k: continue
        if "_cnt" in k: continue
        m = {c: sum(v[len(v) // 3:]) / max(1, len(v[len(v) // 3:])) for c, v in agg[k].items()}     # (drop the warm-up launches: the first third)
        iv, tc = m.get("SQ_INSTS_VALU", 0), m.get("SQ_THREAD_CYCLES_VALU", 0)
        print("%-8d %-28s %14.0f %14.0f %10.1f %12.0f %12.1f" % (p, k.split("::")[-1][:28], iv, tc, 100 * tc / max(1, 64 * iv), m.get("SQ_WAVES", 0), 100 * m.get("SQ_WAIT_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1))))
PY
