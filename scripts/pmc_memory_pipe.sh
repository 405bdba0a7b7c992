#!/bin/bash
# How busy is the vector-memory pipe (TA / TCP / TD) under the traced kernels?  serial schedule, separate --pmc passes.  scripts/pmc_memory_pipe.sh <tag>
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-mempipe}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
# (a set with TA_ADDR_STALLED_BY_TC_CYCLES_sum / TA_DATA_STALLED_BY_TC_CYCLES_sum hung rocprofv3 on this pool: not collected; every pass runs under `timeout`)
for set in "TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1))
  RESTIR_OVERLAP=0 timeout 120 rocprofv3 --pmc $set --output-format csv -d $O -o m$i -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --profile-run > /dev/null 2> $O/err$i.log
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$O/m*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "rt::base::" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    if not any(s in k for s in ("k_direct_stage", "k_indirect_stage", "k_denoise<false")): continue
    print(k)
    for c in sorted(agg[k]): print("   %-44s %16.1f per launch" % (c, agg[k][c] / max(1, n[(k, c)])))
PY
