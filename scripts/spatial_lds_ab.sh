#!/bin/bash
# A/B of the LDS neighbour tile in k_direct_spatial (RESTIR_SPATIAL_LDS=0|1): kernel time and VMEM / wait counters, serial schedule
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-spatial_ab}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in 0 1; do
  RESTIR_OVERLAP=0 RESTIR_SPATIAL_LDS=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t${v}_$rep -- python $GRAFT_REPO_ROOT/scripts/spatial_lds_ab.py 24 > /dev/null 2> $O/err.log
done; done
for v in 0 1; do
  RESTIR_OVERLAP=0 RESTIR_SPATIAL_LDS=$v rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O -o p$v -- python $GRAFT_REPO_ROOT/scripts/spatial_lds_ab.py 6 > /dev/null 2>> $O/err.log
done
python - <<PY
import csv, glob, collections
for v in (0, 1):
    for rep in (1, 2):
        for f in glob.glob("$O/t%d_%d*kernel_stats.csv" % (v, rep)):
            for r in csv.DictReader(open(f)):
                if "k_direct_spatial" in r["Name"] or "k_direct_stage" in r["Name"]:
                    print("LDS tile %d run %d  %-40s calls %4s  avg us %8.2f" % (v, rep, r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3))
    agg = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob("$O/p%d*counter_collection.csv" % v):
        for r in csv.DictReader(open(f)):
            if "k_direct_spatial" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("LDS tile %d  per launch: " % v + "  ".join("%s %.0f" % (k, agg[k] / max(1, n[k])) for k in sorted(agg)))
PY
