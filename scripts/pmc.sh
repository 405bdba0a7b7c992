#!/bin/bash
# PMC passes for the frame kernels (run on the GPU box): scripts/pmc.sh <tag>
# Counters are collected in their own runs (no trace domains besides kernel dispatch), one group per pass.
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 --output-format csv -d $OUT -o $1 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-run > /dev/null 2>&1; }
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
run sq2 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
run sq3 "SQ_LEVEL_WAVES SQ_CYCLES SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE"
run tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
run tcc2 "FETCH_SIZE"
run tcc3 "WRITE_SIZE"
python - <<PY
import csv, glob, collections, os
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(out + "/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[(k, row["Counter_Name"])] += 1
with open(out + "/summary.txt", "w") as fo:
    for k, d in sorted(agg.items()):
        if not k.startswith(("rt::", "void rt::")): continue
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write("   %-28s %16.1f per launch (%d launches)\n" % (c, v / calls[(k, c)], calls[(k, c)]))
import json
def key(k):   # kernel name with its template arguments, without namespaces: k_denoise_lds<false, true>
    k = k.replace("void ", "")
    head = k.split("<")[0]
    return head.split("::")[-1] + k[len(head):]
traffic = {key(k): {"FETCH_SIZE_KB": d.get("FETCH_SIZE", 0) / max(1, calls[(k, "FETCH_SIZE")]),
                                                              "WRITE_SIZE_KB": d.get("WRITE_SIZE", 0) / max(1, calls[(k, "WRITE_SIZE")]),
                                                              "INSTS_VALU": d.get("SQ_INSTS_VALU", 0) / max(1, calls[(k, "SQ_INSTS_VALU")]),
                                                              "THREAD_CYCLES_VALU": d.get("SQ_THREAD_CYCLES_VALU", 0) / max(1, calls[(k, "SQ_THREAD_CYCLES_VALU")])}
           for k, d in agg.items() if k.startswith(("rt::", "void rt::")) and "_cnt::" not in k}   # not the instrumented (counting) variants
frames = max(1, calls[("rt::base::k_direct_stage", "SQ_INSTS_VALU")])
traffic["_frame"] = {"INSTS_VALU": sum(d.get("SQ_INSTS_VALU", 0) for k, d in agg.items() if k.startswith(("rt::base::", "void rt::base::"))) / frames,
                     "note": "sum over the kernels of one frame (count-free variants), SQ_INSTS_VALU per launch x launches per frame"}
import hashlib
lib = os.path.join(os.environ["GRAFT_REPO_ROOT"], "cis-565-final-vr-raytracer_amd", "csrc", "librestir_hip.so")
traffic["_lib_sha256_16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]   # bench.py marks these numbers stale for any other build
json.dump(traffic, open(out + "/pmc_traffic.json", "w"), indent=1)
print(open(out + "/summary.txt").read())
PY
