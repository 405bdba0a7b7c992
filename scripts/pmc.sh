#!/bin/bash
# PMC passes for the frame kernels (run on the GPU box): scripts/pmc.sh <tag> [<workload key> [bench.py arguments ...]]
#   scripts/pmc.sh r04/pmc                              the headline workload (config 4, static camera)   -> key "config4"
#   scripts/pmc.sh r04/pmc3 config3 --config 3          any other bench.py workload under its own key
# Counters are collected in their own runs (no trace domains besides kernel dispatch), one group per pass, every pass under `timeout`.
# The per-kernel figures go to <out>/summary.txt and are MERGED into profiles/pmc_traffic.json under workloads[<key>] (the file is stamped with the hash of the
# library the counters were collected on; entries of another build are dropped).  bench.py reads the entry of the workload it runs.
# Frames (round 5, verdict item 6): the counters must describe the frames the bench line times.  Static camera: every frame is the same work, 4 timed frames after
# 2 warm-up frames.  Moving camera (--moving-camera, --config 5): the bench's own --steps / --warmup (100 / 20 unless PMC_STEPS / PMC_WARMUP say otherwise), i.e.
# the same frame indices and camera poses, and the per-launch averages are taken over the launches of the TIMED frames only (the warm-up launches are dropped
# by dispatch order).  bench.py uses an entry of a moving workload only when its frames match the run's.
TAG=${1:-pmc}; KEY=${2:-config4}; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="$*"
# key "auto": the key bench.py itself uses for these arguments (config, camera, scene footprint)
[ "$KEY" == "auto" ] && KEY=$(python $GRAFT_REPO_ROOT/bench.py --print-workload-key $BENCH_ARGS 2>/dev/null | tail -1)
STEPS=${PMC_STEPS:-4}; WARMUP=${PMC_WARMUP:-2}
case " $BENCH_ARGS " in *" --moving-camera "*|*" --config 5 "*) STEPS=${PMC_STEPS:-100}; WARMUP=${PMC_WARMUP:-20};; esac
run() { local name=$1 ctrs=$2; shift; shift; env "$@" timeout 600 rocprofv3 --pmc $ctrs --output-format csv -d $OUT -o $name -- python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup $WARMUP --no-cpu-baseline --profile-run $BENCH_ARGS > /dev/null 2>&1; }
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" X_=1
run sq2 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" X_=1
run sq3 "SQ_LEVEL_WAVES SQ_CYCLES SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE" X_=1
run sq4 "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_INSTS_SMEM" X_=1      # (scratch_ and global_ accesses are FLAT-encoded: SQ_INSTS_FLAT holds both)
run tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" X_=1
run tcc2 "FETCH_SIZE" X_=1
run tcc3 "WRITE_SIZE" X_=1
# the same write counter with the WHOLE traversal stack in LDS (no HBM overflow area): the difference is what the short stacks of the frames in flight write
run tcc3full "WRITE_SIZE" RESTIR_STACK_LDS=64
python - "$KEY" "$BENCH_ARGS" "$STEPS" "$WARMUP" <<PY
import csv, glob, collections, os, sys, json, hashlib
out = "$OUT"
key, bench_args, steps, warmup = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(out + "/*counter_collection.csv"):
    full = "tcc3full" in os.path.basename(f)
    rows = collections.defaultdict(list)     # (kernel, counter) -> [(dispatch id, value)]: one pass = one process, dispatch ids are its launch order
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        c = row["Counter_Name"] + ("_FULL_LDS_STACK" if full else "")
        rows[(k, c)].append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
    for (k, c), lst in rows.items():
        lst.sort()
        # a counter may be reported once per dispatch or once per dispatch and dimension (XCD / SE instances): group by dispatch id first
        per = collections.OrderedDict()
        for d, v in lst: per[d] = per.get(d, 0.0) + v
        vals = list(per.values())
        per_frame = len(vals) / float(steps + warmup)
        if "_cnt::" not in k and per_frame >= 1 and abs(per_frame - round(per_frame)) < 1e-9:
            vals = vals[int(round(per_frame)) * warmup:]      # the launches of the timed frames only
        agg[k][c] += sum(vals)
        calls[(k, c)] += len(vals)
with open(out + "/summary.txt", "w") as fo:
    fo.write("workload: %s (bench.py %s)\n" % (key, bench_args))
    for k, d in sorted(agg.items()):
        if not k.startswith(("rt::", "void rt::")): continue
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write("   %-36s %16.1f per launch (%d launches)\n" % (c, v / calls[(k, c)], calls[(k, c)]))
def kname(k):   # kernel name with its template arguments, without namespaces: k_denoise_lds<false, true>
    k = k.replace("void ", "")
    head = k.split("<")[0]
    return head.split("::")[-1] + k[len(head):]
def per(k, d, c): return d.get(c, 0) / max(1, calls[(k, c)])
traffic = {kname(k): {"FETCH_SIZE_KB": per(k, d, "FETCH_SIZE"), "WRITE_SIZE_KB": per(k, d, "WRITE_SIZE"), "WRITE_SIZE_KB_full_lds_stack": per(k, d, "WRITE_SIZE_FULL_LDS_STACK"),
                      "INSTS_VALU": per(k, d, "SQ_INSTS_VALU"), "THREAD_CYCLES_VALU": per(k, d, "SQ_THREAD_CYCLES_VALU"),
                      "INSTS_VMEM_WR": per(k, d, "SQ_INSTS_VMEM_WR"), "INSTS_VMEM_RD": per(k, d, "SQ_INSTS_VMEM_RD"), "INSTS_FLAT": per(k, d, "SQ_INSTS_FLAT"),
                      "WAIT_ANY": per(k, d, "SQ_WAIT_ANY"), "WAVE_CYCLES": per(k, d, "SQ_WAVE_CYCLES"),
                      "L2_HIT": per(k, d, "TCC_HIT_sum"), "L2_REQ": per(k, d, "TCC_REQ_sum")}
           for k, d in agg.items() if k.startswith(("rt::", "void rt::")) and "_cnt::" not in k}   # not the instrumented (counting) variants
frames = max(1, calls[("rt::base::k_direct_stage", "SQ_INSTS_VALU")])
traffic["_frame"] = {"INSTS_VALU": sum(d.get("SQ_INSTS_VALU", 0) for k, d in agg.items() if k.startswith(("rt::base::", "void rt::base::"))) / frames,
                     "note": "sum over the kernels of one frame (count-free variants), SQ_INSTS_VALU per launch x launches per frame"}
traffic["_bench_args"] = bench_args + " --steps %d --warmup %d" % (steps, warmup)
traffic["_frames"] = {"warmup": warmup, "steps": steps, "averaged_over": "the launches of the timed frames (warm-up launches dropped by dispatch order)"}
root = os.environ["GRAFT_REPO_ROOT"]
lib = os.path.join(root, "cis-565-final-vr-raytracer_amd", "csrc", "librestir_hip.so")
sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]   # bench.py marks these numbers stale for any other build
path = os.path.join(root, "profiles", "pmc_traffic.json")
try:
    allw = json.load(open(path))
    if allw.get("_lib_sha256_16") != sha or "workloads" not in allw: allw = {}
except (OSError, ValueError):
    allw = {}
allw.setdefault("workloads", {})[key] = traffic
allw["_lib_sha256_16"] = sha
if key in ("config4", "config4_real"):   # the headline workload also at top level (the layout rounds 2-3 committed; since round 5 the headline is the real footprint)
    for k, v in traffic.items(): allw[k] = v
json.dump(allw, open(path, "w"), indent=1)
json.dump(allw, open(out + "/pmc_traffic.json", "w"), indent=1)
print(open(out + "/summary.txt").read())
PY
