cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_pair; mkdir -p $O
for rep in 1 2; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$rep -o stats -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --profile-run > $O/prof_bench$rep.json 2> $O/prof$rep.err
python - $O/prof_bench$rep.json $O/prof$rep <<'PY'
import json, sys, csv, glob
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f = glob.glob(sys.argv[2] + "/**/stats_kernel_stats.csv", recursive=True)[0]
k = [r for r in csv.DictReader(open(f)) if "rt::base::k_direct_stage" in r["Name"] or "rt::base::k_indirect_stage" in r["Name"]]
tr = glob.glob(sys.argv[2] + "/**/stats_kernel_trace.csv", recursive=True)[0]
rows = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(tr)) if "rt::base::k_direct_stage" in r["Kernel_Name"]])
last = rows[-100:]
print("under rocprof: ms/frame", d["ms_per_step"], "| events", d["events_in_this_run"]["stage_ms_per_frame"], "| kernel stats", [(r["Name"].split("::")[2].split("(")[0], r["Calls"], round(float(r["AverageNs"]) / 1e6, 4)) for r in k], "| trace, the 100 timed launches of k_direct_stage: %.4f ms" % (sum(e - s for s, e in last) / len(last) / 1e6))
PY
done
cd $R; timeout 600 python bench.py --no-cpu-baseline > $O/bench_plain.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_plain.json').read().strip().splitlines()[-1]); print('plain run: ms/frame', d['ms_per_step'], 'events', d['roofline']['stage_ms_per_frame'])"
