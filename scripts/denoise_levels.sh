#!/bin/bash
# Per-level durations of the A-Trous kernels (serial schedule) for both filter implementations: scripts/denoise_levels.sh <tag>
TAG=${1:-dn}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  RESTIR_OVERLAP=0 RESTIR_DENOISE_TILE=$m rocprofv3 --kernel-trace --output-format csv -d $O/t$m -o kt -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-run > /dev/null 2> $O/err$m.log
  python - <<PY
import csv, glob, collections
f = glob.glob("$O/t$m/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_denoise" in r["Kernel_Name"] and "_cnt" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# per frame: geom, 4 x direct, geom(ind), 5 x indirect  (serial schedule) -> classify by template args + running index
seq = collections.defaultdict(list)
cnt = collections.Counter()
for r in rows:
    n = r["Kernel_Name"]
    ind = "Lb1E" in n or "<true" in n
    kind = "geom" if "geom" in n else "filt"
    key = (kind, ind)
    lvl = cnt[key] % (1 if kind == "geom" else (5 if ind else 4)); cnt[key] += 1
    seq[(kind, ind, lvl)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("tile=$m")
for k in sorted(seq):
    v = sorted(seq[k]); print("  %-5s %-8s level %d: median %.1f us (%d launches)" % (k[0], "indirect" if k[1] else "direct", k[2], v[len(v)//2], len(v)))
PY
done
