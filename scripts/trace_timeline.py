"""Kernel-trace timeline of the last frames with queue ids: python scripts/trace_timeline.py <kernel_trace.csv> [ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "rt::" not in n: continue
    k = n.split("(")[0].split("::")[-1].split("<")[0] + ("<ind>" if "<true" in n else "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "?"), r.get("Grid_Size", "?") if "Grid_Size" in r else r.get("Grid_Size_X", "?")))
ev.sort()
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
t1 = ev[-1][1]; win = [e for e in ev if e[0] >= t1 - span]
t0 = win[0][0]
for s, e, k, q, g in win:
    if k.startswith("k_denoise_lds") or k.startswith("k_denoise_geom"):
        continue
    print("%8.3f %8.3f  %6.3f  q%-3s grid %-8s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, g, k))
