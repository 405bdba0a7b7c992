"""GPU busy fraction of one solo rank's frames-in-flight period from a rocprofv3 kernel trace: python scripts/solo_busy.py <kernel_trace.csv> [last ms]
union of kernel intervals / wall time, the gaps with no kernel running, and per-kernel-kind busy time"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]) for r in rows if "rt::" in r["Kernel_Name"]]
ev.sort()
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 30e6
t1 = ev[-1][1]; win = [e for e in ev if e[0] >= t1 - span]
t0 = win[0][0]
busy = 0; cur_s, cur_e = win[0][0], win[0][1]; gaps = []
for s, e, k in win[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - t0
per = collections.Counter()
for s, e, k in win: per[k] += e - s
nd = sum(1 for s, e, k in win if k == "k_direct_stage")
print("window %.2f ms, %d direct stages => %.3f ms per frame; GPU busy (any kernel) %.1f %%; idle %.3f ms per frame in %d gaps (largest %.3f ms)" % (wall / 1e6, nd, wall / 1e6 / max(1, nd), 100.0 * busy / wall, (wall - busy) / 1e6 / max(1, nd), len(gaps), max(gaps or [0]) / 1e6))
for k, v in per.most_common(): print("   %-22s %.3f ms per frame" % (k, v / 1e6 / max(1, nd)))
