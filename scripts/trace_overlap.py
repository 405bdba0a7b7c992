"""Reads a rocprofv3 kernel trace CSV and prints, for the last frames, how the stage kernels overlap in time."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "rt::" not in n: continue
    k = n.split("(")[0].split("::")[-1].split("<")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
ev.sort()
t_mid = (ev[0][0] + ev[-1][1]) // 2 if len(sys.argv) < 3 else ev[0][0] + int(float(sys.argv[2]) * 1e6)
win = [e for e in ev if t_mid <= e[0] < t_mid + 20_000_000]   # 20 ms from the middle of the run (or from argv[2] ms after the first kernel)
t0 = win[0][0]
busy = collections.Counter(); total = win[-1][1] - t0
for s, e, k in win: busy[k] += e - s
print("window ms", total / 1e6, {k: round(v / 1e6, 2) for k, v in busy.items()})
# union coverage and pairwise overlap of direct vs indirect
def union(iv):
    iv = sorted(iv); out = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: out += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return out + ce - cs
allv = [(s, e) for s, e, k in win]
print("GPU busy (any rt kernel) %.1f %% of the window" % (100 * union(allv) / total))
d = [(s, e) for s, e, k in win if k == "k_direct_stage"]; i = [(s, e) for s, e, k in win if k == "k_indirect_stage"]
ov = 0
for s1, e1 in d:
    for s2, e2 in i: ov += max(0, min(e1, e2) - max(s1, s2))
print("direct busy %.2f ms, indirect busy %.2f ms, both at once %.2f ms" % (sum(e - s for s, e in d) / 1e6, sum(e - s for s, e in i) / 1e6, ov / 1e6))
for s, e, k in win[-40:]:
    print("%9.3f %9.3f %s" % ((s - t0) / 1e6, (e - t0) / 1e6, k))
