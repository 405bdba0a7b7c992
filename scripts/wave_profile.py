"""Per-wave cycle breakdown of the traced stages on a row band (measurement build: csrc/_ab/librestir_hip_prof.so, -DRT_WAVEPROF=1).

    RESTIR_HIP_LIB=.../csrc/_ab/librestir_hip_prof.so python scripts/wave_profile.py [W H] y0 y1 [y0 y1 ...]

For every band: the direct stage and the indirect stage launched alone on the band (serial schedule, as a rank of the row-tiled frame runs them),
then per stage: launch time, the slowest waves with their round counts / cycles by kind of round (node step, triangle step, cooperative tail),
and how alpha candidates were resolved (opacity micro-map vs texture).  Builds the numbers DESIGN.md §7 quotes for the latency floor of a band."""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer, hip_lib

args = [int(a) for a in sys.argv[1:]]
W, H = (1920, 1080)
if len(args) % 2 == 0 and len(args) >= 2 and args[0] >= 640 and args[1] >= 360 and len(args) >= 4:
    W, H = args[0], args[1]; args = args[2:]
bands = list(zip(args[0::2], args[1::2])) or [(496, 528), (528, 576)]
sc, env = make_scene(getattr(abi, os.environ.get("WAVE_PROFILE_KIND", "PROC_BISTRO_EXT")), 1.0, 1, (2048, 1024))   # WAVE_PROFILE_KIND=PROC_BISTRO_EXT_REAL: the real-footprint scene
st = host.default_state(W, H, sc, env)
r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
r.set_overlap(0)
LAT = os.environ.get("WAVE_PROFILE_LAT") == "1"     # the latency build of the traced kernels instead of the throughput build
r.set_traversal(abi.TRAVERSAL_LATENCY if LAT else abi.TRAVERSAL_THROUGHPUT)
sc.updateCamera(W, H)
for f in range(6):
    st.time = 1000 + f; sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
r.sync()
L = hip_lib()
L.rt_debug_wave_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
NREC = 65536
buf = np.zeros((NREC, 16), dtype=np.uint32)


def prof():
    rc = L.rt_debug_wave_profile(r._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    assert rc == 0, rc
    return buf.copy()


def timed(fn, n=5):
    fn(); r.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    r.sync(); return (time.perf_counter() - t0) / n * 1e3


def report_wide(name, rec, ms):
    acc = rec[65535].astype(np.float64); rec = rec[:65535]
    if acc[7] > 0:
        names = ["record+AlphaRec arrive", "intersection/box/micro-map", "texel addresses", "texels arrive", "filter+draw", "rejoin", "group minimum + update"]
        print("   triangle step, cycles per step by phase (all waves): " + ", ".join(f"{n} {16 * acc[k] / acc[7]:.0f}" for k, n in enumerate(names)) + f"; sum {16 * acc[:7].sum() / acc[7]:.0f}; whole round {16 * acc[8] / acc[7]:.0f}")
    if acc[12] > 0:
        print("   node step, cycles per step (all waves): " + ", ".join(f"{n} {16 * acc[9 + k] / acc[12]:.0f}" for k, n in enumerate(["stack pop + select", "node arrives", "test + reduce"])))
    rec = rec[rec[:, 2] > 0]
    if len(rec) == 0:
        print(name, "no records"); return
    cyc = rec[:, 2].astype(np.float64)
    ghz = np.median(cyc / np.maximum(1, rec[:, 15]) * 0.1)
    order = np.argsort(-cyc)
    us = lambda c: c / ghz / 1e3
    print(f"== {name} (wide build): launch {ms:.3f} ms, {len(rec)} workgroups, clock {ghz:.2f} GHz, slowest workgroup {us(cyc[order[0]]):.1f} us")
    print("   slowest workgroups: tile | total us | raygen | primary trace | pre-shadow shading | shadow trace | post | rounds node-only/with-tris (slowest wave) | cycles per round")
    for i in order[:12]:
        q = rec[i].astype(np.float64)
        print(f"   ({int(q[0]):4d},{int(q[1]):3d}) | {us(q[2]):7.1f} | {us(q[7]):5.1f} | {us(q[3]):7.1f} | {us(q[10]):6.1f} | {us(q[4]):7.1f} | {us(q[11]):5.1f} | {int(q[5]):4d}/{int(q[6]):4d} | {q[8] / max(1, q[5]):6.0f}/{q[9] / max(1, q[6]):6.0f} | ray slots in use {100 * q[12] / max(1, 8 * q[13]):.0f} % of {int(q[13])} wave-rounds")
    m = rec.astype(np.float64).mean(axis=0)
    print(f"   ray slots in use over all workgroups: {100 * rec[:, 12].sum() / max(1, 8 * rec[:, 13].sum()):.0f} %")
    print(f"   mean workgroup: total {us(m[2]):.1f} us = raygen {us(m[7]):.1f} + primary {us(m[3]):.1f} + pre {us(m[10]):.1f} + shadow {us(m[4]):.1f} + post {us(m[11]):.1f}")
    pct = np.percentile(cyc, [50, 90, 99, 100])
    print(f"   workgroup time percentiles (us): p50 {us(pct[0]):.1f} p90 {us(pct[1]):.1f} p99 {us(pct[2]):.1f} max {us(pct[3]):.1f}; sum over workgroups / 512 concurrent = {us(cyc.sum()) / 512 / 1e3:.3f} ms")


def report(name, rec, ms):
    hist = rec[65534].astype(np.float64); rec = rec[:65534]     # launch-wide histograms of the rounds (csrc/stage_common.h waveProfFlush)
    if hist.sum() > 0:
        hl, he = hist[:8], hist[8:]
        mid = np.arange(8) * 8 + 4.5
        print("   rounds by LIVE lanes (1-8 .. 57-64), %: " + " ".join(f"{100 * v / hl.sum():.1f}" for v in hl) + f"   mean {(hl * mid).sum() / hl.sum():.1f} lanes")
        print("   rounds by EXECUTING lanes (the majority kind), %: " + " ".join(f"{100 * v / he.sum():.1f}" for v in he) + f"   mean {(he * mid).sum() / he.sum():.1f} lanes = {100 * (he * mid).sum() / he.sum() / 64:.1f} % of the wave")
    rec = rec[rec[:, 2] > 0]
    if len(rec) == 0:
        print(name, "no records"); return
    cyc = rec[:, 2].astype(np.float64)
    ghz = cyc / np.maximum(1, rec[:, 15]) * 0.1          # cycles per 10 ns tick -> GHz
    order = np.argsort(-cyc)
    print(f"== {name}: launch {ms:.3f} ms, {len(rec)} waves, clock {np.median(ghz):.2f} GHz (median), slowest wave {cyc[order[0]] / np.median(ghz) / 1e6:.3f} ms")
    tot = rec.astype(np.float64).sum(axis=0)
    print(f"   all waves: rounds node/tri/coop {tot[5]:.0f}/{tot[6]:.0f}/{tot[7]:.0f}  cycles per round {tot[8] / max(1, tot[5]):.0f}/{tot[9] / max(1, tot[6]):.0f}/{tot[10] / max(1, tot[7]):.0f}"
          f"  alpha candidates: texture {tot[13]:.0f}, micro-map {tot[14]:.0f} ({100 * tot[13] / max(1, tot[13] + tot[14]):.1f} % texture)")
    print("   slowest waves: tile(x,y) | wave kcyc | closest kcyc | any kcyc | rounds n/t/c | cyc per round n/t/c | max nodes, tris of a lane | alpha tex/omm")
    for i in order[:12]:
        q = rec[i]
        print(f"   ({q[0] & 0xffff:4d},{q[1]:3d}){' MB' if q[0] & 0x10000 else (' K3' if q[0] & 0x20000 else '   ')} | {q[2] / 1e3:8.1f} | {q[3] / 1e3:8.1f} | {q[4] / 1e3:8.1f} | {q[5]:4d}/{q[6]:4d}/{q[7]:4d} | "
              f"{q[8] / max(1, q[5]):6.0f}/{q[9] / max(1, q[6]):6.0f}/{q[10] / max(1, q[7]):6.0f} | {q[11]:4d},{q[12]:4d} | {q[13]}/{q[14]}")
    # how much of the slowest wave is traversal
    q = rec[order[0]]
    print(f"   slowest wave: traversal {100.0 * (q[3] + q[4]) / q[2]:.0f} % of its cycles; rounds account for {100.0 * (q[8] + q[9] + q[10]) / max(1, q[3] + q[4]):.0f} % of the traversal cycles")
    hist = np.percentile(cyc, [50, 90, 99, 100]) / np.median(ghz) / 1e6
    print(f"   wave time percentiles (ms): p50 {hist[0]:.3f} p90 {hist[1]:.3f} p99 {hist[2]:.3f} max {hist[3]:.3f}")


f = 7
st.time = 1000 + f
for (y0, y1) in bands:
    print(f"#### band rows {y0}..{y1} of {W}x{H}")
    for stage, nm, a, b in ((abi.STAGE_DIRECT, "direct", y0, y1), (abi.STAGE_INDIRECT, "indirect", y0 // 2, y1 // 2)):
        ms = timed(lambda: r.run_stage(st, f, stage, 0, a, b))
        prof()
        r.run_stage(st, f, stage, 0, a, b); r.sync()
        (report_wide if (LAT and nm == "direct") else report)(f"{nm} rows {a}..{b}", prof(), ms)
