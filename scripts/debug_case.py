"""Re-renders one case of tests/test_gpu_fuzz.py (seed, case index) on the oracle and on both kernel organisations and prints where the
indirect reservoirs differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_fuzz as F
from helpers import abi, host, make_scene, RendererBackend
from restir_amd.renderer import Renderer
from oracle.binding import Oracle
seed, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for _ in range(case): F._skip_case(rng)
kind, scale = F.KINDS[rng.integers(len(F.KINDS))]
W, H = int(rng.integers(33, 260)), int(rng.integers(17, 150))
env_kind = rng.integers(3)
sc, env = make_scene(kind, scale, int(rng.integers(1, 1000)), (64, 32) if env_kind == 1 else None)
st = host.default_state(W, H, sc, env)
st.maxDepth = int(rng.integers(1, 6)); st.RISSampleNum = int(rng.integers(1, 9)); st.reservoirClamp = int(rng.integers(1, 100))
st.ReSTIRState = int(rng.integers(0, 5)); st.MIS = int(rng.integers(0, 2)); st.denoise = int(rng.integers(0, 2)); st.modulate = int(rng.integers(0, 2))
st.hdrMultiplier = float(rng.choice([1.0, 0.5, 3.0])); st.debugging_mode = int(rng.choice([0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9]))
if env_kind != 1:
    st.environmentProb = 0.0 if env_kind == 0 else 0.5; st.fireflyClampThreshold = float(rng.choice([5.0, 50.0, 1e4])); st.envMapLuminIntegInv = 0.0
st.sigLuminDirect = float(rng.choice([0.4, 0.05, 3.0, 1e-7])); st.sigDepthIndirect = float(rng.choice([1.0, 0.2, 2e6]))
latency = bool(rng.integers(0, 2))
print("case", dict(kind=int(kind), W=W, H=H, env=int(env_kind), depth=st.maxDepth, M=st.RISSampleNum, restir=st.ReSTIRState, mis=st.MIS, dbg=st.debugging_mode, latency=latency))
desc = sc.desc(env)
o = Oracle(0); o.upload_scene(desc); o.resize(W, H)
outs = {}
sc.updateCamera(W, H); st.time = 77; sc.updateCamera(W, H); cam = sc.getCamera()
o.set_camera(cam); o.render_frame(st, 0)
outs["oracle"] = o.readback(abi.BUF_INDIRECT_RESV0).view(np.uint32).reshape(-1, 19)
for name, wf in (("throughput", 1), ("latency", 2)):
    r = Renderer().setup(0); r.load_scene(desc); r.update(W, H); r.set_traversal(wf)
    r.set_camera(cam); r.run(st, 0)
    outs[name] = r.readback(abi.BUF_INDIRECT_RESV0).view(np.uint32).reshape(-1, 19)
    r.destroy()
for a, b in (("oracle", "throughput"), ("oracle", "latency"), ("throughput", "latency")):
    d = np.nonzero((outs[a] != outs[b]).any(1))[0]
    print(a, "vs", b, ":", len(d), "reservoirs differ", d[:5])
    for i in d[:2]:
        x, y = i % (W // 2), i // (W // 2)
        print("  pixel", x, y, "\n   ", a, outs[a][i].view(np.float32)[:16], outs[a][i][16:], "\n   ", b, outs[b][i].view(np.float32)[:16], outs[b][i][16:])
# ---- the first differing pixel: re-trace the GPU's bounce ray with the oracle's BVH and with its brute-force tracer ----
d = np.nonzero((outs["oracle"] != outs["fused"]).any(1))[0]
if len(d):
    g = outs["fused"][d[0]].view(np.float32)
    xv, nv, xs, ns = g[3:6].astype(np.float64), g[6:9].astype(np.float64), g[9:12].astype(np.float64), g[12:15].astype(np.float64)
    dirv = xs - xv; dist = np.linalg.norm(dirv); dirv /= dist
    rays = np.zeros((1, 8), np.float32); rays[0, 0:3] = xv + nv * 1e-4; rays[0, 3:6] = dirv; rays[0, 6] = 1e28
    print("GPU sample: xv", xv, "nv", nv, "xs", xs, "ns", ns, "dist", dist)
    print("oracle BVH  closest:", o.trace_closest(rays)); print("oracle brute closest:", o.trace_closest(rays, brute=True))
    np.save(os.path.join(ROOT, "gpurun_out", "debug_resv_fused.npy"), outs["fused"][d[0]]); np.save(os.path.join(ROOT, "gpurun_out", "debug_resv_oracle.npy"), outs["oracle"][d[0]])
