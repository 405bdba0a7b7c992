"""Randomised sweep of the native multi-GPU context against the single-GPU frame (every buffer, bit for bit):
   python scripts/mgpu_fuzz.py <cases> <seed>   — random scene / size / rank count / ReSTIRState / camera motion / balance"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import abi, host, make_scene, frame_buffers
from restir_amd.renderer import Renderer, MultiGpuRenderer
import torch

cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
KINDS = [(abi.PROC_CORNELL, 1.0), (abi.PROC_HELMET, 0.04), (abi.PROC_SPONZA, 0.02), (abi.PROC_BISTRO_EXT, 0.008), (abi.PROC_BISTRO_INT, 0.01)]
ndev = max(1, torch.cuda.device_count())
bad = 0
for ci in range(cases):
    kind, scale = KINDS[rng.integers(len(KINDS))]
    W, H = int(rng.integers(40, 400)), int(rng.integers(130, 420))
    world = int(rng.integers(2, min(8, (H + 15) // 16) + 1))
    envk = int(rng.integers(2))
    sc, env = make_scene(kind, scale, int(rng.integers(1, 1000)), (64, 32) if envk else None)
    st = host.default_state(W, H, sc, env)
    st.ReSTIRState = int(rng.integers(0, 5)); st.maxDepth = int(rng.integers(1, 5)); st.denoise = int(rng.integers(0, 2)); st.modulate = int(rng.integers(0, 2))
    if not envk: st.environmentProb = 0.0; st.fireflyClampThreshold = 50.0
    balance = bool(rng.integers(0, 2)); frames = int(rng.integers(2, 6))
    vel = rng.normal(scale=0.08, size=3).astype(np.float32) * rng.integers(0, 2)
    lift = float(rng.choice([0.0, 0.0, 6.0]))
    desc = sc.desc(env)
    ref = Renderer().setup(0); ref.load_scene(desc); ref.update(W, H)
    m = MultiGpuRenderer().setup([i % ndev for i in range(world)]); m.load_scene(desc); m.update(W, H); m.set_balance(balance)
    eye, center, up, fov = sc.cameraPose(); sc.updateCamera(W, H)
    info = dict(case=ci, kind=int(kind), W=W, H=H, world=world, restir=st.ReSTIRState, depth=st.maxDepth, den=st.denoise, balance=balance, frames=frames, lift=lift)
    ok = True
    queued = bool(rng.integers(0, 2))     # compare only the last frame: the frames before it stay queued (frames in flight, look-ahead issue of the next direct stage)
    info["queued"] = queued
    for f in range(frames):
        st.time = 3000 + f
        sc.setCamera(eye + vel * f, center + np.array([0, lift * f, 0], dtype=np.float32), up, fov); sc.updateCamera(W, H)
        cam = sc.getCamera(); ref.set_camera(cam); m.set_camera(cam); ref.run(st, f); m.run(st, f)
        if queued and f < frames - 1:
            continue
        bufs = [b for b in frame_buffers(f) if b not in (abi.BUF_DENOISE_DIR_A, abi.BUF_DENOISE_DIR_B, abi.BUF_DENOISE_IND_A, abi.BUF_DENOISE_IND_B)]
        if st.ReSTIRState in (2, 4): bufs.append(abi.BUF_DIRECT_RESV_TEMP)
        for b in bufs:
            if not np.array_equal(m.readback(b), ref.readback(b)):
                ok = False; print("MISMATCH", info, "frame", f, abi.BUFFER_NAMES[b], flush=True)
    bad += 0 if ok else 1
    m.destroy(); ref.destroy()
print("mgpu fuzz cases", cases, "seed", seed, "mismatching", bad, flush=True)
sys.exit(1 if bad else 0)
