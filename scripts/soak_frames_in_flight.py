"""Soak: N consecutive frames of the benchmark scene with frames in flight (rt_set_overlap 2), twice, and once with every launch alone on one stream (overlap 0);
SHA-256 of every screen-space buffer after the last frame must agree between all three runs (a race between frames in flight shows as a differing digest).
    python scripts/soak_frames_in_flight.py [frames]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from helpers import abi, host, make_scene, frame_buffers
from restir_amd.renderer import Renderer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
W, H = 1920, 1080
sc, env = make_scene(abi.PROC_BISTRO_EXT, 1.0, 1, (2048, 1024))
st = host.default_state(W, H, sc, env)
eye, center, up, fov = sc.cameraPose()


def run(overlap):
    r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H); r.set_overlap(overlap)
    sc.setCamera(eye, center, up, fov); sc.updateCamera(W, H); sc.updateCamera(W, H)
    t0 = time.perf_counter()
    for f in range(N):
        st.time = 1000 + f
        if f % 50 == 0:   # a camera step every 50 frames: temporal reuse re-projects, history misses, static stretches in between
            sc.setCamera(eye + np.array([0.02 * (f // 50), 0.0, -0.01 * (f // 50)], dtype=np.float32), center, up, fov)
        sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
    r.sync(); dt = time.perf_counter() - t0
    dig = {abi.BUFFER_NAMES[b]: hashlib.sha256(r.readback(b).tobytes()).hexdigest()[:16] for b in frame_buffers(N - 1) + [abi.BUF_GBUFFER0, abi.BUF_GBUFFER1, abi.BUF_DIRECT_RESV0, abi.BUF_DIRECT_RESV1, abi.BUF_INDIRECT_RESV0, abi.BUF_INDIRECT_RESV1]}
    r.destroy()
    return dig, dt / N * 1e3


a, ta = run(2); b, tb = run(2); c, tc = run(0); d, td = run(3)      # (round 6: + three frames in flight, rt_set_overlap 3)
bad = [k for k in a if not (a[k] == b[k] == c[k] == d[k])]
print(f"{N} frames: in flight {ta:.3f} / {tb:.3f} ms per frame, three in flight {td:.3f}, serial {tc:.3f}; buffers compared {len(a)}, differing {len(bad)} {bad}")
sys.exit(1 if bad else 0)
