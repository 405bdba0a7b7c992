"""Latency profile of the traced stages: each 8-row tile strip of the 1080p benchmark frame launched alone (240 waves on 1024
SIMDs: the time of a strip is the time of its slowest wave).  Shows where the latency floor of small launches comes from."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer
W, H = 1920, 1080
sc, env = make_scene(abi.PROC_BISTRO_EXT, 1.0, 1, (2048, 1024))
st = host.default_state(W, H, sc, env)
r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
sc.updateCamera(W, H)
for f in range(6):
    st.time = 1000 + f; sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
r.sync()
def timed(fn, n=10):
    fn(); r.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    r.sync(); return (time.perf_counter() - t0) / n * 1e3
f = 7
step = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for y in range(0, H, step):
    td = timed(lambda: r.run_stage(st, f, abi.STAGE_DIRECT, 0, y, min(H, y + step)))
    ti = timed(lambda: r.run_stage(st, f, abi.STAGE_INDIRECT, 0, y // 2, min(H // 2, (y + 2 * step) // 2))) if (y // step) % 2 == 0 else 0.0
    print("rows", y, "direct", round(td, 3), "indirect(2 strips)", round(ti, 3), flush=True)
