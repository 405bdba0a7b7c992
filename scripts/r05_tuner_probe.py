"""round 5: why does rt_tune_stream_priorities rank the settings differently from a timed run?  Config 3, one process, python-timed loops before and after the C tuner."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer
W, H = 1920, 1080
kind = getattr(abi, sys.argv[1] if len(sys.argv) > 1 else "PROC_SPONZA_1K")
sc, env = make_scene(kind, 1.0, 1, (2048, 1024))
st = host.default_state(W, H, sc, env)
if kind == abi.PROC_SPONZA_1K: st.maxDepth = 2
r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
sc.updateCamera(W, H); sc.updateCamera(W, H); r.set_camera(sc.getCamera())
f = [0]
def timed(n=60, warm=8, pace=True):
    for _ in range(warm):
        st.time = 1000 + f[0]
        if pace: sc.updateCamera(W, H); r.set_camera(sc.getCamera())
        r.run(st, f[0]); f[0] += 1
    r.sync(); t0 = time.perf_counter()
    for _ in range(n):
        st.time = 1000 + f[0]
        if pace: sc.updateCamera(W, H); r.set_camera(sc.getCamera())
        r.run(st, f[0]); f[0] += 1
    r.sync(); return round((time.perf_counter() - t0) / n * 1e3, 4)
order = [(1, 1), (1, 0), (0, 1), (1, 1), (0, -1), (1, -1), (1, 1), (1, 0)]
if len(sys.argv) > 2 and sys.argv[2] == "default-first": order = [(1, 0), (1, 1), (0, 1), (0, -1), (1, -1), (1, 1), (1, 0)]
print("python loop, paced:", [(lv, (r.set_stream_priorities(*lv), timed())[1]) for lv in order], flush=True)
print("python loop, unpaced:", [(lv, (r.set_stream_priorities(*lv), timed(pace=False))[1]) for lv in [(1, 0), (1, 1), (0, 1)]], flush=True)
print("C tuner (24):", r.tune_stream_priorities(st, 24), flush=True)
f[0] = 0
print("python loop after the tuner:", [(lv, (r.set_stream_priorities(*lv), timed())[1]) for lv in [(1, 0), (1, 1), (0, 1)]], flush=True)
print("C tuner (60):", r.tune_stream_priorities(st, 60), flush=True)
