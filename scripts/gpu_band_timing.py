"""Wall time of each stage restricted to one rank's band (1 GPU, communication stubbed): where does a tiled frame's time go?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer
from restir_amd import tiled
W, H = 1920, 1080
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc, env = make_scene(abi.PROC_BISTRO_EXT, 1.0, 1, (2048, 1024))
st = host.default_state(W, H, sc, env)
r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
sc.updateCamera(W, H)
for f in range(6):
    st.time = 1000 + f; sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
r.sync()
B = tiled.band_height(H, world)
def timed(fn, n=20):
    fn(); r.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    r.sync(); return (time.perf_counter() - t0) / n * 1e3
for rank in range(world):
    y0, y1 = min(rank * B, H), min((rank + 1) * B, H)
    h0, h1 = y0 // 2, y1 // 2
    f = 7
    t = {}
    t["direct"] = timed(lambda: r.run_stage(st, f, abi.STAGE_DIRECT, 0, y0, y1))
    t["indirect"] = timed(lambda: r.run_stage(st, f, abi.STAGE_INDIRECT, 0, h0, h1))
    def dd():
        for l in range(4): r.run_stage(st, f, abi.STAGE_DENOISE_DIRECT, l, max(0, y0 - tiled.DIRECT_GROW[l]), min(H, y1 + tiled.DIRECT_GROW[l]))
    def di():
        for l in range(5): r.run_stage(st, f, abi.STAGE_DENOISE_INDIRECT, l, max(0, h0 - tiled.INDIRECT_GROW[l]), min(H // 2, h1 + tiled.INDIRECT_GROW[l]))
    t["denoise_d"] = timed(dd); t["denoise_i"] = timed(di)
    t["compose"] = timed(lambda: r.run_stage(st, f, abi.STAGE_COMPOSE, 0, y0, y1))
    t["miss_flag"] = timed(lambda: r.history_miss())
    print("rank", rank, "rows", y0, y1, {k: round(v, 3) for k, v in t.items()}, "sum", round(sum(t.values()), 3), flush=True)
