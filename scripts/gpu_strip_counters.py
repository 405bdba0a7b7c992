"""Traversal counters of single 8-row strips of the benchmark frame's direct stage (which strips are slow, and why)."""
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
r.set_counting(True)
prev = None
def snap():
    c = r.counters()
    return dict(closest=c.closestHitRays, any=c.anyHitRays, nodes=c.nodesVisited, tris=c.trisTested, rounds=c.laneRounds, live=c.laneLiveRounds)
for y in [int(a) for a in sys.argv[1:]]:
    a = snap(); r.run_stage(st, 7, abi.STAGE_DIRECT, 0, y, y + 8); r.sync(); b = snap()
    d = {k: b[k] - a[k] for k in a}
    rays = max(1, d["closest"] + d["any"])
    print("rows", y, d, "per ray: nodes %.1f tris %.1f | rounds per wave-ray-phase %.1f live %.2f" % (d["nodes"] / rays, d["tris"] / rays, d["rounds"] / 64 / 240 / 2, d["live"] / max(1, d["rounds"])), flush=True)
