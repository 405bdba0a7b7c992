"""Same-process A/B of the two builds of the traced kernels on fixed row bands of the benchmark frame (serial schedule, stages launched alone):

    python scripts/band_ab.py [W H] y0 y1 [y0 y1 ...]

prints per band: direct / indirect stage time with RT_TRAVERSAL_THROUGHPUT and RT_TRAVERSAL_LATENCY."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer

args = [int(a) for a in sys.argv[1:]]
W, H = 1920, 1080
if len(args) >= 4 and len(args) % 2 == 0 and args[0] >= 640 and args[1] >= 360 and args[0] > args[1]:
    W, H = args[0], args[1]; args = args[2:]
bands = list(zip(args[0::2], args[1::2])) or [(0, 256), (256, 368), (368, 496), (496, 528), (528, 576), (576, 656), (656, 800), (800, 1080)]
sc, env = make_scene(abi.PROC_BISTRO_EXT, 1.0, 1, (2048, 1024))
st = host.default_state(W, H, sc, env)
r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
r.set_overlap(0)
sc.updateCamera(W, H)
for f in range(6):
    st.time = 1000 + f; sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
r.sync()


def timed(fn, n=8):
    fn(); r.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    r.sync(); return (time.perf_counter() - t0) / n * 1e3


f = 7
st.time = 1000 + f
tot = {1: 0.0, 2: 0.0}
for (y0, y1) in bands:
    line = f"rows {y0:4d}..{y1:4d}:"
    for mode, nm in ((abi.TRAVERSAL_THROUGHPUT, "thr"), (abi.TRAVERSAL_LATENCY, "lat")):
        r.set_traversal(mode)
        td = timed(lambda: r.run_stage(st, f, abi.STAGE_DIRECT, 0, y0, y1))
        ti = timed(lambda: r.run_stage(st, f, abi.STAGE_INDIRECT, 0, y0 // 2, y1 // 2))
        line += f"  {nm}: direct {td:.3f} indirect {ti:.3f} sum {td + ti:.3f}"
    r.set_traversal(abi.TRAVERSAL_AUTO)      # what the library picks for this launch (incl. the hybrid indirect stage)
    td = timed(lambda: r.run_stage(st, f, abi.STAGE_DIRECT, 0, y0, y1))
    ti = timed(lambda: r.run_stage(st, f, abi.STAGE_INDIRECT, 0, y0 // 2, y1 // 2))
    line += f"  auto: direct {td:.3f} indirect {ti:.3f} sum {td + ti:.3f}"
    print(line, flush=True)
