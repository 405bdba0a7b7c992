"""Per-pixel traversal statistics of the primary rays (instrumented mode)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer
W, H = 1920, 1080
sc, env = make_scene(abi.PROC_BISTRO_EXT, 1.0, 1, (2048, 1024))
st = host.default_state(W, H, sc, env)
r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
sc.updateCamera(W, H); sc.updateCamera(W, H); r.set_camera(sc.getCamera())
r.run(st, 0); r.sync()
r.set_counting(True); r.run_stage(st, 1, abi.STAGE_DIRECT); r.sync()
d = r.readback(abi.BUF_DENOISE_DIR_B).view(np.float32).reshape(H, W, 4)
dur, nodes, tris = d[..., 0] / 100.0, d[..., 1], d[..., 2]   # wall_clock64: 100 MHz -> us
print("per-lane us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
print("nodes: mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (nodes.mean(), *np.percentile(nodes, [50, 90, 99]), nodes.max()))
print("tris : mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (tris.mean(), *np.percentile(tris, [50, 90, 99]), tris.max()))
t = dur.reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
print("per-tile(wave) us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f sum %.0f" % (t.mean(), *np.percentile(t, [50, 90, 99]), t.max(), t.sum()))
n8 = nodes.reshape(H // 8, 8, W // 8, 8)
print("tile nodes max/mean ratio: %.2f" % (n8.max(axis=(1, 3)).mean() / nodes.mean()))
rowmean = t.mean(axis=1)
print("tile-row mean us (every 8th):", np.round(rowmean[::8], 1).tolist())
print("us per node visited (lane): %.2f" % (dur.sum() / (nodes + tris).sum()))
