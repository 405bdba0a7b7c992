import os, sys, threading, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer
from restir_amd import tiled
from test_gpu_fullsize import ThreadComm
W, H = 640, 368
torch.cuda.init(); torch.zeros(1, device="cuda")
sc, env = make_scene(abi.PROC_SPONZA, 0.1, 1, (256, 128))
st = host.default_state(W, H, sc, env)
def mk():
    r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H); return r
tiled.HIST_HALO = int(os.environ.get("HH", "0"))
world = 3
eye, center, up, fov = sc.cameraPose(); sc.updateCamera(W, H)
cams = []
for f in range(4):
    sc.setCamera(eye + np.array([0.3 * f, 0.05 * f, 0], dtype=np.float32), center, up, fov); sc.updateCamera(W, H); cams.append(sc.getCamera())
for frames in (1, 2, 3):
    ref = mk(); rs = [mk() for _ in range(world)]
    shared = {"renderers": rs, "barrier": threading.Barrier(world), "slot": [None] * world, "flags": [False] * world}
    errs = []
    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            fr = tiled.PipelinedTiledFrame(tiled.RendererTensors(rs[rank]), ThreadComm(rank, world, shared), W, H)
            s = copy.copy(st)
            for f in range(frames):
                s.time = 500 + f; rs[rank].set_camera(cams[f]); fr.render_frame(s, f)
            fr.finish(); rs[rank].sync()
            print("rank", rank, "fallbacks", fr.history_fallbacks, flush=True)
        except Exception as e:
            import traceback; traceback.print_exc(); errs.append(e); shared["barrier"].abort()
    th = [threading.Thread(target=rank_main, args=(i,)) for i in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    for f in range(frames):
        st.time = 500 + f; ref.set_camera(cams[f]); ref.run(st, f)
    cur = (frames - 1) & 1
    B = tiled.band_height(H, world)
    for b, elem, half in [(abi.BUF_GBUFFER0 + cur, 16, False), (abi.BUF_DIRECT_RESV0 + cur, 36, False), (abi.BUF_INDIRECT_RESV0 + cur, 76, True),
                          (abi.BUF_DIRECT_RESULT0 + cur, 16, False), (abi.BUF_INDIRECT_RESULT0 + cur, 16, False)]:
        want = ref.readback(b)
        for rank in range(world):
            w, Bb, Hb = (W // 2, B // 2, H // 2) if half else (W, B, H)
            a, e = min(rank * Bb, Hb), min((rank + 1) * Bb, Hb)
            got = rs[rank].readback(b).reshape(-1, w * elem)[a:e]
            bad = (got != want.reshape(-1, w * elem)[a:e]).any(axis=1)
            print("frames", frames, abi.BUFFER_NAMES[b], "rank", rank, "bad rows", int(bad.sum()), "of", e - a, flush=True)
    for r in rs + [ref]: r.destroy()
