"""SIMD efficiency of the traversal loops per stage (counting mode): how much of the wave-rounds is useful."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer
W, H = 1920, 1080
for name, kind, md in (("sponza", abi.PROC_SPONZA, 2), ("bistro", abi.PROC_BISTRO_EXT, 4)):
    sc, env = make_scene(kind, 1.0, 1, (2048, 1024))
    st = host.default_state(W, H, sc, env); st.maxDepth = md
    r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
    sc.updateCamera(W, H)
    for f in range(4):
        st.time = 1000 + f; sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
    for stage, sname in ((abi.STAGE_DIRECT, "direct"), (abi.STAGE_INDIRECT, "indirect")):
        r.sync(); import time as _t; t0 = _t.perf_counter()
        for _ in range(5): r.run_stage(st, 4, stage)
        r.sync(); ms = (_t.perf_counter() - t0) / 5 * 1e3
        r.set_counting(True); r.run_stage(st, 4, stage); c = r.counters(); r.set_counting(False)
        rays = c.closestHitRays + c.anyHitRays
        steps = c.nodesVisited + c.trisTested
        print(name, sname, "rays %d steps/ray %.1f (nodes %.1f tris %.1f) | vote eff %.2f | not-waiting %.2f | overall %.2f | rounds/ray-lane %.1f | %.3f ms" % (
            rays, steps / rays, c.nodesVisited / rays, c.trisTested / rays, steps / max(1, c.laneLiveRounds), c.laneLiveRounds / max(1, c.laneRounds),
            steps / max(1, c.laneRounds), c.laneRounds / 64.0 / max(1, rays) * 64, ms), flush=True)
    r.destroy()
