"""Tree-quality A/B on the bench scene: node / triangle steps per ray and stage times for the current builder settings (env)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer
W, H = 1920, 1080
KIND = sys.argv[1] if len(sys.argv) > 1 else "PROC_BISTRO_EXT"      # e.g. PROC_BISTRO_EXT_REAL (round 5: the `real` footprint scene)
sc, env = make_scene(getattr(abi, KIND), 1.0, 1, (2048, 1024))
st = host.default_state(W, H, sc, env)
r = Renderer().setup(0); t0 = time.time(); r.load_scene(sc.desc(env)); tb = time.time() - t0
r.update(W, H); r.set_overlap(0)
sc.updateCamera(W, H)
for f in range(4):
    st.time = 1000 + f; sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
out = [KIND, "split=%s alpha=%s collapse=%s" % (os.environ.get("RESTIR_BVH_SPLIT", "default"), os.environ.get("RESTIR_BVH_SPLIT_ALPHA", "default"), os.environ.get("RESTIR_BVH_COLLAPSE", "greedy")),
       str(r.accel_stats()), "upload + build %.2fs" % tb]
for stage, name in ((abi.STAGE_DIRECT, "direct"), (abi.STAGE_INDIRECT, "indirect")):
    r.sync(); t0 = time.perf_counter()
    for _ in range(8): r.run_stage(st, 4, stage)
    r.sync(); ms = (time.perf_counter() - t0) / 8 * 1e3
    r.set_counting(True); r.run_stage(st, 4, stage); c = r.counters(); r.set_counting(False)
    rays = c.closestHitRays + c.anyHitRays
    out.append("%s: nodes/ray %.2f tris/ray %.2f  %.3f ms" % (name, c.nodesVisited / rays, c.trisTested / rays, ms))
print(" | ".join(out), flush=True)
