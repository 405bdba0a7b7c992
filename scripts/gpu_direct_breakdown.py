"""Where the direct stage's time goes: serial stage time against RISSampleNum and ReSTIRState (timing only; the images differ)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer
W, H = 1920, 1080
KIND = sys.argv[1] if len(sys.argv) > 1 else "PROC_BISTRO_EXT"      # round 5: PROC_BISTRO_EXT_REAL (with RESTIR_SCENE_TEXSIZE=<n>: the same geometry, n x n textures)
sc, env = make_scene(getattr(abi, KIND), 1.0, 1, (2048, 1024))
print(KIND, "texsize", os.environ.get("RESTIR_SCENE_TEXSIZE", "default"), "split", os.environ.get("RESTIR_BVH_SPLIT", "default"), flush=True)
r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
sc.updateCamera(W, H)
def timed(fn, n=12):
    fn(); r.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    r.sync(); return (time.perf_counter() - t0) / n * 1e3
for mode, M in [(abi.RESTIR_TEMPORAL, 4), (abi.RESTIR_RIS, 4), (abi.RESTIR_RIS, 1), (abi.RESTIR_RIS, 2), (abi.RESTIR_RIS, 8), (abi.RESTIR_RIS, 16), (abi.RESTIR_NONE, 4)]:
    st = host.default_state(W, H, sc, env); st.ReSTIRState = mode; st.RISSampleNum = M
    for f in range(3):
        st.time = 1000 + f; sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
    r.sync()
    t = timed(lambda: r.run_stage(st, 3, abi.STAGE_DIRECT, 0, 0, H))
    print("mode", mode, "M", M, "direct ms", round(t, 3), flush=True)
st = host.default_state(W, H, sc, env); st.debugging_mode = 3
for f in range(2):
    st.time = 1000 + f; sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
print("debug view (primary ray + GetState + materials only) ms", round(timed(lambda: r.run_stage(st, 3, abi.STAGE_DIRECT, 0, 0, H)), 3))
