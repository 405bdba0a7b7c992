"""Render serial frames with spatiotemporal reuse (for rocprofv3: scripts/spatial_lds_ab.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer

W, H = 1920, 1080
sc, env = make_scene(abi.PROC_SPONZA, 1.0, 1, (1024, 512))
st = host.default_state(W, H, sc, env); st.ReSTIRState = abi.RESTIR_SPATIOTEMPORAL; st.maxDepth = 2
r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
sc.updateCamera(W, H)
for f in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    st.time = 1000 + f; sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
r.sync(); r.destroy()
