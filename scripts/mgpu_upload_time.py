"""Scene upload + BVH8 build time of an N-rank rt_mgpu context on one device (the host products are built once per distinct scene and shared)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer, MultiGpuRenderer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc, env = make_scene(abi.PROC_BISTRO_EXT, 1.0, 1, (2048, 1024))
desc = sc.desc(env)
t0 = time.perf_counter(); r = Renderer().setup(0); r.load_scene(desc); t1 = time.perf_counter()
print(f"single context: upload + build {t1 - t0:.2f} s")
m = MultiGpuRenderer().setup([0] * n)
t0 = time.perf_counter(); m.load_scene(desc); t1 = time.perf_counter()
print(f"rt_mgpu, {n} ranks: upload + build {t1 - t0:.2f} s")
m.destroy(); r.destroy()
