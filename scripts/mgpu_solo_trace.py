"""One emulated rank of the 8-way frame with frames in flight, alone on the GPU (rt_mgpu_set_solo), for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d out -o solo -- python scripts/mgpu_solo_trace.py <rank> [frames]
prints the wall-clock period; scripts/trace_overlap.py reads the trace."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
from helpers import abi, host, make_scene
from restir_amd.renderer import MultiGpuRenderer
rank = int(sys.argv[1]); frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, H, n = 1920, 1080, 8
bands = [int(x) for x in os.environ.get("BANDS", "0,208,352,448,512,576,656,832,1080").split(",")]
sc, env = make_scene(abi.PROC_BISTRO_EXT, 1.0, 1, (2048, 1024))
st = host.default_state(W, H, sc, env)
m = MultiGpuRenderer().setup([0] * n); m.load_scene(sc.desc(env)); m.update(W, H)
sc.updateCamera(W, H)
f = 0
def frame():
    global f
    st.time = 1000 + f; sc.updateCamera(W, H); m.set_camera(sc.getCamera()); m.run(st, f); f += 1
m.set_serialize(True)
for _ in range(4): frame()
m.sync()
m.set_bands(bands); m.set_serialize(False); m.set_pipeline(True)
for _ in range(3): frame()
m.set_solo(rank)
for _ in range(6): frame()
m.sync(); t0 = time.perf_counter()
for _ in range(frames): frame()
m.sync()
print("rank", rank, "rows", bands[rank], bands[rank + 1], "period ms", (time.perf_counter() - t0) / frames * 1e3)
