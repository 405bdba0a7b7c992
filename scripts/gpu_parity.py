"""GPU-vs-oracle parity sweep (run on the GPU box): python scripts/gpu_parity.py [--quick]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import abi, host, make_scene, frame_buffers, compare_buffers, RendererBackend
from restir_amd.renderer import Renderer
from oracle.binding import Oracle

def run_case(name, kind, scale, W, H, nframes, env_size=None, env_prob=0.25, max_depth=4, moving=False, restir=abi.RESTIR_TEMPORAL):
    sc, env = make_scene(kind, scale, 1, env_size)
    st = host.default_state(W, H, sc, env)
    st.environmentProb = env_prob if env is not None else 0.0
    if env is None: st.fireflyClampThreshold = 100.0
    st.maxDepth = max_depth; st.ReSTIRState = restir
    desc = sc.desc(env)
    orc = Oracle(0); orc.upload_scene(desc); orc.resize(W, H)
    r = Renderer().setup(0); t0 = time.time(); r.load_scene(desc); tb = time.time() - t0; r.update(W, H)
    gpu = RendererBackend(r)
    eye, center, up, fov = sc.cameraPose()
    res = {"case": name, "stats": sc.getStat(), "accel": r.accel_stats(), "build_s": round(tb, 2), "frames": []}
    sc.updateCamera(W, H)  # prime the history like the application's first updateFrame
    for f in range(nframes):
        st.time = 1000 + f
        if moving:
            ang = 0.01 * f
            sc.setCamera(eye + np.array([np.sin(ang), 0, np.cos(ang) - 1], dtype=np.float32) * 0.5, center, up, fov)
        sc.updateCamera(W, H)
        cam = sc.getCamera()
        orc.set_camera(cam); gpu.set_camera(cam)
        t0 = time.time(); orc.render_frame(st, f); tc = time.time() - t0
        t0 = time.time(); gpu.render_frame(st, f); r.sync(); tg = time.time() - t0
        cmp = compare_buffers(orc, gpu, frame_buffers(f))
        bad = {k: v for k, v in cmp.items() if v[0]}
        res["frames"].append({"frame": f, "cpu_s": round(tc, 3), "gpu_s": round(tg, 4), "mismatch": bad})
        print(name, "frame", f, "cpu %.3fs gpu %.4fs" % (tc, tg), "MISMATCH " + str(bad) if bad else "bit-exact", flush=True)
    return res

if __name__ == "__main__":
    quick = "--quick" in sys.argv
    out = []
    out.append(run_case("cornell-256", abi.PROC_CORNELL, 1.0, 256, 256, 3))
    out.append(run_case("helmet-128-env", abi.PROC_HELMET, 0.05, 128, 128, 3, env_size=(256, 128)))
    if not quick:
        out.append(run_case("sponza-0.02-320x180-env-moving", abi.PROC_SPONZA, 0.02, 320, 180, 4, env_size=(512, 256), moving=True))
        out.append(run_case("bistro-ext-0.01-320x180-env", abi.PROC_BISTRO_EXT, 0.01, 320, 180, 3, env_size=(512, 256)))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity.json"), "w"), indent=1, default=str)
