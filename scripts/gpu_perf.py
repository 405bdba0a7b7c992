"""Per-stage timing + traversal counters at benchmark sizes (run on the GPU box)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import abi, host, make_scene
from restir_amd.renderer import Renderer

STAGES = ["direct", "indirect", "denoise_direct", "denoise_indirect", "compose", "direct_gen", "direct_reuse"]

def perf_case(name, kind, scale, W, H, max_depth=4, frames=12, warm=4):
    t0 = time.time(); sc, env = make_scene(kind, scale, 1, (2048, 1024)); tgen = time.time() - t0
    st = host.default_state(W, H, sc, env); st.maxDepth = max_depth
    r = Renderer().setup(0)
    t0 = time.time(); r.load_scene(sc.desc(env)); tbuild = time.time() - t0
    r.update(W, H)
    sc.updateCamera(W, H)
    def frame(f):
        st.time = 1000 + f; sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
    for f in range(warm): frame(f)
    r.set_counting(False)
    r.sync(); t0 = time.time()
    for f in range(warm, warm + frames): frame(f)
    r.sync(); wall = (time.time() - t0) / frames * 1e3
    c = r.counters()
    ms = {STAGES[i]: c.stageMs[i] / max(1, c.framesTimed) for i in range(5)}
    r.set_counting(True)
    for f in range(warm + frames, warm + frames + 2): frame(f)
    k = r.counters(); r.set_counting(False)
    rays = (k.closestHitRays + k.anyHitRays) / 2
    d = r.readback(abi.BUF_DIRECT_RESULT0 + ((warm + frames + 1) & 1)).view(np.float32).reshape(H, W, 4)
    i = r.readback(abi.BUF_INDIRECT_RESULT0 + ((warm + frames + 1) & 1)).view(np.float32).reshape(H, W, 4)
    out = {"case": name, "tris": sc.getStat()["instancedTriangles"], "accel": r.accel_stats(), "gen_s": round(tgen, 2), "build_s": round(tbuild, 2),
           "wall_ms": round(wall, 3), "event_ms": round(c.frameMs / max(1, c.framesTimed), 3), "stage_ms": {k_: round(v, 3) for k_, v in ms.items()},
           "rays_per_frame": rays, "Mrays_s": round(rays / wall / 1e3, 1), "nodes_per_ray": round(k.nodesVisited / 2 / rays, 2), "tris_per_ray": round(k.trisTested / 2 / rays, 2),
           "closest": k.closestHitRays // 2, "any": k.anyHitRays // 2, "direct_mean": float(d[..., :3].mean()), "indirect_mean": float(i[..., :3].mean()),
           "nan": int(np.isnan(d).sum() + np.isnan(i).sum())}
    print(json.dumps(out), flush=True)
    try:
        from PIL import Image
        img = d[..., :3] + i[..., :3]; tm = np.clip(img / (1 + img), 0, 1) ** (1 / 2.2)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        Image.fromarray((tm * 255).astype(np.uint8)).resize((W // 2, H // 2)).save(os.path.join(ROOT, "gpurun_out", name + ".jpg"), quality=85)
    except Exception as e:
        print("no image:", e)
    r.destroy()
    return out

if __name__ == "__main__":
    res = []
    which = sys.argv[1:] or ["cornell", "sponza", "bistro"]
    if "cornell" in which: res.append(perf_case("cornell-512", abi.PROC_CORNELL, 1.0, 512, 512))
    if "sponza" in which: res.append(perf_case("sponza-1080p", abi.PROC_SPONZA, 1.0, 1920, 1080, max_depth=2))
    if "bistro" in which: res.append(perf_case("bistro-ext-1080p", abi.PROC_BISTRO_EXT, 1.0, 1920, 1080, max_depth=4))
    if "interior" in which: res.append(perf_case("bistro-int-4k", abi.PROC_BISTRO_INT, 1.0, 3840, 2160, max_depth=4, frames=6))
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "perf.json"), "w"), indent=1)
