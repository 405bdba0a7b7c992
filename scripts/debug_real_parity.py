#!/usr/bin/env python
"""Round 5: localise a HIP-vs-oracle difference on the `real` footprint exterior scene (run on the GPU box).

  1. traversal alone: rt_trace_rays (closest + any hit) against the oracle's BVH2 on random / grazing / camera rays: (t, id, u, v) bits and the occlusion verdict
  2. the direct stage of frame 3 on the bands of tests/test_gpu_fullsize_allstages.py: which pixels differ, what their reservoirs hold on either side

    python scripts/debug_real_parity.py [kind] [rays]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from helpers import abi, host, make_scene  # noqa: E402
import test_trace_pin as pin  # noqa: E402


def main():
    from restir_amd.renderer import Renderer
    from oracle.binding import Oracle
    kind = getattr(abi, sys.argv[1] if len(sys.argv) > 1 else "PROC_BISTRO_EXT_REAL")
    nrays = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    scale = float(os.environ.get("DBG_SCALE", "1.0"))
    W, H = 1920, 1080
    sc, env = make_scene(kind, scale, 1, (2048, 1024))
    st = host.default_state(W, H, sc, env)
    desc = sc.desc(env)
    r = Renderer().setup(0); r.load_scene(desc); r.update(W, H)
    o = Oracle(0); o.upload_scene(desc); o.resize(W, H)
    V, UV, opaque, nocull, flip, mat, mats = pin.world_triangles(desc)
    print("triangles", len(V), "materials", len(mats), flush=True)
    rays = pin.make_rays(nrays, V, 11)
    # camera rays through the scene (primary-ray distribution) + rays from surface points towards the sky / lamps
    eye, center, up, fov = sc.cameraPose()
    rng = np.random.default_rng(5)
    n2 = nrays // 2
    d = rng.normal(size=(n2, 3)); d[:, 1] = np.abs(d[:, 1]) * 0.3; d += (center - eye) / np.linalg.norm(center - eye) * 1.5
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    cam = np.zeros((n2, 8), dtype=np.float32); cam[:, :3] = eye; cam[:, 3:6] = d; cam[:, 6] = 1e28
    cam[:, 7] = rng.integers(0, 2**31, n2).astype(np.uint32).view(np.float32)
    rays = np.concatenate([rays, cam])
    g = r.trace_closest(rays); c = o.trace_closest(rays)
    bad = np.nonzero((g.view(np.uint32) != c.view(np.uint32)).any(axis=1))[0]
    print(f"closest: {len(bad)} of {len(rays)} rays differ", flush=True)
    # second generation: from the hit points, random directions (the shadow / bounce ray distribution), finite tmax
    hit = c[:, 1].view(np.uint32) != 0xffffffff
    p = rays[hit, :3] + rays[hit, 3:6] * c[hit, 0:1]
    d2 = rng.normal(size=p.shape); d2[:, 1] = np.abs(d2[:, 1]); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    sec = np.zeros((len(p), 8), dtype=np.float32); sec[:, :3] = p - rays[hit, 3:6] * 1e-3; sec[:, 3:6] = d2; sec[:, 6] = rng.uniform(0.5, 60.0, len(p))
    sec[:, 7] = rng.integers(0, 2**31, len(p)).astype(np.uint32).view(np.float32)
    g2 = r.trace_closest(sec); c2 = o.trace_closest(sec)
    bad2 = np.nonzero((g2.view(np.uint32) != c2.view(np.uint32)).any(axis=1))[0]
    print(f"closest (second generation): {len(bad2)} of {len(sec)} rays differ", flush=True)
    ga = r.trace_any(sec); ca = o.trace_any(sec)
    bad3 = np.nonzero(ga != ca)[0]
    print(f"any hit (second generation): {len(bad3)} of {len(sec)} rays differ", flush=True)
    ga0 = r.trace_any(rays); ca0 = o.trace_any(rays)
    bad4 = np.nonzero(ga0 != ca0)[0]
    print(f"any hit (first generation): {len(bad4)} of {len(rays)} rays differ", flush=True)

    def describe(rr, gg, cc, idxs, tag):
        for i in idxs[:12]:
            gi, ci = int(gg[i, 1].view(np.uint32)), int(cc[i, 1].view(np.uint32))
            info = {"tag": tag, "ray": [float(x) for x in rr[i, :7]], "seed": int(rr[i, 7].view(np.uint32)), "gpu": [float(gg[i, 0]), gi, float(gg[i, 2]), float(gg[i, 3])],
                    "oracle": [float(cc[i, 0]), ci, float(cc[i, 2]), float(cc[i, 3])]}
            for nm, t in (("gpu_tri", gi), ("oracle_tri", ci)):
                if t != 0xffffffff and t < len(V):
                    m = mats[mat[t]]
                    info[nm] = {"mat": int(mat[t]), "opaque_inst": bool(opaque[t]), "nocull": bool(nocull[t]), "alphaMode": int(m[17]), "baseTex": int(np.int32(m[4])),
                                "verts": V[t].tolist(), "uv": UV[t].tolist(), "mat_words": [int(x) for x in m]}
            print(json.dumps(info), flush=True)
    describe(rays, g, c, bad, "closest1")
    describe(sec, g2, c2, bad2, "closest2")
    for i in bad3[:12]:
        print(json.dumps({"tag": "any2", "ray": [float(x) for x in sec[i, :7]], "seed": int(sec[i, 7].view(np.uint32)), "gpu": int(ga[i]), "oracle": int(ca[i]),
                          "closest_gpu": [float(g2[i, 0]), int(g2[i, 1].view(np.uint32))], "closest_oracle": [float(c2[i, 0]), int(c2[i, 1].view(np.uint32))]}), flush=True)

    # ---- the direct stage of the failing test ------------------------------------------------------------------------------------------------------------
    eye, center, up, fov = sc.cameraPose()
    cams = []
    sc.updateCamera(W, H)
    for f in range(4):
        sc.setCamera(eye + np.array([0.06 * f, 0.015 * f, -0.05 * f], dtype=np.float32), center, up, fov)
        sc.updateCamera(W, H); cams.append(sc.getCamera())
    for f in range(3):
        st.time = 9000 + f; r.set_camera(cams[f]); r.run(st, f)
    hist_ids = [abi.BUF_GBUFFER0, abi.BUF_DIRECT_RESV0, abi.BUF_LIGHT_ID0, abi.BUF_INDIRECT_RESV0]
    hist = {b: r.readback(b) for b in hist_ids}
    st.time = 9003; r.set_camera(cams[3]); r.run(st, 3)
    got = {b: r.readback(b) for b in (abi.BUF_GBUFFER0 + 1, abi.BUF_DIRECT_RESV0 + 1, abi.BUF_LIGHT_ID0 + 1)}
    for b, data in hist.items():
        o.upload_history(b, data)
    o.set_camera(cams[3])
    rng = np.random.default_rng(6)
    bands = sorted(int(2 * rng.integers(0, (H - 16) // 2)) for _ in range(3))
    for y0 in bands:
        o.run_stage(st, 3, abi.STAGE_DIRECT, 0, y0, y0 + 16)
        gr = got[abi.BUF_DIRECT_RESV0 + 1].view(np.uint32).reshape(H, W, 9)[y0:y0 + 16]
        orr = o.readback(abi.BUF_DIRECT_RESV0 + 1).view(np.uint32).reshape(H, W, 9)[y0:y0 + 16]
        gg = got[abi.BUF_GBUFFER0 + 1].view(np.uint32).reshape(H, W, 4)[y0:y0 + 16]
        og = o.readback(abi.BUF_GBUFFER0 + 1).view(np.uint32).reshape(H, W, 4)[y0:y0 + 16]
        gl = got[abi.BUF_LIGHT_ID0 + 1].view(np.uint32).reshape(H, W)[y0:y0 + 16]
        ol = o.readback(abi.BUF_LIGHT_ID0 + 1).view(np.uint32).reshape(H, W)[y0:y0 + 16]
        ys, xs = np.nonzero((gr != orr).any(axis=2))
        print(f"band {y0}: {len(ys)} pixels with differing reservoirs; gbuffer differing words {(gg != og).sum()}, light ids {(gl != ol).sum()}", flush=True)
        for y, x in list(zip(ys, xs))[:10]:
            print(json.dumps({"px": [int(x), int(y0 + y)], "gbuf": [hex(int(v)) for v in gg[y, x]], "hitT": float(gg[y, x, 0:1].view(np.float32)[0]),
                              "gpu_resv_f": [float(v) for v in gr[y, x].view(np.float32)], "gpu_resv_u": [int(v) for v in gr[y, x]],
                              "orc_resv_f": [float(v) for v in orr[y, x].view(np.float32)], "orc_resv_u": [int(v) for v in orr[y, x]],
                              "light": [int(gl[y, x]), int(ol[y, x])]}), flush=True)


if __name__ == "__main__":
    main()
