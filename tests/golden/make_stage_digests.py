"""Regenerates tests/golden/stage_digests.json: SHA-256 of every screen-space buffer the oracle produces for small seeded
frames ("oracle-generated per-stage dumps ... at fixed `time` seeds, frames 0..3").  The digests freeze the
oracle — and, through tests/test_golden_digests.py, the HIP path — round over round: a change in the numerics contract, the
RNG order or a stage's control flow shows up here even if it is made on both sides at once.  Run from the repo root:
    python tests/golden/make_stage_digests.py
The scenes are the host library's procedural generators with fixed seeds; their geometry digest is stored too, so that a drift
in scene generation is reported as such and not as a renderer mismatch."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402
from helpers import abi, host, make_scene, frame_buffers  # noqa: E402

# name -> (scene kind, scale, seed, env size, W, H, frames, state overrides, camera velocity per frame)
CASES = {
    "cornell_di_64": (abi.PROC_CORNELL, 1.0, 1, None, 64, 64, 4, dict(environmentProb=0.0, maxDepth=1, denoise=0), (0.0, 0.0, 0.0)),
    "cornell_full_64": (abi.PROC_CORNELL, 1.0, 1, None, 64, 64, 4, dict(environmentProb=0.0), (0.0, 0.0, 0.0)),
    "sponza_full_96x64_moving": (abi.PROC_SPONZA, 0.01, 3, (64, 32), 96, 64, 4, dict(), (0.02, 0.0, 0.01)),
    "bistro_spatiotemporal_80x48": (abi.PROC_BISTRO_EXT, 0.004, 5, (64, 32), 80, 48, 3, dict(ReSTIRState=abi.RESTIR_SPATIOTEMPORAL, RISSampleNum=8), (0.0, 0.0, 0.0)),
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()


def scene_digest(sc, env):
    import ctypes as C
    d = sc.desc(env)
    h = hashlib.sha256()
    h.update(C.string_at(d.vertices, d.numVertices * 32))
    h.update(C.string_at(d.indices, d.numIndices * 4))
    h.update(C.string_at(d.materials, d.numMaterials * 80))
    if d.envRgba32f:
        h.update(C.string_at(d.envRgba32f, d.envWidth * d.envHeight * 16))
    return h.hexdigest()


def run_case(backend_factory, case):
    """Yields (frame, {buffer name: digest}) for one case on a backend made by backend_factory(desc, W, H, sun_and_sky)."""
    kind, scale, seed, env_size, W, H, frames, overrides, vel = CASES[case]
    sc, env = make_scene(kind, scale, seed, env_size)
    st = host.default_state(W, H, sc, env)
    for k, v in overrides.items():
        setattr(st, k, v)
    if env is None:
        st.envMapLuminIntegInv = 0.0
    b = backend_factory(sc.desc(env), W, H)
    eye, center, up, fov = sc.cameraPose()
    sc.updateCamera(W, H)
    out = {"scene": scene_digest(sc, env), "frames": []}
    for f in range(frames):
        st.time = 4242 + f
        sc.setCamera(np.asarray(eye, np.float32) + np.asarray(vel, np.float32) * f, center, up, fov)
        sc.updateCamera(W, H)
        b.set_camera(sc.getCamera())
        b.render_frame(st, f)
        bufs = frame_buffers(f, indirect=st.maxDepth > 1)
        if st.ReSTIRState in (abi.RESTIR_SPATIAL, abi.RESTIR_SPATIOTEMPORAL):
            bufs = bufs + [abi.BUF_DIRECT_RESV_TEMP]
        out["frames"].append({abi.BUFFER_NAMES[x]: sha(b.readback(x)) for x in bufs})
    return out


def oracle_factory(desc, W, H):
    from oracle.binding import Oracle
    o = Oracle(0); o.upload_scene(desc); o.resize(W, H)
    return o


if __name__ == "__main__":
    res = {name: run_case(oracle_factory, name) for name in CASES}
    with open(os.path.join(HERE, "stage_digests.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("wrote tests/golden/stage_digests.json:", {k: len(v["frames"]) for k, v in res.items()})
