"""The tiled == untiled gate of the multi-GPU bench (round 5).

`bench.py --gpus N` times a row-tiled frame on a node nobody can inspect afterwards: a number for a wrong image would be recorded as a result.  After the timed
region (and outside it) the bench therefore renders a short frame sequence twice from a cold history — row-tiled over the N ranks on the partition that was
timed, and untiled on rank 0's device — and compares SHA-256 digests of EVERY buffer a frame leaves behind: G-buffer, direct / indirect reservoirs, light
ids (the "reservoir sample indices"), and both result images.  Seeds use global pixel indices, so the two are equal bit for bit or something
is broken; the JSON line carries the verdict and the bench exits non-zero on a mismatch.

The schedule it holds the tiled hosts to is Renderer::run's (src/renderer.cpp:154-206).  Backend-agnostic: the CPU test runs it over gloo with the oracle as
the backend (tests/test_tiled_gloo.py), the bench over RCCL with the HIP renderer (bench.py verify_rccl) and through the native context (verify_native).
"""
import hashlib

from . import abi
from . import tiled


def frame_buffers(cur):
    """every buffer frame parity `cur` leaves behind (the same list the parity tests compare with the oracle, tests/helpers.py frame_buffers)"""
    return [abi.BUF_GBUFFER0 + cur, abi.BUF_DIRECT_RESV0 + cur, abi.BUF_LIGHT_ID0 + cur, abi.BUF_INDIRECT_RESV0 + cur, abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_INDIRECT_RESULT0 + cur]


def digests(readback, cur):
    """{buffer name: first 16 hex digits of the SHA-256 of its bytes}; readback(buf) -> numpy uint8 array"""
    return {abi.BUFFER_NAMES[b]: hashlib.sha256(readback(b).tobytes()).hexdigest()[:16] for b in frame_buffers(cur)}


def compare(tiled_d, untiled_d):
    bufs = {k: {"tiled": tiled_d.get(k), "untiled": untiled_d.get(k), "equal": tiled_d.get(k) == untiled_d.get(k)} for k in untiled_d}
    return {"equal": all(v["equal"] for v in bufs.values()), "buffers": bufs}


def render_tiled(frame_cls, backend, comm, width, height, part, cams, state, set_camera, time0=5000):
    """len(cams) frames on a FRESH frame object (the caller has reset the backend's screen buffers: cold history) over the row partition `part`, then the
    distributed buffers of the last frame — G-buffer, reservoirs, light ids: every rank owns the authoritative copy of its band — gathered to rank 0 (the two
    result images are gathered by the frame itself).  Returns (frame, parity of the last frame)."""
    fr = frame_cls(backend, comm, width, height, part=list(part))
    for k, cam in enumerate(cams):
        state.time = time0 + k
        set_camera(cam)
        fr.render_frame(state, k)
    fr.finish()
    cur = (len(cams) - 1) & 1
    if comm.world > 1:
        for buf, pt in ((abi.BUF_GBUFFER0 + cur, fr.part), (abi.BUF_DIRECT_RESV0 + cur, fr.part), (abi.BUF_LIGHT_ID0 + cur, fr.part), (abi.BUF_INDIRECT_RESV0 + cur, fr.parth)):
            t, p = backend.tensor(buf)
            comm.gather_rows_to(t, p, pt, dst=0)
        sync = getattr(backend, "sync_all", None)
        if sync:
            sync()
        comm.barrier()
    return fr, cur


def render_untiled(run_frame, set_camera, cams, state, time0=5000):
    """the same frames through the single-device entry point (run_frame(state, frames) = Renderer::run); returns the parity of the last frame"""
    for k, cam in enumerate(cams):
        state.time = time0 + k
        set_camera(cam)
        run_frame(state, k)
    return (len(cams) - 1) & 1


def verify_cameras(scene, width, height, pose, orbit, n):
    """n cameras of the bench's camera path (static, or orbiting its centre of interest by 0.5 degrees per frame) from a freshly primed history"""
    import numpy as np
    eye0, center0, up0, fov0 = pose
    scene.setCamera(eye0, center0, up0, fov0)
    scene.updateCamera(width, height); scene.updateCamera(width, height)
    cams = []
    for k in range(n):
        if orbit:
            a = np.deg2rad(0.5 * (k + 1))
            rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=np.float32)
            scene.setCamera(center0 + rot @ (eye0 - center0), center0, up0, fov0)
        scene.updateCamera(width, height)
        cams.append(scene.getCamera())
    return cams


__all__ = ["frame_buffers", "digests", "compare", "render_tiled", "render_untiled", "verify_cameras", "tiled"]
