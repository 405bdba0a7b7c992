"""Shared helpers for the parity tests: build a scene, drive the oracle and the HIP renderer through the same
frames, compare every screen-space buffer bit for bit."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import restir_amd  # noqa: E402,F401  (import shim for the hyphenated package directory)
from restir_amd import abi, host  # noqa: E402


def make_scene(kind, scale=1.0, seed=1, env_size=None, sun_peak=5e4):
    sc = host.Scene().makeProcedural(kind, scale, seed)
    env = None
    if env_size:
        env = host.HdrSampling()
        env.makeSyntheticSky(env_size[0], env_size[1], sun_peak, 7)
    return sc, env


def frame_buffers(frames, indirect=True):
    cur = frames & 1
    bufs = [abi.BUF_GBUFFER0 + cur, abi.BUF_MOTION, abi.BUF_DIRECT_RESV0 + cur, abi.BUF_LIGHT_ID0 + cur, abi.BUF_DIRECT_RESULT0 + cur]
    if indirect:
        bufs += [abi.BUF_INDIRECT_RESV0 + cur, abi.BUF_INDIRECT_RESULT0 + cur, abi.BUF_DENOISE_DIR_A, abi.BUF_DENOISE_DIR_B,
                 abi.BUF_DENOISE_IND_A, abi.BUF_DENOISE_IND_B]
    return bufs


def compare_buffers(a, b, buffers):
    """Returns {name: (mismatching 32-bit words, total words)} comparing raw bytes."""
    out = {}
    for buf in buffers:
        x = a.readback(buf).view(np.uint32)
        y = b.readback(buf).view(np.uint32)
        assert x.shape == y.shape, abi.BUFFER_NAMES[buf]
        out[abi.BUFFER_NAMES[buf]] = (int((x != y).sum()), int(x.size))
    return out


def run_frames(backend, scene, state, nframes, width, height, camera_path=None, stages=None, time0=1000):
    """Drive `backend` (Oracle or Renderer adapter exposing set_camera/render_frame|run) for nframes.
    The scene's camera history is advanced exactly like SampleExample::updateFrame does."""
    for f in range(nframes):
        state.time = time0 + f
        if camera_path is not None:
            eye, center = camera_path(f)
            scene.setCamera(eye, center, (0, 1, 0), scene.cameraPose()[3])
        scene.updateCamera(width, height)
        backend.set_camera(scene.getCamera())
        if stages is None:
            backend.render_frame(state, f)
        else:
            for (stage, level) in stages:
                backend.run_stage(state, f, stage, level)


class RendererBackend:
    """Adapter giving the HIP Renderer the Oracle's method names."""
    def __init__(self, renderer):
        self.r = renderer
    def set_camera(self, cam): self.r.set_camera(cam)
    def render_frame(self, state, frames): self.r.run(state, frames)
    def run_stage(self, state, frames, stage, level=0, row_begin=0, row_end=0): self.r.run_stage(state, frames, stage, level, row_begin, row_end)
    def readback(self, buf): return self.r.readback(buf)
