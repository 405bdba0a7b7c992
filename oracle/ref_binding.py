"""ctypes binding of oracle/_ref/libref_stages.so — the REFERENCE's own stage shaders compiled for the CPU from where they lie
under /root/reference (oracle/kat/build_ref_stages.sh; authoring container only).

TEST INFRASTRUCTURE.  Same call surface as oracle.binding.Oracle, so a test can drive both with the same code.  Only tests/ and
tests/golden/make_ref_stage_vectors.py import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref_stages.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def build():
    """(re)build from /root/reference when it is present; returns True when the library exists afterwards"""
    if os.path.isdir("/root/reference/shaders"):
        subprocess.check_call(["bash", os.path.join(_HERE, "kat", "build_ref_stages.sh")], stdout=subprocess.DEVNULL)
    return available()


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.ref_create.restype = C.c_void_p
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_upload_scene.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_set_sun_and_sky.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_resize.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_set_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_render_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ref_run_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ref_buffer_bytes.argtypes = [C.c_void_p, C.c_int]; L.ref_buffer_bytes.restype = C.c_size_t
        L.ref_readback.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.ref_upload_history.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


class Reference:
    def __init__(self):
        self._h = lib().ref_create()
    def __del__(self):
        if getattr(self, "_h", None):
            lib().ref_destroy(self._h); self._h = None
    def _chk(self, rc, what):
        if rc != 0: raise RuntimeError(f"reference {what} failed: {rc}")
    def upload_scene(self, desc): self._chk(lib().ref_upload_scene(self._h, C.byref(desc)), "upload_scene")
    def set_sun_and_sky(self, ss): self._chk(lib().ref_set_sun_and_sky(self._h, C.byref(ss)), "set_sun_and_sky")
    def resize(self, w, h): self._chk(lib().ref_resize(self._h, w, h), "resize")
    def set_camera(self, cam): self._chk(lib().ref_set_camera(self._h, C.byref(cam)), "set_camera")
    def render_frame(self, state, frames): self._chk(lib().ref_render_frame(self._h, C.byref(state), frames), "render_frame")
    def run_stage(self, state, frames, stage, level=0): self._chk(lib().ref_run_stage(self._h, C.byref(state), frames, stage, level), "run_stage")
    def buffer_bytes(self, buf): return lib().ref_buffer_bytes(self._h, buf)
    def readback(self, buf):
        out = np.empty(self.buffer_bytes(buf), dtype=np.uint8)
        self._chk(lib().ref_readback(self._h, buf, out.ctypes.data, out.nbytes), "readback")
        return out
    def upload_history(self, buf, data):
        a = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        self._chk(lib().ref_upload_history(self._h, buf, a.ctypes.data, a.nbytes), "upload_history")
