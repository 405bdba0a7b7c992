"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
the product package never does (the product fails loudly when its HIP library is missing instead of falling
back to anything in here).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RESTIR_ORACLE_LIB") or os.path.join(_HERE, "_build", "liboracle.so")   # env override: the sanitizer build
_lib = None

def build(force=False):
    # make decides whether the library is stale (a no-op when it is up to date); without a compiler the prebuilt file is used
    import shutil
    try:
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        # a machine without a compiler uses the prebuilt file; with one, a failed build must not fall back to a stale library silently
        if not os.path.exists(LIB_PATH) or (shutil.which("g++") and shutil.which("make")):
            raise
    return LIB_PATH

def lib():
    global _lib
    if _lib is None:
        if not os.environ.get("RESTIR_ORACLE_LIB"):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int]
        L.orc_buffer_bytes.restype = C.c_size_t
        L.orc_num_triangles.restype = C.c_uint64
        L.orc_rand.restype = C.c_float
        L.orc_tea.restype = C.c_uint32; L.orc_pcg.restype = C.c_uint32; L.orc_hash8bit.restype = C.c_uint32
        L.orc_compress_unit_vec.restype = C.c_uint32; L.orc_pack_unorm4x8.restype = C.c_uint32
        L.orc_tea.argtypes = [C.c_uint32, C.c_uint32]; L.orc_hash8bit.argtypes = [C.c_uint32]
        L.orc_pcg.argtypes = [C.c_void_p]; L.orc_rand.argtypes = [C.c_void_p]
        L.orc_compress_unit_vec.argtypes = [C.c_float] * 3; L.orc_pack_unorm4x8.argtypes = [C.c_float] * 4
        L.orc_decompress_unit_vec.argtypes = [C.c_uint32, C.c_void_p]
        for n in ["orc_destroy", "orc_reset_counters", "orc_num_triangles", "orc_threads"]:
            getattr(L, n).argtypes = [C.c_void_p]
        L.orc_set_threads.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_upload_scene.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_resize.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_set_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_render_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_run_stage.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5
        L.orc_buffer_bytes.argtypes = [C.c_void_p, C.c_int]
        L.orc_readback.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_upload_history.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_get_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_history_rows.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_tonemap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_set_sun_and_sky.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pick.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_sun_and_sky_eval.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_history_miss.argtypes = [C.c_void_p]
        L.orc_history_miss_stage.argtypes = [C.c_void_p, C.c_int]
        L.orc_buffer_ptr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        for n in ["orc_trace_closest", "orc_trace_any", "orc_trace_closest_brute"]:
            getattr(L, n).argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_offset_ray.argtypes = [C.c_void_p] * 3
        L.orc_concentric_disk.argtypes = [C.c_float, C.c_float, C.c_void_p]
        L.orc_bsdf_eval.argtypes = [C.c_void_p] * 5
        L.orc_power_heuristic.argtypes = [C.c_float, C.c_float]; L.orc_power_heuristic.restype = C.c_float
        L.orc_luminance.argtypes = [C.c_void_p]; L.orc_luminance.restype = C.c_float
        L.orc_hdr_to_ldr.argtypes = [C.c_void_p] * 2
        L.orc_ldr_to_hdr.argtypes = [C.c_void_p] * 2
        L.orc_resv_op.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_bsdf_sample.argtypes = [C.c_void_p] * 5
        L.orc_spherical_uv.argtypes = [C.c_void_p] * 2
        L.orc_coordinate_system.argtypes = [C.c_void_p] * 2
        L.orc_post_fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_pcg3d.argtypes = [C.c_void_p]
        L.orc_detmath.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class Oracle:
    """Same call surface as the product's Renderer (create/upload/resize/set_camera/render_frame/readback)."""
    def __init__(self, threads=0):
        self._h = lib().orc_create(threads)
    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h); self._h = None
    @property
    def threads(self): return lib().orc_threads(self._h)
    def set_threads(self, threads, pin=False): lib().orc_set_threads(self._h, int(threads), 1 if pin else 0)
    def _chk(self, rc, what):
        if rc != 0: raise RuntimeError(f"oracle {what} failed: {rc}")
    def upload_scene(self, desc): self._chk(lib().orc_upload_scene(self._h, C.byref(desc)), "upload_scene")
    def resize(self, w, h): self._chk(lib().orc_resize(self._h, w, h), "resize")
    def set_camera(self, cam):
        self._chk(lib().orc_set_camera(self._h, C.byref(cam)), "set_camera")
        self.camera = type(cam).from_buffer_copy(cam)
    def render_frame(self, state, frames): self._chk(lib().orc_render_frame(self._h, C.byref(state), frames), "render_frame")
    def run_stage(self, state, frames, stage, level=0, row_begin=0, row_end=0):
        self._chk(lib().orc_run_stage(self._h, C.byref(state), frames, stage, level, row_begin, row_end), "run_stage")
    def set_sun_and_sky(self, ss): self._chk(lib().orc_set_sun_and_sky(self._h, C.byref(ss)), "set_sun_and_sky")
    def pick(self, view_inv, proj_inv, x, y):
        from restir_amd import abi
        out = abi.PickResult()
        self._chk(lib().orc_pick(self._h, C.byref(view_inv), C.byref(proj_inv), x, y, C.byref(out)), "pick")
        return out
    def tonemap(self, tm, debugging_mode=0, frames=0):
        self._chk(lib().orc_tonemap(self._h, C.byref(tm), debugging_mode, frames), "tonemap")
    def buffer_bytes(self, buf): return lib().orc_buffer_bytes(self._h, buf)
    def readback(self, buf):
        out = np.empty(self.buffer_bytes(buf), dtype=np.uint8)
        self._chk(lib().orc_readback(self._h, buf, out.ctypes.data, out.nbytes), "readback")
        return out
    def upload_history(self, buf, data):
        a = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        self._chk(lib().orc_upload_history(self._h, buf, a.ctypes.data, a.nbytes), "upload_history")
    def set_history_rows(self, r0, r1): lib().orc_set_history_rows(self._h, r0, r1)
    def history_miss(self): return bool(lib().orc_history_miss(self._h))
    def history_miss_stage(self, stage): return bool(lib().orc_history_miss_stage(self._h, stage))
    def buffer_array(self, buf):
        """(flat uint8 numpy view of the whole allocation incl. slack rows, row pitch in bytes) — zero copy"""
        p, n, pitch = C.c_void_p(), C.c_size_t(), C.c_size_t()
        self._chk(lib().orc_buffer_ptr(self._h, buf, C.byref(p), C.byref(n), C.byref(pitch)), "buffer_ptr")
        arr = np.ctypeslib.as_array((C.c_uint8 * n.value).from_address(p.value))
        return arr, pitch.value
    def counters(self):
        from importlib import import_module
        import sys
        abi = sys.modules.get("restir_amd.abi") or import_module("restir_amd.abi")
        c = abi.Counters()
        lib().orc_get_counters(self._h, C.byref(c))
        return c
    def reset_counters(self): lib().orc_reset_counters(self._h)
    def num_triangles(self): return lib().orc_num_triangles(self._h)
    def trace_closest(self, rays, brute=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.empty((rays.shape[0], 4), dtype=np.float32)
        (lib().orc_trace_closest_brute if brute else lib().orc_trace_closest)(self._h, rays.shape[0], rays.ctypes.data, out.ctypes.data)
        return out
    def trace_any(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.empty(rays.shape[0], dtype=np.int32)
        lib().orc_trace_any(self._h, rays.shape[0], rays.ctypes.data, out.ctypes.data)
        return out


def sun_and_sky_eval(ss, dirs):
    """sun_and_sky(ss, dir) (sun_and_sky.glsl:453-601) for an (n,3) float32 array of directions."""
    d = np.ascontiguousarray(dirs, dtype=np.float32)
    out = np.empty_like(d)
    lib().orc_sun_and_sky_eval(C.byref(ss), len(d), d.ctypes.data, out.ctypes.data)
    return out
