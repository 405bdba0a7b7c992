"""Renderer: the reference's stage dispatcher (src/renderer.hpp:49-61) over the C-ABI of csrc/librestir_hip.so.

setup / create / run / update / destroy keep the reference's names and order; the Vulkan handle parameters are gone.
There is no CPU path: if the HIP library or a GPU is missing every entry point raises.
"""
import ctypes as C
import os
import numpy as np
from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("RESTIR_HIP_LIB") or os.path.join(_HERE, "csrc", "librestir_hip.so")  # env override: A/B builds
_lib = None
PRIO_FILTER_SHARE = 0.30   # the threshold of rt_render_frame's stream-priority rule (csrc/rt_api.cpp PRIO_FILTER_SHARE; profiles/r06_prio_rule.txt) — for reports and tests

# every symbol include/rt_abi.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = ["rt_create", "rt_destroy", "rt_set_stream", "rt_upload_scene", "rt_build_accel", "rt_resize", "rt_set_camera",
               "rt_render_frame", "rt_run_stage", "rt_readback", "rt_upload_history", "rt_buffer_bytes", "rt_device_ptr",
               "rt_set_counting", "rt_get_counters", "rt_sync", "rt_last_error", "rt_abi_version", "rt_set_traversal", "rt_set_history_rows", "rt_history_miss", "rt_set_overlap", "rt_tonemap", "rt_set_sun_and_sky", "rt_pick", "rt_trace_rays", "rt_history_miss_stage", "rt_rotate_buffers", "rt_select_frame", "rt_measure_valu_peak", "rt_set_stream_priorities", "rt_get_stream_priorities", "rt_get_streams", "rt_get_stream_layout",
               "rt_mgpu_create", "rt_mgpu_destroy", "rt_mgpu_upload_scene", "rt_mgpu_resize", "rt_mgpu_set_camera", "rt_mgpu_render_frame", "rt_mgpu_readback",
               "rt_mgpu_sync", "rt_mgpu_set_balance", "rt_mgpu_set_serialize", "rt_mgpu_set_pipeline", "rt_mgpu_set_gather", "rt_mgpu_set_solo", "rt_mgpu_set_bands", "rt_mgpu_get_stats", "rt_mgpu_get_link_stats", "rt_mgpu_get_stream_layout", "rt_mgpu_last_error", "rt_mgpu_plan_bands"]


def hip_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise RuntimeError(f"{HIP_LIB_PATH} is missing — build it with __graft_entry__.build(); this package has no fallback path")
        # PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).  If torch
        # is loaded first, librestir_hip.so binds to that one runtime and streams / device pointers can be shared with
        # torch.distributed (RCCL).  Loaded the other way round the process ends up with two HIP runtimes and torch then
        # reports "No HIP GPUs are available".  So: when torch is installed, load it first.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(HIP_LIB_PATH)
        L.rt_last_error.restype = C.c_char_p
        L.rt_last_error.argtypes = [C.c_void_p]
        L.rt_buffer_bytes.restype = C.c_size_t
        L.rt_buffer_bytes.argtypes = [C.c_void_p, C.c_int]
        L.rt_abi_version.restype = C.c_uint32
        L.rt_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        for n in ["rt_destroy", "rt_build_accel", "rt_sync"]:
            getattr(L, n).argtypes = [C.c_void_p]
        L.rt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_upload_scene.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_resize.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rt_set_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_render_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rt_run_stage.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5
        L.rt_readback.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.rt_upload_history.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.rt_device_ptr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.rt_set_counting.argtypes = [C.c_void_p, C.c_int]
        L.rt_set_traversal.argtypes = [C.c_void_p, C.c_int]
        L.rt_set_overlap.argtypes = [C.c_void_p, C.c_int]
        L.rt_tonemap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.rt_set_sun_and_sky.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_history_miss_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rt_rotate_buffers.argtypes = [C.c_void_p, C.c_int]
        L.rt_select_frame.argtypes = [C.c_void_p, C.c_int]
        L.rt_trace_rays.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.rt_pick.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.rt_set_history_rows.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rt_history_miss.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.rt_get_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_measure_valu_peak.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.rt_mgpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        for n in ["rt_mgpu_destroy", "rt_mgpu_sync"]:
            getattr(L, n).argtypes = [C.c_void_p]
        L.rt_mgpu_upload_scene.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_mgpu_resize.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rt_mgpu_set_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_mgpu_render_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rt_mgpu_readback.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.rt_mgpu_set_balance.argtypes = [C.c_void_p, C.c_int]
        L.rt_mgpu_set_serialize.argtypes = [C.c_void_p, C.c_int]
        L.rt_mgpu_set_pipeline.argtypes = [C.c_void_p, C.c_int]
        L.rt_mgpu_set_gather.argtypes = [C.c_void_p, C.c_int]
        L.rt_mgpu_set_solo.argtypes = [C.c_void_p, C.c_int]
        L.rt_mgpu_set_bands.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_mgpu_get_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_mgpu_last_error.argtypes = [C.c_void_p]; L.rt_mgpu_last_error.restype = C.c_char_p
        L.rt_mgpu_plan_bands.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.rt_set_stream_priorities.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rt_get_stream_priorities.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if hasattr(L, "rt_get_streams"):   # (absent from the library round 5 shipped, which scripts/ab_libs2.sh loads through RESTIR_HIP_LIB for same-box A/Bs)
            L.rt_get_streams.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
            L.rt_get_stream_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int * 3)]
            L.rt_mgpu_get_stream_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        L.rt_accel_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.rt_accel_quality.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib = L
    return _lib


class RtError(RuntimeError):
    pass


class _DevArray:
    """Zero-copy view of a ctx-owned HBM buffer for torch.as_tensor / RCCL (via __cuda_array_interface__)."""
    def __init__(self, ptr, nbytes, owner):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}
        self._owner = owner


class Renderer:
    def __init__(self):
        self._h = None
        self.size = (0, 0)

    # Renderer::setup (renderer.cpp:62-73)
    def setup(self, device=0):
        h = C.c_void_p()
        rc = hip_lib().rt_create(C.byref(h), device)
        if rc != 0:
            raise RtError(f"rt_create failed ({rc}): {hip_lib().rt_last_error(None).decode()}")
        self._h = h
        return self

    def _chk(self, rc, what):
        if rc != 0:
            raise RtError(f"{what} failed ({rc}): {hip_lib().rt_last_error(self._h).decode()}")

    # Renderer::destroy (renderer.cpp:75-91)
    def destroy(self):
        if self._h:
            hip_lib().rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # AccelStructure::create + the upload half of Scene::load / HdrSampling::loadEnvironment
    def load_scene(self, desc):
        self._chk(hip_lib().rt_upload_scene(self._h, C.byref(desc)), "rt_upload_scene")
        self._chk(hip_lib().rt_build_accel(self._h), "rt_build_accel")

    # Renderer::create (renderer.cpp:97-148) / Renderer::update (renderer.cpp:209-225)
    def create(self, width, height, scene=None, env=None):
        if scene is not None:
            self.load_scene(scene.desc(env))
        self.update(width, height)
        return self

    def update(self, width, height):
        self._chk(hip_lib().rt_resize(self._h, width, height), "rt_resize")
        self.size = (width, height)

    def set_camera(self, cam):
        self._chk(hip_lib().rt_set_camera(self._h, C.byref(cam)), "rt_set_camera")
        self.camera = type(cam).from_buffer_copy(cam)   # the camera the next launches will use (hosts that defer work re-set it)

    def set_stream(self, stream_ptr):
        self._chk(hip_lib().rt_set_stream(self._h, stream_ptr), "rt_set_stream")

    # Renderer::run (renderer.cpp:154-206)
    def run(self, state, frames):
        self._chk(hip_lib().rt_render_frame(self._h, C.byref(state), frames), "rt_render_frame")

    def run_stage(self, state, frames, stage, level=0, row_begin=0, row_end=0):
        self._chk(hip_lib().rt_run_stage(self._h, C.byref(state), frames, stage, level, row_begin, row_end), "rt_run_stage")

    def sync(self):
        self._chk(hip_lib().rt_sync(self._h), "rt_sync")

    def buffer_bytes(self, buf):
        return hip_lib().rt_buffer_bytes(self._h, buf)

    def readback(self, buf):
        out = np.empty(self.buffer_bytes(buf), dtype=np.uint8)
        self._chk(hip_lib().rt_readback(self._h, buf, out.ctypes.data, out.nbytes), "rt_readback")
        return out

    def upload_history(self, buf, data):
        a = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        self._chk(hip_lib().rt_upload_history(self._h, buf, a.ctypes.data, a.nbytes), "rt_upload_history")

    def device_array(self, buf):
        p, n, pitch = C.c_void_p(), C.c_size_t(), C.c_size_t()
        self._chk(hip_lib().rt_device_ptr(self._h, buf, C.byref(p), C.byref(n), C.byref(pitch)), "rt_device_ptr")
        return _DevArray(p.value, n.value, self), pitch.value

    def set_traversal(self, mode):
        """abi.TRAVERSAL_AUTO (per launch, by its size) / TRAVERSAL_THROUGHPUT / TRAVERSAL_LATENCY: which build of the traced kernels runs; same bits."""
        self._chk(hip_lib().rt_set_traversal(self._h, int(mode)), "rt_set_traversal")

    def tonemap(self, tm=None, debugging_mode=0, frames=0):
        """RenderOutput::run (render_output.cpp:224-237): post.frag over the result images of `frames` -> BUF_LDR (RGBA8)."""
        tm = tm if tm is not None else abi.Tonemapper()
        self._chk(hip_lib().rt_tonemap(self._h, C.byref(tm), debugging_mode, frames), "rt_tonemap")

    def pick(self, view_inv, proj_inv, x, y):
        """SampleExample::screenPicking (sample_example.cpp:456-497): x, y = normalised window position; returns abi.PickResult."""
        out = abi.PickResult()
        self._chk(hip_lib().rt_pick(self._h, C.byref(view_inv), C.byref(proj_inv), x, y, C.byref(out)), "rt_pick")
        return out

    def trace_closest(self, rays):
        """rays: (n, 8) float32 (origin, direction, tmax, seed bits) -> (n, 4) float32: t, triangle index bits, u, v  (the oracle's trace_closest format)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        out = np.zeros((rays.shape[0], 4), dtype=np.float32)
        self._chk(hip_lib().rt_trace_rays(self._h, rays.shape[0], rays.ctypes.data, out.ctypes.data, 0), "rt_trace_rays")
        return out

    def trace_any(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        out = np.zeros((rays.shape[0], 4), dtype=np.float32)
        self._chk(hip_lib().rt_trace_rays(self._h, rays.shape[0], rays.ctypes.data, out.ctypes.data, 1), "rt_trace_rays")
        return (out[:, 0] > 0.5).astype(np.int32)

    def set_sun_and_sky(self, ss):
        """updateUniformBuffer's SunAndSky upload (sample_example.cpp:172); ss.in_use = 1 switches the environment to the procedural sky."""
        self._chk(hip_lib().rt_set_sun_and_sky(self._h, C.byref(ss)), "rt_set_sun_and_sky")

    def select_frame(self, frames):
        """RT_BUF_DENOISE_IND_A exists once per frame parity: name the one of `frames` for the next device_array / readback."""
        self._chk(hip_lib().rt_select_frame(self._h, int(frames)), "rt_select_frame")

    def set_overlap(self, mode):
        """0 = serial launches, 1 = direct A-Trous beside the indirect stage, 2 = 1 + frames in flight (default), 3 = 2 with a third frame in flight."""
        self._chk(hip_lib().rt_set_overlap(self._h, int(mode)), "rt_set_overlap")

    def history_miss_stage(self, stage):
        m = C.c_int()
        self._chk(hip_lib().rt_history_miss_stage(self._h, stage, C.byref(m)), "rt_history_miss_stage")
        return bool(m.value)

    def rotate_buffers(self, frames):
        self._chk(hip_lib().rt_rotate_buffers(self._h, frames), "rt_rotate_buffers")

    def set_history_rows(self, row0, row1):
        self._chk(hip_lib().rt_set_history_rows(self._h, row0, row1), "rt_set_history_rows")

    def history_miss(self):
        m = C.c_int()
        self._chk(hip_lib().rt_history_miss(self._h, C.byref(m)), "rt_history_miss")
        return bool(m.value)

    def set_counting(self, enable):
        self._chk(hip_lib().rt_set_counting(self._h, 1 if enable else 0), "rt_set_counting")

    def counters(self):
        c = abi.Counters()
        self._chk(hip_lib().rt_get_counters(self._h, C.byref(c)), "rt_get_counters")
        return c

    def measure_valu_peak(self, variant=0, waves_per_simd=8):
        """wave-level VALU instructions per second of a chain-free loop on this device (csrc/microbench.hip)"""
        v = C.c_double()
        self._chk(hip_lib().rt_measure_valu_peak(self._h, variant, waves_per_simd, C.byref(v)), "rt_measure_valu_peak")
        return v.value

    def set_stream_priorities(self, indirect_level, filter_level):
        """levels -1 / 0 / +1 of the indirect-stage stream and the filter stream of the frames-in-flight schedule (same bits under every setting)"""
        self._chk(hip_lib().rt_set_stream_priorities(self._h, int(indirect_level), int(filter_level)), "rt_set_stream_priorities")

    def stream_priorities(self):
        """{"chosen": [indirect, filter], "filter_share": filters / (direct + indirect) of the last probe frame or None, "decided": bool} (rt_get_stream_priorities)"""
        a, b, sh, d = C.c_int(), C.c_int(), C.c_float(), C.c_int()
        self._chk(hip_lib().rt_get_stream_priorities(self._h, C.byref(a), C.byref(b), C.byref(sh), C.byref(d)), "rt_get_stream_priorities")
        return {"chosen": [a.value, b.value], "filter_share": (round(sh.value, 4) if sh.value >= 0 else None), "decided": bool(d.value)}

    def streams(self):
        """{"main", "ind", "side"}: the context's three hipStream_t handles of the frames-in-flight schedule (rt_get_streams; created on first call — filter stream, then
        indirect stream — with the current levels).  A host that issues the stages itself runs on these (restir_amd/tiled.py) instead of on streams of its own."""
        m, i, f = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(hip_lib().rt_get_streams(self._h, C.byref(m), C.byref(i), C.byref(f)), "rt_get_streams")
        return {"main": m.value, "ind": i.value, "side": f.value}

    def stream_layout(self):
        """{"library_streams_created": n, "creation_index": {"main": i, "ind": j, "side": k}} (rt_get_stream_layout; -1 = not created)"""
        n, idx = C.c_int(), (C.c_int * 3)()
        self._chk(hip_lib().rt_get_stream_layout(self._h, C.byref(n), C.byref(idx)), "rt_get_stream_layout")
        return {"library_streams_created": n.value, "creation_index": {"main": idx[0], "ind": idx[1], "side": idx[2]}}

    def accel_stats(self):
        n, t, d = C.c_uint64(), C.c_uint64(), C.c_int()
        self._chk(hip_lib().rt_accel_stats(self._h, C.byref(n), C.byref(t), C.byref(d)), "rt_accel_stats")
        refs, sp, sn, stt = C.c_uint64(), C.c_uint64(), C.c_double(), C.c_double()
        self._chk(hip_lib().rt_accel_quality(self._h, C.byref(refs), C.byref(sp), C.byref(sn), C.byref(stt)), "rt_accel_quality")
        return {"nodes": n.value, "triangles": t.value, "max_depth": d.value, "references": refs.value, "spatial_splits": sp.value,
                "sah_node_steps": round(sn.value, 3), "sah_tri_steps": round(stt.value, 3)}


class MgpuStats(C.Structure):  # rt_mgpu_stats
    _fields_ = [("numRanks", C.c_int32), ("frames", C.c_uint32), ("historyFallbacks", C.c_uint32), ("pad", C.c_uint32), ("haloBytes", C.c_uint64),
                ("bandBegin", C.c_int32 * 16), ("bandEnd", C.c_int32 * 16), ("tracedMs", C.c_float * 16), ("filterMs", C.c_float * 16), ("haloBytesKind", C.c_uint64 * 6), ("haloBytesRankKind", (C.c_uint64 * 6) * 16)]


class MgpuLinkStats(C.Structure):  # rt_mgpu_link_stats (ABI 2.1)
    _fields_ = [("numRanks", C.c_int32), ("devices", C.c_int32 * 16), ("peerAccess", (C.c_uint8 * 16) * 16), ("pullMs", (C.c_float * 4) * 16), ("pullBytes", (C.c_uint64 * 4) * 16)]


LINK_GROUPS = ("history", "history_indirect", "filter_direct", "filter_indirect")


class MultiGpuRenderer:
    """rt_mgpu_*: one process drives N devices (csrc/mgpu.cpp) — the Renderer interface over the native row-tiled frame."""
    def __init__(self):
        self._h = None
        self.size = (0, 0)

    def setup(self, devices):
        devices = list(devices)
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = hip_lib().rt_mgpu_create(C.byref(h), len(devices), arr)
        if rc != 0:
            raise RtError(f"rt_mgpu_create failed ({rc}): {hip_lib().rt_last_error(None).decode()}")
        self._h, self.world = h, len(devices)
        return self

    def _chk(self, rc, what):
        if rc != 0:
            raise RtError(f"{what} failed ({rc}): {hip_lib().rt_mgpu_last_error(self._h).decode()}")

    def destroy(self):
        if self._h:
            hip_lib().rt_mgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def load_scene(self, desc): self._chk(hip_lib().rt_mgpu_upload_scene(self._h, C.byref(desc)), "rt_mgpu_upload_scene")
    def update(self, width, height):
        self._chk(hip_lib().rt_mgpu_resize(self._h, width, height), "rt_mgpu_resize")
        self.size = (width, height)
    def set_camera(self, cam): self._chk(hip_lib().rt_mgpu_set_camera(self._h, C.byref(cam)), "rt_mgpu_set_camera")
    def run(self, state, frames): self._chk(hip_lib().rt_mgpu_render_frame(self._h, C.byref(state), frames), "rt_mgpu_render_frame")
    def sync(self): self._chk(hip_lib().rt_mgpu_sync(self._h), "rt_mgpu_sync")
    def set_balance(self, on): self._chk(hip_lib().rt_mgpu_set_balance(self._h, int(on)), "rt_mgpu_set_balance")   # True / 1: cost weighted, False / 0: equal, 2: freeze
    def set_serialize(self, on): self._chk(hip_lib().rt_mgpu_set_serialize(self._h, 1 if on else 0), "rt_mgpu_set_serialize")
    def set_pipeline(self, on): self._chk(hip_lib().rt_mgpu_set_pipeline(self._h, 1 if on else 0), "rt_mgpu_set_pipeline")
    def set_gather(self, on): self._chk(hip_lib().rt_mgpu_set_gather(self._h, 1 if on else 0), "rt_mgpu_set_gather")
    def set_solo(self, rank): self._chk(hip_lib().rt_mgpu_set_solo(self._h, int(rank)), "rt_mgpu_set_solo")
    def set_bands(self, bands):
        arr = (C.c_int * len(bands))(*[int(b) for b in bands])
        self._chk(hip_lib().rt_mgpu_set_bands(self._h, arr), "rt_mgpu_set_bands")
    @staticmethod
    def plan_bands(height, ranks, stripe_cost, prev=None, max_move=-1):
        cost = (C.c_float * len(stripe_cost))(*[float(x) for x in stripe_cost])
        pb = (C.c_int * (ranks + 1))(*prev) if prev is not None else None
        out = (C.c_int * (ranks + 1))()
        rc = hip_lib().rt_mgpu_plan_bands(height, ranks, cost, pb, max_move, out)
        if rc != 0:
            raise RtError(f"rt_mgpu_plan_bands failed ({rc})")
        return list(out)
    def stats(self):
        s = MgpuStats()
        self._chk(hip_lib().rt_mgpu_get_stats(self._h, C.byref(s)), "rt_mgpu_get_stats")
        return s
    def stream_layout(self):
        """per rank: creation index of its main / indirect / filter stream in the process (-1 = not created yet), + the levels (rt_mgpu_get_stream_layout)"""
        n = self.world
        created, idx, lv = C.c_int(), (C.c_int * (3 * n))(), (C.c_int * 3)()
        self._chk(hip_lib().rt_mgpu_get_stream_layout(self._h, C.byref(created), idx, lv), "rt_mgpu_get_stream_layout")
        return {"library_streams_created": created.value, "levels_main_ind_side": [lv[0], lv[1], lv[2]],
                "creation_index": [{"main": idx[3 * r], "ind": idx[3 * r + 1], "side": idx[3 * r + 2]} for r in range(n)]}
    def link_stats(self):
        s = MgpuLinkStats()
        self._chk(hip_lib().rt_mgpu_get_link_stats(self._h, C.byref(s)), "rt_mgpu_get_link_stats")
        return s
    def readback(self, buf):
        W, H = self.size
        half = buf in (abi.BUF_INDIRECT_RESV0, abi.BUF_INDIRECT_RESV0 + 1, abi.BUF_INDIRECT_RESV0 + 2)
        elem = {abi.BUF_MOTION: 4, abi.BUF_LIGHT_ID0: 4, abi.BUF_LIGHT_ID0 + 1: 4, abi.BUF_LDR: 4, abi.BUF_DIRECT_RESV0: 36, abi.BUF_DIRECT_RESV0 + 1: 36,
                abi.BUF_DIRECT_RESV0 + 2: 36, abi.BUF_INDIRECT_RESV0: 76, abi.BUF_INDIRECT_RESV0 + 1: 76, abi.BUF_INDIRECT_RESV0 + 2: 76}.get(buf, 16)
        n = ((W // 2) * (H // 2) if half else W * H) * elem
        out = np.empty(n, dtype=np.uint8)
        self._chk(hip_lib().rt_mgpu_readback(self._h, buf, out.ctypes.data, out.nbytes), "rt_mgpu_readback")
        return out
