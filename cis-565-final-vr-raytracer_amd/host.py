"""Python face of the host side (host/scene.hpp, host/hdr_sampling.hpp): scene + environment preparation.

Mirrors the reference call order SampleExample::loadScene / loadEnvironmentHdr / updateUniformBuffer
(src/sample_example.cpp:82-106, 164-173) without Vulkan.
"""
import ctypes as C
import os
import numpy as np
from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.environ.get("RESTIR_HOST_LIB") or os.path.join(_HERE, "host", "librestir_host.so")   # env override: the sanitizer build (tests/test_ingest_fuzz.py)
_lib = None

def host_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} missing: run __graft_entry__.build() first")
        L = C.CDLL(HOST_LIB_PATH)
        L.rth_scene_create.restype = C.c_void_p
        L.rth_env_create.restype = C.c_void_p
        L.rth_env_integral.restype = C.c_float
        L.rth_env_average.restype = C.c_float
        for name, args in {
            "rth_scene_destroy": [C.c_void_p], "rth_scene_load": [C.c_void_p, C.c_char_p],
            "rth_scene_make_procedural": [C.c_void_p, C.c_int, C.c_float, C.c_uint32], "rth_scene_save_gltf": [C.c_void_p, C.c_char_p], "rth_decode_jpeg": [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
            "rth_decode_png": [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t], "rth_write_png": [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int],
            "rth_scene_set_camera": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float],
            "rth_scene_get_camera_pose": [C.c_void_p, C.c_void_p], "rth_scene_update_camera": [C.c_void_p, C.c_int, C.c_int],
            "rth_scene_get_camera": [C.c_void_p, C.c_void_p], "rth_scene_light_weights": [C.c_void_p, C.c_void_p, C.c_void_p],
            "rth_scene_stats": [C.c_void_p, C.c_void_p], "rth_scene_desc": [C.c_void_p, C.c_void_p, C.c_void_p],
            "rth_env_destroy": [C.c_void_p], "rth_env_load": [C.c_void_p, C.c_char_p], "rth_env_set": [C.c_void_p, C.c_void_p, C.c_int, C.c_int],
            "rth_env_make_sky": [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint32], "rth_env_integral": [C.c_void_p], "rth_env_average": [C.c_void_p],
            "rth_env_width": [C.c_void_p], "rth_env_height": [C.c_void_p], "rth_env_get_accel": [C.c_void_p, C.c_void_p],
            "rth_default_state": [C.c_void_p], "rth_alias_table": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
        }.items():
            getattr(L, name).argtypes = args
        _lib = L
    return _lib


class HdrSampling:
    """hdr_sampling.hpp:38-66 (data products only)."""
    def __init__(self):
        self._h = host_lib().rth_env_create()
    def __del__(self):
        if getattr(self, "_h", None) and host_lib is not None:      # (module globals are gone at interpreter shutdown)
            host_lib().rth_env_destroy(self._h); self._h = None
    def loadEnvironment(self, path):
        return host_lib().rth_env_load(self._h, path.encode()) == 0
    def setEnvironment(self, rgba):
        a = np.ascontiguousarray(rgba, dtype=np.float32)
        host_lib().rth_env_set(self._h, a.ctypes.data, a.shape[1], a.shape[0])
    def makeSyntheticSky(self, w, h, sun_peak=5e4, seed=7):
        host_lib().rth_env_make_sky(self._h, w, h, sun_peak, seed)
    def getIntegral(self): return host_lib().rth_env_integral(self._h)
    def getAverage(self): return host_lib().rth_env_average(self._h)
    @property
    def size(self): return host_lib().rth_env_width(self._h), host_lib().rth_env_height(self._h)
    def accel(self):
        w, h = self.size
        a = np.zeros(w * h, dtype=np.dtype([("alias", "<i4"), ("q", "<f4"), ("pdf", "<f4"), ("aliasPdf", "<f4")]))
        host_lib().rth_env_get_accel(self._h, a.ctypes.data)
        return a


class Scene:
    """scene.hpp:59-80: setup / load / updateCamera / getCamera / getStat + light weights."""
    STAT_NAMES = ["triangles", "instancedTriangles", "vertices", "primMeshes", "nodes", "materials", "textures", "puncLights", "trigLights"]
    def __init__(self):
        self._h = host_lib().rth_scene_create()
    def __del__(self):
        if getattr(self, "_h", None) and host_lib is not None:
            host_lib().rth_scene_destroy(self._h); self._h = None
    def load(self, filename):
        return host_lib().rth_scene_load(self._h, filename.encode()) == 0
    def saveGltf(self, filename):
        """Write the loaded scene as a self-contained .gltf (host/gltf_loader.cpp writer; exchange/test utility)."""
        return host_lib().rth_scene_save_gltf(self._h, filename.encode()) == 0
    def makeProcedural(self, kind, scale=1.0, seed=1):
        if host_lib().rth_scene_make_procedural(self._h, kind, scale, seed) != 0:
            raise RuntimeError("procedural scene generation failed")
        return self
    def setCamera(self, eye, center, up=(0, 1, 0), fov=45.0):
        e, c, u = (np.asarray(v, dtype=np.float32) for v in (eye, center, up))
        host_lib().rth_scene_set_camera(self._h, e.ctypes.data, c.ctypes.data, u.ctypes.data, fov)
    def cameraPose(self):
        out = np.zeros(10, dtype=np.float32)
        host_lib().rth_scene_get_camera_pose(self._h, out.ctypes.data)
        return out[0:3].copy(), out[3:6].copy(), out[6:9].copy(), float(out[9])
    def updateCamera(self, width, height):
        host_lib().rth_scene_update_camera(self._h, width, height)
    def getCamera(self):
        cam = abi.SceneCamera()
        host_lib().rth_scene_get_camera(self._h, C.byref(cam))
        return cam
    def getStat(self):
        out = np.zeros(9, dtype=np.uint64)
        host_lib().rth_scene_stats(self._h, out.ctypes.data)
        return dict(zip(self.STAT_NAMES, (int(v) for v in out)))
    @property
    def lightWeights(self):
        p, t = C.c_float(), C.c_float()
        host_lib().rth_scene_light_weights(self._h, C.byref(p), C.byref(t))
        return p.value, t.value
    def desc(self, env=None):
        d = abi.SceneDesc()
        host_lib().rth_scene_desc(self._h, env._h if env is not None else None, C.byref(d))
        return d


def scene_digest(desc):
    """SHA-256 (first 16 hex digits) of everything rt_upload_scene reads from a scene description — prim meshes (20 B), vertices (32 B), indices, instances (56 B),
    materials (80 B), every texture's size / sampler and texels, punctual (80 B) and triangle (96 B) lights, light info — per part and over all of them.  Two scenes with
    the same digest upload the same bytes: the same frames follow (bench.py --via-gltf: the procedural scene against itself written to glTF and read back by Scene::load)."""
    import hashlib

    def raw(ptr, nbytes):
        return bytes((C.c_char * nbytes).from_address(ptr)) if ptr and nbytes else b""
    parts = {"primMeshes": raw(desc.primMeshes, desc.numPrimMeshes * 20), "vertices": raw(desc.vertices, desc.numVertices * 32), "indices": raw(desc.indices, desc.numIndices * 4),
             "instances": raw(desc.instances, desc.numInstances * 56), "materials": raw(desc.materials, desc.numMaterials * 80),
             "puncLights": raw(desc.puncLights, desc.lightInfo.puncLightSize * 80), "trigLights": raw(desc.trigLights, desc.lightInfo.trigLightSize * 96),
             "lightInfo": bytes(desc.lightInfo)}
    h = hashlib.sha256()
    n = int(desc.numTextures)
    if desc.textures and n:
        t = np.frombuffer((C.c_char * (n * 32)).from_address(desc.textures), dtype=np.dtype([("ptr", "<u8"), ("w", "<i4"), ("h", "<i4"), ("rest", "<i4", 4)]))
        for k in range(n):
            h.update(t[k]["w"].tobytes() + t[k]["h"].tobytes() + t[k]["rest"][:3].tobytes())
            h.update(raw(int(t[k]["ptr"]), int(t[k]["w"]) * int(t[k]["h"]) * 4))
    out = {k: hashlib.sha256(v).hexdigest()[:16] for k, v in parts.items()}
    out["textures"] = h.hexdigest()[:16]
    out["all"] = hashlib.sha256("".join(out[k] for k in sorted(out)).encode()).hexdigest()[:16]
    return out


def default_state(width, height, scene=None, env=None, time=1000):
    """RtxState as SampleExample fills it: defaults (sample_example.hpp:154-184) + the derived constants of
    sample_example.cpp:87 (lightLuminIntegInv) and :104-105 (fireflyClampThreshold, envMapLuminIntegInv)."""
    st = abi.RtxState()
    host_lib().rth_default_state(C.byref(st))
    st.size.x, st.size.y = width, height
    st.time = time
    if scene is not None:
        p, t = scene.lightWeights
        st.lightLuminIntegInv = 1.0 / (p + t) if (p + t) > 0 else float("inf")
    if env is not None:
        st.fireflyClampThreshold = env.getIntegral() * 4.0
        st.envMapLuminIntegInv = 1.0 / env.getIntegral()
    return st


def decode_jpeg(data):
    """host/jpeg_decoder.cpp on a bytes object -> (H, W, 4) uint8 BGRA array, or None when the stream is not supported."""
    buf = np.frombuffer(data, dtype=np.uint8)
    w, h = C.c_int(), C.c_int()
    out = np.empty(64 << 20, dtype=np.uint8)
    rc = host_lib().rth_decode_jpeg(buf.ctypes.data, buf.size, C.byref(w), C.byref(h), out.ctypes.data, out.size)
    if rc != 0:
        return None
    return out[: w.value * h.value * 4].reshape(h.value, w.value, 4).copy()


def write_png(path, rgba, keep_alpha=False):
    """host/png_writer.cpp: (H, W, 4) uint8 array, R first (the layout of BUF_LDR) -> 8-bit PNG file."""
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = a.shape[:2]
    if host_lib().rth_write_png(path.encode(), a.ctypes.data, w, h, 1 if keep_alpha else 0) != 0:
        raise IOError(f"cannot write {path}")


def decode_png(data):
    """host/gltf_loader.cpp's PNG reader on a bytes object -> (H, W, 4) uint8 BGRA array, or None."""
    buf = np.frombuffer(data, dtype=np.uint8)
    w, h = C.c_int(), C.c_int()
    out = np.empty(64 << 20, dtype=np.uint8)
    if host_lib().rth_decode_png(buf.ctypes.data, buf.size, C.byref(w), C.byref(h), out.ctypes.data, out.size) != 0:
        return None
    return out[: w.value * h.value * 4].reshape(h.value, w.value, 4).copy()
