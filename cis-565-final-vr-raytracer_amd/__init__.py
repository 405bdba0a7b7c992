"""restir_amd — MI355X-native ReSTIR DI+GI frame path (drop-in for the per-frame hot path of
IwakuraRein/CIS-565-Final-VR-Raytracer).  The directory is named `cis-565-final-vr-raytracer_amd`; because of the
hyphens it is imported through the `restir_amd` shim at the repo root (restir_amd.py).

  abi       ctypes mirrors of include/rt_abi.h
  host      Scene / HdrSampling (host/*.cpp through librestir_host.so)
  renderer  Renderer over the C-ABI of csrc/librestir_hip.so (HIP, gfx950) — raises if the library is missing
"""
from . import abi  # noqa: F401
