"""The C-ABI shared library loads without a GPU and exports every symbol include/rt_abi.h declares."""
import ctypes as C
import os
import re
import numpy as np
from helpers import ROOT, abi
from restir_amd import renderer


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    body = src[src.index("typedef struct rt_ctx rt_ctx;"):]
    return sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", body)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(renderer.HIP_LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 18 and set(names) == set(renderer.ABI_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n
    lib.rt_abi_version.restype = C.c_uint32
    assert lib.rt_abi_version() >> 16 == 1


def test_struct_sizes_match_host_device_h():
    assert C.sizeof(abi.SceneCamera) == 336 and C.sizeof(abi.RtxState) == 100
    assert C.sizeof(abi.ImptSamp) == 16 and C.sizeof(abi.LightBufInfo) == 16


def test_error_convention_without_gpu():
    lib = renderer.hip_lib()
    assert lib.rt_create(None, 0) == -1                      # RT_ERR_INVALID_ARG, never throws
    assert b"NULL" in lib.rt_last_error(None)
    assert lib.rt_destroy(None) == -1 and lib.rt_resize(None, 4, 4) == -1
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        h = C.c_void_p()
        assert lib.rt_create(C.byref(h), 0) == -2            # RT_ERR_NO_DEVICE: the product has no CPU path
        assert not h.value and b"no CPU path" in lib.rt_last_error(None)
