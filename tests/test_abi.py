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


def test_header_is_valid_c_and_cpp(tmp_path):
    """include/rt_abi.h and include/rt_detmath.h are the boundary: they must compile as plain C11 and as C++17 on their own."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    (tmp_path / "t.c").write_text('#include "rt_abi.h"\n#include "rt_detmath.h"\nint main(void) { rt_state s; (void)s; return (int)(rt_exp(0.0f) != 1.0f) + (int)(sizeof(rt_tonemapper) != 48); }\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", inc, str(tmp_path / "t.c"), "-o", str(tmp_path / "tc"), "-lm"])
    assert subprocess.call([str(tmp_path / "tc")]) == 0
    (tmp_path / "t.cpp").write_text('#include "rt_abi.h"\n#include "rt_detmath.h"\nint main() { rt_scene_desc d{}; (void)d; return rt_sin(0.0f) != 0.0f; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", inc, str(tmp_path / "t.cpp"), "-o", str(tmp_path / "tcpp")])
    assert subprocess.call([str(tmp_path / "tcpp")]) == 0
