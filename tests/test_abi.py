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
    assert lib.rt_abi_version() >> 16 == 2


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
    (tmp_path / "t.c").write_text('#include "rt_abi.h"\n#include "rt_detmath.h"\n#include "rt_cpus.h"\nint main(void) { rt_state s; (void)s; return (int)(rt_exp(0.0f) != 1.0f) + (int)(sizeof(rt_tonemapper) != 48) + (int)(rt_cpu_budget() < 1); }\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", inc, str(tmp_path / "t.c"), "-o", str(tmp_path / "tc"), "-lm"])
    assert subprocess.call([str(tmp_path / "tc")]) == 0
    (tmp_path / "t.cpp").write_text('#include "rt_abi.h"\n#include "rt_detmath.h"\n#include "rt_cpus.h"\nint main() { rt_scene_desc d{}; (void)d; return (rt_sin(0.0f) != 0.0f) + (rt_cpu_budget() < 1) + (rt_cpu_quota() < 0); }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", inc, str(tmp_path / "t.cpp"), "-o", str(tmp_path / "tcpp")])
    assert subprocess.call([str(tmp_path / "tcpp")]) == 0


def test_mgpu_band_planner_matches_the_python_planner():
    """rt_mgpu_plan_bands (csrc/mgpu.cpp, no GPU needed) and tiled.plan_bands implement one rule: same boundaries on random costs"""
    import ctypes as C
    import numpy as np
    from restir_amd import renderer, tiled
    lib = renderer.hip_lib()
    rng = np.random.default_rng(9)
    for _ in range(200):
        H = int(rng.integers(64, 2200)); world = int(rng.integers(1, min(8, (H + 15) // 16) + 1))
        stripes = (H + 15) // 16
        cost = (rng.random(stripes) ** 3 * 10 + 0.01).astype(np.float32)
        prev = np.array(tiled.equal_partition(H, world) if rng.integers(2) else tiled.plan_bands(H, world, rng.random(stripes) + 0.1), dtype=np.int32)
        use_prev = bool(rng.integers(2)); mv = int(rng.integers(0, 5))
        out = np.zeros(world + 1, dtype=np.int32)
        rc = lib.rt_mgpu_plan_bands(H, world, cost.ctypes.data, prev.ctypes.data if use_prev else None, mv if use_prev else -1, out.ctypes.data)
        assert rc == 0
        want = tiled.plan_bands(H, world, [float(c) for c in cost], prev=[int(p) for p in prev] if use_prev else None, max_move=mv if use_prev else None)
        assert list(out) == want, (H, world, list(out), want)
        assert out[0] == 0 and out[-1] == H and all(b % 16 == 0 for b in out[:-1]) and all(out[i + 1] > out[i] for i in range(world))
    assert lib.rt_mgpu_plan_bands(32, 4, cost.ctypes.data, None, -1, out.ctypes.data) != 0     # fewer stripes than ranks
