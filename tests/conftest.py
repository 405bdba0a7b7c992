import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_libs():
    """Build (if stale/missing) the host library, the HIP library (cross-compiles without a GPU) and the oracle."""
    import restir_amd  # noqa: F401
    from restir_amd import build
    build.build_host()
    if not os.path.exists(build.HIP_LIB) or os.path.exists("/opt/rocm/bin/hipcc"):
        build.build_hip()
    from oracle import binding
    binding.build()
    yield


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
