"""GPU: the HIP path (through the C-ABI, both builds of the traced kernels) against tests/golden/ref_stage_vectors.npz — the per-pixel
output of the REFERENCE's own shaders compiled for the CPU (tests/golden/make_ref_stage_vectors.py).  Every screen-space buffer
of every frame, bit for bit; the oracle is not involved in this comparison."""
import pytest
from helpers import RendererBackend
from test_ref_stages import GOLDEN_CASES, check_against_vectors

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("latency", [False, True], ids=["throughput", "latency"])
@pytest.mark.parametrize("name", sorted(GOLDEN_CASES))
def test_hip_reproduces_reference_vectors(name, latency):
    from restir_amd.renderer import Renderer
    def mk(desc, W, H):
        r = Renderer().setup(0); r.load_scene(desc); r.update(W, H); r.set_traversal(2 if latency else 1)
        return RendererBackend(r)
    check_against_vectors(mk, name)
