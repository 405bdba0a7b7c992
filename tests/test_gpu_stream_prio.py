"""rt_set_stream_priorities / rt_tune_stream_priorities (ABI 2.2): the priorities of the frames-in-flight schedule's streams change when kernels run, never what
they compute; the load-time tuner leaves the history cold, so a tuned context renders the frames of a context that was never tuned."""
import numpy as np
import pytest
from helpers import abi, host, make_scene, frame_buffers

pytestmark = pytest.mark.gpu
W, H, FRAMES = 320, 208, 4


def _render(prepare):
    from restir_amd.renderer import Renderer
    sc, env = make_scene(abi.PROC_SPONZA, 0.02, 1, (128, 64))
    st = host.default_state(W, H, sc, env)
    r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
    sc.updateCamera(W, H); r.set_camera(sc.getCamera())
    info = prepare(r, st)
    eye, center, up, fov = sc.cameraPose()
    for f in range(FRAMES):
        st.time = 300 + f
        sc.setCamera(eye + np.array([0.04 * f, 0.01 * f, -0.02 * f], dtype=np.float32), center, up, fov)
        sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
    out = {b: r.readback(b) for b in frame_buffers(FRAMES - 1)}
    r.destroy()
    return out, info


def test_tuned_context_renders_the_same_frames():
    ref, _ = _render(lambda r, st: None)
    got, info = _render(lambda r, st: r.tune_stream_priorities(st, 4))
    assert info["chosen"][0] in (-1, 0, 1) and info["chosen"][1] in (-1, 0, 1) and len(info["ms_per_frame"]) == 5 and all(v > 0 for v in info["ms_per_frame"].values())
    for b in ref:
        assert np.array_equal(ref[b], got[b]), abi.BUFFER_NAMES[b]


@pytest.mark.parametrize("levels", [(0, 1), (0, -1), (1, 1), (-1, -1)])
def test_every_priority_setting_gives_the_same_bits(levels):
    ref, _ = _render(lambda r, st: None)
    got, _ = _render(lambda r, st: r.set_stream_priorities(*levels))
    for b in ref:
        assert np.array_equal(ref[b], got[b]), (abi.BUFFER_NAMES[b], levels)
