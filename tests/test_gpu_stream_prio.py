"""Stream priorities of the frames-in-flight schedule (ABI 2.2): rt_set_stream_priorities / rt_get_stream_priorities and the rule a context applies at its first frame (that
frame runs every stage alone, is timed, and decides before the two other streams exist).  Priorities change when kernels run, never what they compute."""
import numpy as np
import pytest
from helpers import abi, host, make_scene, frame_buffers

pytestmark = pytest.mark.gpu
W, H, FRAMES = 320, 208, 4


def _render(prepare, frames=FRAMES):
    from restir_amd.renderer import Renderer
    sc, env = make_scene(abi.PROC_SPONZA, 0.02, 1, (128, 64))
    st = host.default_state(W, H, sc, env)
    r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
    sc.updateCamera(W, H); r.set_camera(sc.getCamera())
    prepare(r, st)
    before = r.stream_priorities()
    eye, center, up, fov = sc.cameraPose()
    for f in range(frames):
        st.time = 300 + f
        sc.setCamera(eye + np.array([0.04 * f, 0.01 * f, -0.02 * f], dtype=np.float32), center, up, fov)
        sc.updateCamera(W, H); r.set_camera(sc.getCamera()); r.run(st, f)
    out = {b: r.readback(b) for b in frame_buffers(frames - 1)}
    info = (before, r.stream_priorities())
    r.destroy()
    return out, info


def test_first_frame_rule_decides_and_reports(monkeypatch):
    monkeypatch.delenv("RESTIR_PRIO", raising=False)
    _, (before, after) = _render(lambda r, st: None)
    assert before["decided"] is False and before["filter_share"] is None
    assert after["decided"] is True and after["filter_share"] is not None and after["filter_share"] > 0
    assert after["chosen"] == [1, 1 if after["filter_share"] >= 0.14 else 0]


def test_explicit_levels_and_the_environment_override_the_rule(monkeypatch):
    monkeypatch.delenv("RESTIR_PRIO", raising=False)
    _, (before, after) = _render(lambda r, st: r.set_stream_priorities(0, -1))
    assert before == {"chosen": [0, -1], "filter_share": None, "decided": True} and after == before
    monkeypatch.setenv("RESTIR_PRIO", "00+")
    _, (before, after) = _render(lambda r, st: None)
    assert before["chosen"] == [0, 1] and before["decided"] is True and after["filter_share"] is None


@pytest.mark.parametrize("levels", [None, (1, 0), (0, 1), (0, -1), (1, 1), (-1, -1)])
def test_every_priority_setting_gives_the_same_bits(levels, monkeypatch):
    """... and a context whose first frame ran serially (the rule) renders the frames of one that ran every frame in flight (explicit levels)"""
    monkeypatch.setenv("RESTIR_OVERLAP", "0")
    ref, _ = _render(lambda r, st: None)
    monkeypatch.delenv("RESTIR_OVERLAP")
    got, _ = _render((lambda r, st: None) if levels is None else (lambda r, st: r.set_stream_priorities(*levels)))
    for b in ref:
        assert np.array_equal(ref[b], got[b]), (abi.BUFFER_NAMES[b], levels)
