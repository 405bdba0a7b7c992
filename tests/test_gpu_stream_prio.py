"""Stream priorities of the frames-in-flight schedule (ABI 2.2 / 2.3): rt_set_stream_priorities / rt_get_stream_priorities, the rule a context applies on its first three
("probe") frames — they run every stage alone, are timed, and the last one decides before the two other streams exist —, what re-opens that decision (round 6: resize, a
denoise toggle), and rt_get_streams / rt_get_stream_layout for hosts that issue the stages themselves.  Priorities change when kernels run, never what they compute."""
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


def test_probe_frame_rule_decides_and_reports(monkeypatch):
    monkeypatch.delenv("RESTIR_PRIO", raising=False); monkeypatch.delenv("RESTIR_PRIO_PROBE", raising=False)
    _, (before, after) = _render(lambda r, st: None)
    assert before["decided"] is False and before["filter_share"] is None
    assert after["decided"] is True and after["filter_share"] is not None and after["filter_share"] > 0
    from restir_amd.renderer import PRIO_FILTER_SHARE
    assert after["chosen"] == [1, 1 if after["filter_share"] >= PRIO_FILTER_SHARE else 0]
    # two frames are not enough: the third probe frame decides (warm history), and only then do the two other streams exist
    _, (_, mid) = _render(lambda r, st: None, frames=2)
    assert mid["decided"] is False and mid["filter_share"] is not None


def test_resize_and_denoise_toggle_reopen_the_decision(monkeypatch):
    """advisor finding of round 5: the decision was taken once, on whatever the first frame happened to be (a denoise = 0 frame measures compose alone)"""
    monkeypatch.delenv("RESTIR_PRIO", raising=False); monkeypatch.delenv("RESTIR_PRIO_PROBE", raising=False)
    from restir_amd.renderer import Renderer
    sc, env = make_scene(abi.PROC_SPONZA, 0.02, 1, (128, 64))
    st = host.default_state(W, H, sc, env)
    r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
    sc.updateCamera(W, H); r.set_camera(sc.getCamera())
    f = 0
    def frames(n):
        nonlocal f
        for _ in range(n):
            st.time = 300 + f; r.run(st, f); f += 1
    st.denoise = 0
    frames(4)
    off = r.stream_priorities()
    from restir_amd.renderer import PRIO_FILTER_SHARE
    assert off["decided"] and off["filter_share"] < PRIO_FILTER_SHARE and off["chosen"] == [1, 0]      # compose alone is a few percent of the traced stages
    st.denoise = 1
    frames(1)
    assert r.stream_priorities()["decided"] is False                                        # re-opened by the toggle: probing again
    frames(3)
    on = r.stream_priorities()
    assert on["decided"] and on["filter_share"] > off["filter_share"]
    r.update(W, H + 16); st.size.y = H + 16                                                  # rt_resize
    assert r.stream_priorities()["decided"] is False
    # an explicit choice survives all of it
    r.set_stream_priorities(0, 1); r.update(W, H); st.size.y = H
    sc.updateCamera(W, H); r.set_camera(sc.getCamera())
    f = 0; frames(2)
    assert r.stream_priorities() == {"chosen": [0, 1], "filter_share": None, "decided": True}
    r.destroy()


def test_get_streams_hands_out_the_contexts_own_streams(monkeypatch):
    monkeypatch.delenv("RESTIR_PRIO", raising=False)
    from restir_amd.renderer import Renderer
    r = Renderer().setup(0)
    lay0 = r.stream_layout()
    assert lay0["creation_index"]["main"] >= 0 and lay0["creation_index"]["ind"] == -1 and lay0["creation_index"]["side"] == -1
    s = r.streams()
    assert all(s[k] for k in ("main", "ind", "side")) and len({s["main"], s["ind"], s["side"]}) == 3
    lay = r.stream_layout()
    ci = lay["creation_index"]
    assert ci["main"] < ci["side"] < ci["ind"] and ci["ind"] == ci["side"] + 1                  # filter stream first, then the indirect stream, nothing in between
    assert lay["library_streams_created"] == lay0["library_streams_created"] + 2
    assert r.streams() == s and r.stream_layout() == lay                                          # idempotent
    assert r.stream_priorities()["decided"] is True                                               # the levels the streams were created with stand
    r.destroy()


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
