"""Full-size parity of ALL 12 dispatches against the oracle, warm history, BASELINE configs 3, 4 and 5.

The oracle cannot render 1080p frames of a multi-million-triangle scene in seconds, but it can render row ranges
(`run_stage(..., rowBegin, rowEnd)`).  For a 16-row band of frame f the test feeds the oracle what the frame consumes —
the GPU's own history of frame f-1 (G-buffer, direct + indirect reservoirs, light ids; that history was itself produced by
stages held to the oracle in the small-frame tests) — and runs, on the oracle, exactly the row ranges the band depends on:

    compose          rows [y0, y1)
    direct A-Trous   level 3 on the band, level 2 on +-16, level 1 on +-24, level 0 on +-28; noisy direct on +-30
    indirect A-Trous level 4 on the band's half rows, 3 on +-32, 2 on +-48, 1 on +-56, 0 on +-60; noisy indirect on +-62 half rows
    indirect stage   those +-62 half rows  =>  direct stage (G-buffer + motion + reservoirs) on +-124 full rows

so every buffer the GPU leaves behind for the band — G-buffer, motion, direct / indirect reservoirs, light ids, the composed
direct and indirect result images (= all 4 + 5 filter levels + compose) and the last filter temporaries — is compared with the
oracle bit for bit: no stage is HIP-vs-HIP only at full size.  Bands are drawn from a fixed-seed generator.
"""
import os
import numpy as np
import pytest
from helpers import abi, host, make_scene

pytestmark = pytest.mark.gpu
BAND = 16


def _clip(a, b, n):
    return max(0, a), min(n, b)


def oracle_band(o, st, f, y0, y1, W, H):
    """run on the oracle every row range the 16-row band [y0, y1) of frame f depends on (see module docstring)"""
    Hh = H // 2
    h0, h1 = y0 // 2, (y1 + 1) // 2
    a, b = _clip(min(y0 - 30, 2 * (h0 - 62)), max(y1 + 30, 2 * (h1 + 62)), H)
    o.run_stage(st, f, abi.STAGE_DIRECT, 0, a, b)
    ia, ib = _clip(h0 - 62, h1 + 62, Hh)
    o.run_stage(st, f, abi.STAGE_INDIRECT, 0, ia, ib)
    if st.denoise > 0:
        for level, reach in ((0, 28), (1, 24), (2, 16), (3, 0)):
            la, lb = _clip(y0 - reach, y1 + reach, H)
            o.run_stage(st, f, abi.STAGE_DENOISE_DIRECT, level, la, lb)
        for level, reach in ((0, 60), (1, 56), (2, 48), (3, 32), (4, 0)):
            la, lb = _clip(h0 - reach, h1 + reach, Hh)
            o.run_stage(st, f, abi.STAGE_DENOISE_INDIRECT, level, la, lb)
    o.run_stage(st, f, abi.STAGE_COMPOSE, 0, y0, y1)
    return h0, h1


def compare_band(got, o, f, y0, y1, h0, h1, W, H):
    cur = f & 1
    full = [(abi.BUF_GBUFFER0 + cur, 16), (abi.BUF_MOTION, 4), (abi.BUF_DIRECT_RESV0 + cur, 36), (abi.BUF_LIGHT_ID0 + cur, 4),
            (abi.BUF_DIRECT_RESULT0 + cur, 16), (abi.BUF_INDIRECT_RESULT0 + cur, 16), (abi.BUF_DENOISE_DIR_A, 16)]
    for buf, elem in full:
        g = got[buf].reshape(-1, W * elem)[y0:y1]
        r = o.readback(buf).reshape(-1, W * elem)[y0:y1]
        nbad = int((g.view(np.uint32) != r.view(np.uint32)).sum())
        assert nbad == 0, f"frame {f} rows {y0}..{y1} {abi.BUFFER_NAMES[buf]}: {nbad} words differ from the oracle"
    g = got[abi.BUF_INDIRECT_RESV0 + cur].reshape(-1, (W // 2) * 76)[h0:h1]
    r = o.readback(abi.BUF_INDIRECT_RESV0 + cur).reshape(-1, (W // 2) * 76)[h0:h1]
    assert np.array_equal(g, r), f"frame {f} half rows {h0}..{h1} indirect reservoirs"
    # half-res images live in the top-left quarter of full-pitch images: the filtered indirect colour (level 4 output, HDR)
    g = got[abi.BUF_DENOISE_IND_B].reshape(H, W * 16)[h0:h1, :(W // 2) * 16]
    r = o.readback(abi.BUF_DENOISE_IND_B).reshape(H, W * 16)[h0:h1, :(W // 2) * 16]
    assert np.array_equal(g, r), f"frame {f} half rows {h0}..{h1} filtered indirect colour"


def run_config(kind, scale, env_size, tweak, nframes, moving, bands_seed, tri_range, W=1920, H=1080, orbit_deg=0.0):
    from restir_amd.renderer import Renderer
    from oracle.binding import Oracle
    sc, env = make_scene(kind, scale, 1, env_size)
    st = host.default_state(W, H, sc, env)
    tweak(st)
    desc = sc.desc(env)
    eye, center, up, fov = sc.cameraPose()
    cams = []
    sc.updateCamera(W, H)
    rel = eye - center
    for f in range(nframes):
        if orbit_deg:      # the camera orbits its centre of interest
            a = np.deg2rad(orbit_deg * (f + 1))
            rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=np.float32)
            sc.setCamera(center + rot @ rel, center, up, fov)
        elif moving:
            sc.setCamera(eye + np.array([0.06 * f, 0.015 * f, -0.05 * f], dtype=np.float32), center, up, fov)
        sc.updateCamera(W, H); cams.append(sc.getCamera())
    prev = os.environ.get("RESTIR_OVERLAP")
    os.environ["RESTIR_OVERLAP"] = "2"          # the product's default schedule: frames in flight
    try:
        r = Renderer().setup(0); r.load_scene(desc); r.update(W, H)
    finally:
        if prev is None: os.environ.pop("RESTIR_OVERLAP", None)
        else: os.environ["RESTIR_OVERLAP"] = prev
    assert tri_range[0] < r.accel_stats()["triangles"] < tri_range[1]
    f_last = nframes - 1
    for f in range(f_last):
        st.time = 9000 + f; r.set_camera(cams[f]); r.run(st, f)
    last = (f_last & 1) ^ 1
    hist_ids = [abi.BUF_GBUFFER0 + last, abi.BUF_DIRECT_RESV0 + last, abi.BUF_LIGHT_ID0 + last, abi.BUF_INDIRECT_RESV0 + last]
    hist = {b: r.readback(b) for b in hist_ids}                         # frame f_last - 1 = history of frame f_last
    # ... and what frame f_last - 2 left in the buffers frame f_last writes: pixels that return early (miss, emitter) leave their reservoir slot and light
    # id untouched (reference quirk), so those slots are INPUT of the frame too.  Thin geometry against the sky under a moving camera (the
    # cables and rails of the `real` exterior scene) makes such pixels common; round 5 found the comparison blind to it.
    cur_ = f_last & 1
    for b in (abi.BUF_DIRECT_RESV0 + cur_, abi.BUF_LIGHT_ID0 + cur_, abi.BUF_INDIRECT_RESV0 + cur_):
        hist[b] = r.readback(b)
    st.time = 9000 + f_last; r.set_camera(cams[f_last]); r.run(st, f_last)
    cur = f_last & 1
    keep = [abi.BUF_GBUFFER0 + cur, abi.BUF_MOTION, abi.BUF_DIRECT_RESV0 + cur, abi.BUF_LIGHT_ID0 + cur, abi.BUF_INDIRECT_RESV0 + cur,
            abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_INDIRECT_RESULT0 + cur, abi.BUF_DENOISE_DIR_A, abi.BUF_DENOISE_IND_B]
    got = {b: r.readback(b) for b in keep}
    r.destroy()
    img = got[abi.BUF_DIRECT_RESULT0 + cur].view(np.float32)
    assert np.isfinite(img).all() and img.max() > 0.01
    if moving or orbit_deg:
        mv = got[abi.BUF_MOTION].view(np.int16).reshape(H, W, 2)
        assert (mv[..., 0] != np.arange(W)[None, :]).mean() > 0.2       # reprojection is not the identity
    o = Oracle(0); o.upload_scene(desc); o.resize(W, H)
    for b, data in hist.items():
        o.upload_history(b, data)
    o.set_camera(cams[f_last])
    rng = np.random.default_rng(bands_seed)
    bands = sorted(int(2 * rng.integers(0, (H - BAND) // 2)) for _ in range(3))
    warm = 0
    for y0 in bands:
        h0, h1 = oracle_band(o, st, f_last, y0, y0 + BAND, W, H)
        compare_band(got, o, f_last, y0, y0 + BAND, h0, h1, W, H)
        resv = got[abi.BUF_DIRECT_RESV0 + cur].view(np.uint32).reshape(H, W, 9)[y0:y0 + BAND]
        warm += int((resv[..., 7] > st.RISSampleNum).sum())             # M above one frame's candidates => history was merged
    assert f_last == 0 or warm > 0, "no pixel of the compared bands reused history"
    return bands


def test_config3_sponza_class_1080p_all_stages():
    """BASELINE config 3: Sponza-class ~262 k triangles + HDR env, 1920x1080, DI + GI maxDepth 2 (one indirect bounce), MIS on,
    denoise on; frame 2 (warm reservoirs, static camera as §8d measures it)."""
    def tweak(st):
        st.maxDepth = 2; st.MIS = 1; st.denoise = 1
    run_config(abi.PROC_SPONZA, 1.0, (2048, 1024), tweak, nframes=3, moving=False, bands_seed=3, tri_range=(2.2e5, 3.0e5))


def test_config3_sponza_1k_textures_1080p_all_stages():
    """the scene `bench.py --config 3` TIMES since round 5 (PROC_SPONZA_1K: the same atrium with SURVEY 8(d)'s 1k^2 textures uploaded at full size) — the round-5 verdict found
    that no -m gpu test rendered it: same state as the test above, frame 2, three other bands"""
    def tweak(st):
        st.maxDepth = 2; st.MIS = 1; st.denoise = 1
    run_config(abi.PROC_SPONZA_1K, 1.0, (2048, 1024), tweak, nframes=3, moving=False, bands_seed=13, tri_range=(2.2e5, 3.0e5))


def test_config4_bistro_exterior_class_1080p_all_stages_moving_camera():
    """BASELINE config 4 (single-GPU leg): Bistro-Exterior-class ~2.8 M triangles with alpha-masked foliage + HDR env, 1920x1080,
    defaults (maxDepth 4, MIS, denoise), frame 3 under a moving camera (temporal reuse through real reprojection)."""
    run_config(abi.PROC_BISTRO_EXT, 1.0, (2048, 1024), lambda st: None, nframes=4, moving=True, bands_seed=4, tri_range=(2.6e6, 3.0e6))


def test_config5_bistro_interior_class_4k_all_stages_orbiting_camera():
    """BASELINE config 5 (single-GPU leg): Bistro-Interior-class ~1.0 M triangles with ~2 k emissive triangles, 3840x2160, DI + GI with
    temporal reuse under a camera orbiting 0.5 degrees per frame; frame 3, three bands, all 12 dispatches."""
    run_config(abi.PROC_BISTRO_INT, 1.0, (512, 256), lambda st: None, nframes=4, moving=False, bands_seed=5, tri_range=(0.8e6, 1.3e6), W=3840, H=2160, orbit_deg=0.5)


def test_config4_real_footprint_1080p_all_stages_moving_camera():
    """BASELINE config 4 on the `real` footprint variant of the exterior scene (round 5: 147 materials, 251 textures = 3.3 GB of BGRA8 texels uploaded at full size like
    src/scene.cpp:554-646, 16 distinct cut-out cards, rails / cables / awning strips): 1920x1080, defaults, frame 3 under a moving camera, three bands, all 12 dispatches."""
    run_config(abi.PROC_BISTRO_EXT_REAL, 1.0, (2048, 1024), lambda st: None, nframes=4, moving=True, bands_seed=6, tri_range=(2.6e6, 3.0e6))
