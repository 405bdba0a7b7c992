"""Display pass (post.frag:103-175, tonemapping.glsl) — §8f rank 1.
CPU: the oracle against an independent float64 numpy statement of the same formulas (<= 1 code value: float32 vs float64
and the dither threshold), plus properties.  GPU: HIP kernel == oracle, bit for bit, for every mode."""
import numpy as np
import pytest
from helpers import abi, host, make_scene


def _pcg3d(x, y):
    v = [x.astype(np.uint64), y.astype(np.uint64), np.zeros_like(x, dtype=np.uint64)]
    M = np.uint64(0xffffffff)
    v = [(c * np.uint64(1664525) + np.uint64(1013904223)) & M for c in v]
    v[0] = (v[0] + v[1] * v[2]) & M; v[1] = (v[1] + v[2] * v[0]) & M; v[2] = (v[2] + v[0] * v[1]) & M
    v = [c ^ (c >> np.uint64(16)) for c in v]
    v[0] = (v[0] + v[1] * v[2]) & M; v[1] = (v[1] + v[2] * v[0]) & M; v[2] = (v[2] + v[0] * v[1]) & M
    return [((c >> np.uint64(9)).astype(np.uint32) | np.uint32(0x3f800000)).view(np.float32).astype(np.float64) - 1.0 for c in v]


def _bilinear(img, fx, fy):
    h, w = img.shape[:2]
    x0, y0 = np.floor(fx), np.floor(fy)
    ax, ay = (fx - x0)[..., None], (fy - y0)[..., None]
    xa, xb = np.clip(x0, 0, w - 1).astype(int), np.clip(x0 + 1, 0, w - 1).astype(int)
    ya, yb = np.clip(y0, 0, h - 1).astype(int), np.clip(y0 + 1, 0, h - 1).astype(int)
    top = img[ya, xa] * (1 - ax) + img[ya, xb] * ax
    bot = img[yb, xa] * (1 - ax) + img[yb, xb] * ax
    return top * (1 - ay) + bot * ay


def _mips(img):
    """vkCmdBlitImage, linear filter, level by level (render_output.cpp:243-254)"""
    levels = [img[..., :3].astype(np.float64)]
    for _ in range(7):
        src = levels[-1]
        sh, sw = src.shape[:2]
        w, h = max(1, sw // 2), max(1, sh // 2)
        ys, xs = np.mgrid[0:h, 0:w]
        levels.append(_bilinear(src, (xs + 0.5) * (sw / w) - 0.5, (ys + 0.5) * (sh / h) - 0.5))
    return levels


def _local_adaptation(D, I, tm, dbg, factor):
    H, W, _ = D.shape
    ys, xs = np.mgrid[0:H, 0:W]
    u, v = (xs + 0.5) / W * tm.zoom, (ys + 0.5) / H * tm.zoom
    MD, MI = _mips(D), _mips(I)
    lw = np.array([0.2126, 0.7152, 0.0722])
    def lum(levels, i):
        h, w = levels[i].shape[:2]
        return _bilinear(levels[i], u * w - 0.5, v * h - 0.5) @ lw
    La = np.zeros((H, W)); done = np.zeros((H, W), bool)
    for i in range(7):
        if dbg == 1: v1, v2 = lum(MD, i) * factor, lum(MD, i + 1) * factor
        elif dbg == 2: v1, v2 = lum(MI, i) * factor, lum(MI, i + 1) * factor
        else: v1, v2 = (lum(MD, i) + lum(MI, i)) * factor, np.zeros((H, W))   # `v2 ==` at post.frag:91: undefined, 0 in this build
        hit = (np.abs(v1 - v2) / (tm.key * 4.0 / (2.0 ** i) ** 2 + v1) > 0.05) & ~done
        La = np.where(hit, v1, np.where(done, La, v2))
        done |= hit
    return La


def reference_tonemap(D, I, tm, dbg=0):
    """float64 numpy restatement written from the GLSL, not from the oracle."""
    H, W, _ = D.shape
    hdr = {abi.__dict__.get("DBG_DIRECT", 1): D[..., :3], 2: I[..., :3]}.get(dbg, D[..., :3] + I[..., :3]).astype(np.float64)
    if tm.autoExposure & 1:
        avg = {1: D[..., :3].reshape(-1, 3).mean(0), 2: I[..., :3].reshape(-1, 3).mean(0)}.get(dbg, D[..., :3].reshape(-1, 3).astype(np.float64).mean(0) + I[..., :3].reshape(-1, 3).astype(np.float64).mean(0))
        lum = avg @ np.array([0.2126, 0.7152, 0.0722])
        Yxyz = hdr @ np.array([0.3575761, 0.7151522, 0.1191920])
        Y = tm.key / lum * Yxyz
        if tm.autoExposure & 2:      # toneLocalExposure (post.frag:70-101) over the blitted mip pyramid
            Yd = Y / (1 + _local_adaptation(D, I, tm, dbg, tm.key / lum))
        else:
            Yd = Y * (1 + Y / (tm.Ywhite * tm.Ywhite)) / (1 + Y)
        with np.errstate(divide="ignore", invalid="ignore"):
            hdr = hdr / Yxyz[..., None] * Yd[..., None]
    def impl(c):
        A, B, C_, D_, E, F = 0.15, 0.50, 0.10, 0.20, 0.02, 0.30
        return (c * (A * c + C_ * B) + D_ * E) / (c * (A * c + B) + D_ * F) - E / F
    def gpow(c, e):
        return np.where(c > 0, np.power(np.maximum(c, 1e-300), e), 0.0)
    color = gpow(impl(hdr * tm.avgLum * 2.0) / impl(np.float64(11.2)), 1 / 2.2)
    ys, xs = np.mgrid[0:H, 0:W]
    noise = np.stack(_pcg3d(xs, ys), axis=-1)
    lin = gpow(color, 2.2)
    q = 1 / 255.0
    c0 = np.floor(gpow(lin, 1 / 2.2) / q) * q
    c1 = c0 + q
    a, b = gpow(c0, 2.2), gpow(c1, 2.2)
    discr = a * (1 - noise) + b * noise
    color = np.where(discr < lin, c1, c0)
    color = np.clip(0.5 * (1 - tm.contrast) + color * tm.contrast, 0, 1)
    color = gpow(color, 1 / tm.brightness)
    i = color @ np.array([0.299, 0.587, 0.114])
    color = i[..., None] * (1 - tm.saturation) + color * tm.saturation
    u = ((xs + 0.5) / W * tm.renderingRatio[0] - 0.5) * 2
    v = ((ys + 0.5) / H * tm.renderingRatio[1] - 0.5) * 2
    color = color * (1 - (u * u + v * v) * tm.vignette)[..., None]
    return np.floor(np.clip(color, 0, 1) * 255 + 0.5).astype(np.int32)


def _oracle_with_images(W, H, D, I, frames=0):
    from oracle.binding import Oracle
    sc, _ = make_scene(abi.PROC_CORNELL)
    o = Oracle(0); o.upload_scene(sc.desc(None)); o.resize(W, H)
    o.upload_history(abi.BUF_DIRECT_RESULT0 + (frames & 1), D.astype(np.float32))
    o.upload_history(abi.BUF_INDIRECT_RESULT0 + (frames & 1), I.astype(np.float32))
    return o


def _images(W, H, seed=3):
    rng = np.random.default_rng(seed)
    D = np.exp(rng.normal(-1.0, 1.5, (H, W, 4))).astype(np.float32); D[..., 3] = rng.random((H, W)) * 3
    I = np.exp(rng.normal(-2.0, 1.0, (H, W, 4))).astype(np.float32)
    D[0, :8, :3] = 0.0; I[0, :8, :3] = 0.0            # black pixels
    D[1, :4, :3] = 5000.0                             # far above white
    return D, I


TM_CASES = {
    "default": {},
    "auto_exposure": {"autoExposure": 1, "key": 0.35, "Ywhite": 0.8},
    "graded": {"contrast": 1.3, "brightness": 0.8, "saturation": 1.6, "vignette": 0.7, "avgLum": 2.5},
    "auto_local_bit": {"autoExposure": 3},
}


@pytest.mark.parametrize("name", list(TM_CASES))
def test_oracle_tonemap_matches_float64_reference(name):
    W, H = 96, 40
    D, I = _images(W, H)
    tm = abi.Tonemapper(**TM_CASES[name])
    o = _oracle_with_images(W, H, D, I)
    o.tonemap(tm, 0, 0)
    got = o.readback(abi.BUF_LDR).reshape(H, W, 4).astype(np.int32)
    ref = reference_tonemap(D, I, tm)
    assert (got[..., 3] == 255).all()
    diff = np.abs(got[..., :3] - ref)
    assert diff.max() <= 1, diff.max()                 # one code value: the dither comparison flips on float32/float64 ties
    assert (diff == 0).mean() > 0.97


def test_oracle_tonemap_properties():
    W, H = 64, 64
    ramp = np.zeros((H, W, 4), np.float32)
    ramp[..., :3] = np.linspace(0, 20, W, dtype=np.float32)[None, :, None]
    zero = np.zeros_like(ramp)
    o = _oracle_with_images(W, H, ramp, zero)
    o.tonemap(abi.Tonemapper(), 0, 0)
    img = o.readback(abi.BUF_LDR).reshape(H, W, 4)
    assert (img[:, 0, :3] <= 1).all()                               # black stays black up to one dithered code value
    col = img[..., 0].astype(np.int32).mean(0)
    assert (np.diff(col) >= -1.0).all() and col[-1] > 240           # monotone up to dither noise, saturates near white
    assert (np.abs(img[..., 0].astype(int) - img[..., 1]) <= 1).all()   # grey in => grey out (per-channel dither only)
    # debug views (post.frag:106-118)
    o.tonemap(abi.Tonemapper(), abi.__dict__.get("DBG_NORMAL", 4), 0)
    dbg = o.readback(abi.BUF_LDR).reshape(H, W, 4)
    want = np.floor(np.clip(ramp[..., :3], 0, 1) * 255 + 0.5).astype(np.uint8)
    assert np.array_equal(dbg[..., :3], want)
    # direct-only / indirect-only selectors
    o.tonemap(abi.Tonemapper(), 2, 0)
    z = o.readback(abi.BUF_LDR).reshape(H, W, 4)[..., :3]
    # the float32 Uncharted curve leaves ~4e-9 at 0 (D*E/(D*F) vs E/F), which the dither turns into code value 1 for ~0.1 % of the pixels
    assert z.max() <= 1 and (z == 0).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(TM_CASES))
def test_gpu_tonemap_bit_exact(name):
    from restir_amd.renderer import Renderer
    W, H = 200, 77
    D, I = _images(W, H, seed=11)
    tm = abi.Tonemapper(**TM_CASES[name])
    o = _oracle_with_images(W, H, D, I, frames=1)
    sc, _ = make_scene(abi.PROC_CORNELL)
    r = Renderer().setup(0); r.load_scene(sc.desc(None)); r.update(W, H)
    r.upload_history(abi.BUF_DIRECT_RESULT0 + 1, D); r.upload_history(abi.BUF_INDIRECT_RESULT0 + 1, I)
    for dbg in (0, 1, 2, 3, 4, 5):
        o.tonemap(tm, dbg, 1); r.tonemap(tm, dbg, 1)
        assert np.array_equal(r.readback(abi.BUF_LDR), o.readback(abi.BUF_LDR)), (name, dbg)


@pytest.mark.gpu
def test_gpu_tonemap_after_rendered_frames():
    """End to end: frames in flight -> rt_tonemap (joins the streams) -> RGBA8 equals the oracle's."""
    from restir_amd.renderer import Renderer
    from oracle.binding import Oracle
    W, H = 256, 144
    sc, env = make_scene(abi.PROC_SPONZA, 0.02, 1, (256, 128))
    st = host.default_state(W, H, sc, env)
    desc = sc.desc(env)
    o = Oracle(0); o.upload_scene(desc); o.resize(W, H)
    r = Renderer().setup(0); r.load_scene(desc); r.update(W, H)
    sc.updateCamera(W, H)
    for f in range(3):
        st.time = 1000 + f; sc.updateCamera(W, H)
        o.set_camera(sc.getCamera()); r.set_camera(sc.getCamera())
        o.render_frame(st, f); r.run(st, f)
    tm = abi.Tonemapper(autoExposure=1)
    o.tonemap(tm, 0, 2); r.tonemap(tm, 0, 2)
    a, b = r.readback(abi.BUF_LDR), o.readback(abi.BUF_LDR)
    assert np.array_equal(a, b)
    assert a.reshape(H, W, 4)[..., :3].std() > 10       # an actual picture
