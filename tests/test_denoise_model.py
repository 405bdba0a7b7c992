"""The oracle's A-Trous filters and compose pass against an independent float64 numpy statement of the same shaders, written
from the GLSL alone (denoise_common.glsl:15-55, denoise_direct.comp:19-71 + :139-172, denoise_indirect.comp:23-75 + :132-171,
compose.comp:23-43) and vectorised differently (one shifted array per tap instead of a loop per pixel).  Inputs are a real
noisy frame: the oracle's direct + indirect stage on a small textured scene.  Tolerance: 2e-4 relative + 1e-6 absolute
(float32 + rt_exp in the oracle, float64 + libm here; the 25-tap sums amplify little)."""
import ctypes as C
import numpy as np
from helpers import abi, host, make_scene
from oracle.binding import Oracle, lib

GAUSS = np.array([[.0030, .0133, .0219, .0133, .0030],
                  [.0133, .0596, .0983, .0596, .0133],
                  [.0219, .0983, .1621, .0983, .0219],
                  [.0133, .0596, .0983, .0596, .0133],
                  [.0030, .0133, .0219, .0133, .0030]], dtype=np.float32).astype(np.float64)
INVALID = 0xFF000000


def _mat(m):                      # column-major nvmath mat4 -> numpy [row, col]
    return np.array(list(m.m), dtype=np.float64).reshape(4, 4).T


def _geometry(g, cam, coords_x, coords_y, image_size):
    """loadThisGeometry (denoise_common.glsl:42-47) for the texels (coords_y, coords_x) of the full-res G-buffer; the ray is
    spawned for `image_size` (w, h) exactly as the shader does — the indirect filter passes 2q with the HALF-res size."""
    tex = g[coords_y, coords_x]
    depth = tex[..., 0].view(np.float32).astype(np.float64)
    packed = np.ascontiguousarray(tex[..., 1])
    normal = np.zeros(packed.shape + (3,), dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    for idx in np.ndindex(packed.shape):      # compress.glsl decode: pinned bit-exactly by tests/test_kat.py
        lib().orc_decompress_unit_vec(int(packed[idx]), out.ctypes.data)
        normal[idx] = out
    vi, pi = _mat(cam.viewInverse), _mat(cam.projInverse)
    u = (coords_x + 0.5) / image_size[0] * 2.0 - 1.0
    v = (coords_y + 0.5) / image_size[1] * 2.0 - 1.0
    target = np.stack([u, v, np.ones_like(u), np.ones_like(u)], -1) @ pi.T
    t3 = target[..., :3] / np.linalg.norm(target[..., :3], axis=-1, keepdims=True)
    direction = t3 @ vi[:3, :3].T                                       # viewInverse * vec4(normalize(target.xyz), 0): not re-normalised
    origin = vi[:3, 3]
    pos = origin + direction * depth[..., None]
    return normal.astype(np.float64), pos, tex[..., 3] & np.uint32(INVALID)


def _wavelet(img, normal, pos, mat, level, sig, indirect):
    h, w = mat.shape
    step = 1 << level
    lum = lambda c: 0.2126 * c[..., 0] + 0.7152 * c[..., 1] + 0.0722 * c[..., 2]  # noqa: E731
    total, weight_sum = np.zeros((h, w, 3)), np.zeros((h, w))
    yy, xx = np.mgrid[0:h, 0:w]
    for j in range(-2, 3):
        for i in range(-2, 3):
            qx, qy = xx + i * step, yy + j * step
            ok = (qx >= 0) & (qy >= 0) & (qx < w) & (qy < h)
            cx, cy = np.clip(qx, 0, w - 1), np.clip(qy, 0, h - 1)
            ok &= (mat[cy, cx] == mat) & (mat[cy, cx] != INVALID)
            cq = img[cy, cx]
            dcol = ((img - cq) ** 2).sum(-1) if indirect else np.abs(lum(img) - lum(cq))
            wcol = np.exp(-dcol / sig[0]) + 1e-2
            wnorm = np.minimum(1.0, np.exp(-((normal - normal[cy, cx]) ** 2).sum(-1) / sig[1]))
            wdep = np.exp(-((pos - pos[cy, cx]) ** 2).sum(-1) / sig[2]) + 1e-2
            wt = np.where(ok, wcol * wnorm * wdep * GAUSS[i + 2][j + 2], 0.0)
            total += cq * wt[..., None]
            weight_sum += wt
    res = np.where((weight_sum < 1e-5)[..., None], 0.0, total / np.maximum(weight_sum, 1e-300)[..., None])
    bad = np.isnan(res).any(-1) | (res < 0).any(-1) | (res > 1e8).any(-1) | (mat == INVALID)
    res[bad] = 0.0
    return res


def _close(got, want, what):
    err = np.abs(got - want) - (2e-4 * np.abs(want) + 1e-6)
    assert (err <= 0).all(), (what, float(err.max()), np.unravel_index(err.argmax(), err.shape))


def test_atrous_chains_and_compose_match_an_independent_model():
    W, H = 48, 32
    sc, env = make_scene(abi.PROC_SPONZA, 0.01, 1, (64, 32))
    st = host.default_state(W, H, sc, env)
    o = Oracle(0); o.upload_scene(sc.desc(env)); o.resize(W, H)
    sc.updateCamera(W, H)
    for f in range(2):                                   # frame 1: temporal history exists, cur = 1
        st.time = 300 + f; sc.updateCamera(W, H); cam = sc.getCamera(); o.set_camera(cam)
        o.run_stage(st, f, abi.STAGE_DIRECT); o.run_stage(st, f, abi.STAGE_INDIRECT)
        if f == 0:
            for l in range(4): o.run_stage(st, f, abi.STAGE_DENOISE_DIRECT, l)
            for l in range(5): o.run_stage(st, f, abi.STAGE_DENOISE_INDIRECT, l)
            o.run_stage(st, f, abi.STAGE_COMPOSE)
    cur = 1
    img = lambda b: o.readback(b).view(np.float32).reshape(H, W, 4)  # noqa: E731
    g = o.readback(abi.BUF_GBUFFER0 + cur).view(np.uint32).reshape(H, W, 4)
    noisy_d = img(abi.BUF_DIRECT_RESULT0 + cur)[..., :3].astype(np.float64)
    noisy_i = img(abi.BUF_DENOISE_IND_A)[:H // 2, :W // 2, :3].astype(np.float64)
    assert (g[..., 3] & INVALID != INVALID).sum() > W * H // 2 and noisy_d.max() > 0 and noisy_i.max() > 0

    # ---- direct chain: result -> A -> B -> A -> LDRToHDR -> result --------------------------------------------------------
    yy, xx = np.mgrid[0:H, 0:W]
    n, p, m = _geometry(g, cam, xx, yy, (W, H))
    sig = (st.sigLuminDirect, st.sigNormalDirect, st.sigDepthDirect)
    d = noisy_d
    for level in range(4):
        d = _wavelet(d, n, p, m, level, sig, indirect=False)
        o.run_stage(st, 1, abi.STAGE_DENOISE_DIRECT, level)
        if level < 3:
            _close(img(abi.BUF_DENOISE_DIR_A if level % 2 == 0 else abi.BUF_DENOISE_DIR_B)[..., :3], d, f"direct level {level}")
    d = d / (1.01 - d)                                                    # LDRToHDR, common.glsl:198-200
    _close(img(abi.BUF_DIRECT_RESULT0 + cur)[..., :3], d, "direct chain")

    # ---- indirect chain at half resolution: A -> B -> A -> thisIndirectResult (scratch) -> A -> LDRToHDR -> B --------------
    hh, hw = H // 2, W // 2
    yy, xx = np.mgrid[0:hh, 0:hw]
    n, p, m = _geometry(g, cam, xx * 2, yy * 2, (hw, hh))                # loadThisGeometry(q * 2, ..., indSize())
    sig = (st.sigLuminIndirect, st.sigNormalIndirect, st.sigDepthIndirect)
    ind = noisy_i
    for level in range(5):
        ind = _wavelet(ind, n, p, m, level, sig, indirect=True)
        o.run_stage(st, 1, abi.STAGE_DENOISE_INDIRECT, level)
    ind = ind / (1.01 - ind)
    got_i = img(abi.BUF_DENOISE_IND_B)[:hh, :hw, :3]
    _close(got_i, ind, "indirect chain")

    # ---- compose (modulate = 1): both images times the G-buffer albedo, indirect fetched at coord / 2 ----------------------
    assert st.modulate == 1 and st.denoise > 0
    o.run_stage(st, 1, abi.STAGE_COMPOSE)
    w3 = g[..., 3]
    albedo = np.stack([w3 & 0xff, (w3 >> 8) & 0xff, (w3 >> 16) & 0xff], -1).astype(np.float64) / 255.0
    yy, xx = np.mgrid[0:H, 0:W]
    _close(img(abi.BUF_DIRECT_RESULT0 + cur)[..., :3], d * albedo, "compose direct")
    _close(img(abi.BUF_INDIRECT_RESULT0 + cur)[..., :3], ind[yy // 2, xx // 2] * albedo, "compose indirect")
