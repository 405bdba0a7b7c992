"""Sun & sky environment (sun_and_sky.glsl:453-601; env_sampling.glsl:111-125; pathtrace.glsl:40-72) — §8f rank 3.
CPU: the oracle against an independent float64 statement of the model written from the GLSL.  GPU: bit-exact frames."""
import math
import numpy as np
import pytest
from helpers import abi, host, make_scene, frame_buffers, compare_buffers, RendererBackend

PI = 3.1415926535


def _norm(v):
    v = np.asarray(v, float); return v / math.sqrt(float(v @ v))


def _perez(A, B, C, D, E, ct, g, cg, ts, cts):
    return ((1 + A * math.exp(B / ct)) * (1 + C * math.exp(D * g) + E * cg * cg)) / ((1 + A * math.exp(B)) * (1 + C * math.exp(D * ts) + E * cts * cts))


def _env_color(sun, d, T):
    ts = math.acos(sun[2])
    chi = (4.0 / 9.0 - T / 120.0) * (PI - 2 * ts)
    L = 1000.0 * ((4.0453 * T - 4.9710) * math.tan(chi) - 0.2155 * T + 2.4192)
    cg = float(sun @ d)
    cgl = min(max(cg, 0.0), 1.0) if cg <= 1 else 2 - cg
    L *= _perez(0.178721 * T - 1.463037, -0.355402 * T + 0.427494, -0.022669 * T + 5.325056, 0.120647 * T - 2.577052, -0.066967 * T + 0.370275,
                d[2], math.acos(cgl), cgl, ts, sun[2])
    cgx = 2 - cg if cg > 1 else cg
    g = math.acos(max(-1.0, cgx))
    t2, ts2, ts3 = T * T, ts * ts, ts ** 3
    zx = (0.001650 * ts3 - 0.003742 * ts2 + 0.002088 * ts) * t2 + (-0.029028 * ts3 + 0.063773 * ts2 - 0.032020 * ts + 0.003948) * T + (0.116936 * ts3 - 0.211960 * ts2 + 0.060523 * ts + 0.258852)
    zy = (0.002759 * ts3 - 0.006105 * ts2 + 0.003162 * ts) * t2 + (-0.042149 * ts3 + 0.089701 * ts2 - 0.041536 * ts + 0.005158) * T + (0.153467 * ts3 - 0.267568 * ts2 + 0.066698 * ts + 0.266881)
    x = zx * _perez(-0.019257 * T - (0.29 - math.sqrt(sun[2]) * 0.09), -0.066513 * T + 0.000818, -0.000417 * T + 0.212479, -0.064097 * T - 0.898875, -0.003251 * T + 0.045178, d[2], g, cgx, ts, sun[2])
    y = zy * _perez(-0.016698 * T - 0.260787, -0.094958 * T + 0.009213, -0.007928 * T + 0.210230, -0.044050 * T - 1.653694, -0.010922 * T + 0.052919, d[2], g, cgx, ts, sun[2])
    X, Y, Z = x / y * L, L, (1 - x - y) / y * L
    return PI * np.array([3.241 * X - 1.537 * Y - 0.499 * Z, -0.969 * X + 1.876 * Y + 0.042 * Z, 0.056 * X - 0.204 * Y + 1.057 * Z])


def _sun_color(sun, T):
    if sun[2] <= 0: return np.zeros(3)
    m = 1.0 / (sun[2] + 0.15 * (93.885 - math.acos(sun[2]) * 180 / PI) ** -1.253)
    wl = np.array([0.610, 0.550, 0.470]); beta = 0.04608 * T - 0.04586
    ta = np.exp(-m * beta * wl ** -1.3); to = np.exp(-m * np.array([12.0, 8.5, 0.9]) * 0.0035); tr = np.exp(-m * 0.008735 * wl ** -4.08)
    return tr * ta * to * np.array([1.0, 0.992, 0.911]) * 127500 / 0.9878


def _smooth(a, b, x):
    t = min(max((x - a) / (b - a), 0.0), 1.0); return t * t * (3 - 2 * t)


def reference_sky(ss, direction):
    d = np.asarray(direction, float)
    hh = ss.horizon_height / 10.0
    def tweak(v):
        v = np.array([v[0], v[2], v[1]]) if ss.y_is_up == 1 else np.array(v, float)
        if hh != 0: v = _norm(v - np.array([0, 0, hh]))
        return v
    d = tweak(d)
    haze = max(2.0, 2.0 + ss.haze)
    sat = ss.saturation
    if sat <= 1.0:
        lh = min(max((haze - 2.0) / 15.0, 0), 1) ** 3
        lsat = sat * (1 - lh) + sat ** 3 * lh
    else:
        lsat = 1.0
    scale = np.array(ss.rgb_unit_conversion[:], float) * ss.multiplier
    if ss.multiplier <= 0: return np.zeros(3)
    down = d[2]; real = d.copy()
    if d[2] < 0.001: d = _norm(np.array([d[0], d[1], 0.001]))
    sun = tweak(_norm(ss.sun_direction[:])); real_sun = sun.copy(); factor = 1.0
    if sun[2] < 0.001:
        if sun[2] < 0:
            lmt = 0.30901699437494742
            factor = 0.0 if sun[2] <= -lmt else ((sun[2] + lmt) / lmt) ** 4
        sun = _norm(np.array([sun[0], sun[1], 0.001]))
    tint = _env_color(sun, d, haze) * min(factor, 1.0) if factor > 0 else np.zeros(3)
    sc = _sun_color(sun, haze if down > 0 else 2.0)
    if ss.sun_disk_intensity > 0 and ss.sun_disk_scale > 0:
        ang = math.acos(min(1.0, max(-1.0, float(real @ real_sun)))); rad = 0.00465 * ss.sun_disk_scale * 10
        if ang < rad:
            ds = gs = 1.0
            if ss.physically_scaled_sun == 1:
                r = 0.00465 * ss.sun_disk_scale * 10
                gi = ss.sun_glow_intensity * (4 * PI - 24 * PI / r ** 2 + 24 * PI * math.sin(r) / r ** 3)
                tgt = ss.sun_disk_intensity * PI
                if gi > 0.5 * tgt: gs = 0.5 * tgt / gi; tgt -= 0.5 * tgt
                else: tgt -= gi
                area = 2 * PI * (1 - math.cos(0.00465 * ss.sun_disk_scale))
                ti = tgt / area
                ds = 0.0 if ti == 0 else ti / (ss.sun_disk_intensity * 100.0)
            sf = (1 - ang / rad) * 10
            tint = tint + sc * ((sf / 10) ** 3 * 2 * ss.sun_glow_intensity * gs + _smooth(8.5, 9.5 + haze / 50, sf) * 100 * ss.sun_disk_intensity * ds)
    out = tint * scale; night = 1.0
    if down <= 0:
        acc = np.zeros(3)
        for u in (0.1, 0.3, 0.5, 0.7, 0.9):
            for v in (0.1, 0.3, 0.5, 0.7, 0.9):
                lx, ly = 2 * u - 1, 2 * v - 1
                if lx == 0 and ly == 0: r = phi = 0.0
                elif lx > -ly:
                    r, phi = (lx, PI / 4 * (1 + ly / lx)) if lx > ly else (ly, PI / 4 * (3 - lx / ly))
                else:
                    r, phi = (-lx, PI / 4 * (5 + ly / lx)) if lx < ly else (-ly, PI / 4 * (7 - lx / ly))
                x, y = r * math.cos(phi), r * math.sin(phi)
                acc += _env_color(sun, np.array([x, y, math.sqrt(max(0.0, 1 - x * x - y * y))]), 2.0)
        downc = np.array(ss.ground_color[:], float) * (acc / 25 + sc * sun[2]) * scale * min(factor, 1.0)
        hb = ss.horizon_blur / 10
        if hb > 0:
            dn = _smooth(0, 1, min(1.0, -down / hb)); out = out * (1 - dn) + downc * dn; night = 1 - dn
        else:
            out = downc; night = 0.0
    inten = out @ np.array([0.2126, 0.7152, 0.0722])
    out = (np.full(3, inten) if lsat <= 0 else out * lsat + inten * (1 - lsat)) * np.array([1 + ss.redblueshift, 1, 1 - ss.redblueshift])
    if night > 0: out = np.maximum(out, np.array(ss.night_color[:], float) * night)
    return out * PI


SKIES = {
    "default": {},
    "hazy_low_sun": {"haze": 4.0, "sun_direction": [0.3, 0.12, -0.9], "redblueshift": 0.2, "saturation": 0.7},
    "below_horizon": {"sun_direction": [0.0, -0.1, 1.0], "horizon_height": 0.3, "horizon_blur": 0.0},
    "z_up_unscaled": {"y_is_up": 0, "sun_direction": [0.2, 0.3, 0.9], "physically_scaled_sun": 0, "sun_disk_scale": 2.0, "multiplier": 0.00002},
}


def _dirs(ss, n=200, seed=1):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    sun = _norm(ss.sun_direction[:])
    near = sun + rng.normal(scale=0.01, size=(40, 3)); near /= np.linalg.norm(near, axis=1, keepdims=True)   # inside the disk + glow
    return np.concatenate([d, near, sun[None]]).astype(np.float32)


@pytest.mark.parametrize("name", list(SKIES))
def test_oracle_sky_matches_float64_model(name):
    from oracle.binding import sun_and_sky_eval
    ss = abi.SunAndSky(in_use=1, **SKIES[name])
    d = _dirs(ss)
    got = sun_and_sky_eval(ss, d).astype(np.float64)
    ref = np.array([reference_sky(ss, v.astype(np.float64)) for v in d])
    scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-9
    err = np.abs(got - ref) / scale
    # float32 + <=4 ulp transcendentals through a chain of exp/acos/pow; the sun-disk edge (smoothstep of a difference of
    # nearly equal angles) is the least well conditioned part
    assert np.quantile(err, 0.95) < 2e-3 and err.max() < 5e-2, (np.quantile(err, 0.95), err.max())
    assert np.isfinite(got).all() and (got >= 0).all()


def test_oracle_sky_frame_uses_the_sky():
    """EnvRadiance / EnvSample / EnvEval switch over (pathtrace.glsl:40-72): a Cornell box lit only by the procedural sky."""
    from oracle.binding import Oracle
    W, H = 48, 32
    sc, _ = make_scene(abi.PROC_SPONZA, 0.02, 1)
    st = host.default_state(W, H, sc, None)
    st.environmentProb = 0.5; st.fireflyClampThreshold = 50.0; st.envMapLuminIntegInv = 0.0
    o = Oracle(0); o.upload_scene(sc.desc(None)); o.resize(W, H)
    sc.updateCamera(W, H); sc.updateCamera(W, H); o.set_camera(sc.getCamera())
    o.render_frame(st, 0)
    dark = o.readback(abi.BUF_DIRECT_RESULT0).view(np.float32).reshape(H, W, 4)[..., :3].copy()
    lid0 = o.readback(abi.BUF_LIGHT_ID0).view(np.uint32)
    o.set_sun_and_sky(abi.SunAndSky(in_use=1))
    o.render_frame(st, 0)
    lit = o.readback(abi.BUF_DIRECT_RESULT0).view(np.float32).reshape(H, W, 4)[..., :3]
    lid1 = o.readback(abi.BUF_LIGHT_ID0).view(np.uint32)
    assert np.isfinite(lit).all() and lit.mean() > dark.mean() + 1e-3
    assert (lid1 == 0xBFFFFFFF).any() and not (lid0 == 0xBFFFFFFF).any()     # reservoirs hold sun samples


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default", "hazy_low_sun", "below_horizon"])
def test_gpu_sky_frames_bit_exact(name):
    from restir_amd.renderer import Renderer
    from oracle.binding import Oracle
    W, H = 160, 96
    sc, _ = make_scene(abi.PROC_SPONZA, 0.02, 1)
    st = host.default_state(W, H, sc, None)
    st.environmentProb = 0.5; st.fireflyClampThreshold = 50.0; st.envMapLuminIntegInv = 0.0
    ss = abi.SunAndSky(in_use=1, **SKIES[name])
    desc = sc.desc(None)
    o = Oracle(0); o.upload_scene(desc); o.resize(W, H); o.set_sun_and_sky(ss)
    r = Renderer().setup(0); r.load_scene(desc); r.update(W, H); r.set_sun_and_sky(ss)
    gpu = RendererBackend(r)
    sc.updateCamera(W, H)
    for f in range(3):
        st.time = 1000 + f; sc.updateCamera(W, H)
        o.set_camera(sc.getCamera()); gpu.set_camera(sc.getCamera())
        o.render_frame(st, f); gpu.render_frame(st, f)
        cmp = compare_buffers(o, gpu, frame_buffers(f))
        bad = {k: v for k, v in cmp.items() if v[0]}
        assert not bad, (name, f, bad)
    img = r.readback(abi.BUF_DIRECT_RESULT0 + 0).view(np.float32)
    assert img.max() > 0.01
