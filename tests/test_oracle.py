"""The CPU oracle against analytic answers, its own brute force, and the invariants of the algorithm."""
import ctypes as C
import numpy as np
import pytest
from helpers import abi, host, make_scene, frame_buffers
from oracle.binding import Oracle, lib


def _rays(n, bbox_lo, bbox_hi, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(bbox_lo, bbox_hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), dtype=np.float32)
    r[:, 0:3], r[:, 3:6], r[:, 6] = o, d, 1e28
    r[:, 7] = rng.integers(0, 2**31, n).astype(np.uint32).view(np.float32)
    return r


@pytest.mark.parametrize("kind,scale", [(abi.PROC_SPONZA, 0.02), (abi.PROC_BISTRO_EXT, 0.004), (abi.PROC_BISTRO_EXT_REAL, 0.004)])
def test_bvh_equals_brute_force(kind, scale):
    sc, _ = make_scene(kind, scale)
    o = Oracle(1)
    o.upload_scene(sc.desc())
    rays = _rays(3000, [-12, 0.2, -5], [12, 8, 5], 3)
    a, b = o.trace_closest(rays), o.trace_closest(rays, brute=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))          # t, triangle id, barycentrics: bit-exact
    hit = a[:, 0] < 1e27
    assert 0.3 < hit.mean() <= 1.0
    rays[:, 6] = np.where(hit, a[:, 0] * 1.001, 1.0)
    assert np.array_equal(o.trace_any(rays) == 1, hit)                    # any-hit agrees with closest-hit
    rays[:, 6] = np.where(hit, a[:, 0] * 0.999, 1.0)
    assert (o.trace_any(rays)[hit] == 0).all()                            # nothing in front of the closest hit


def test_cornell_analytic_gbuffer():
    W = H = 64
    sc, _ = make_scene(abi.PROC_CORNELL)
    st = host.default_state(W, H, sc, None); st.environmentProb = 0.0
    o = Oracle(2); o.upload_scene(sc.desc()); o.resize(W, H)
    sc.updateCamera(W, H); sc.updateCamera(W, H)
    o.set_camera(sc.getCamera())
    o.run_stage(st, 0, abi.STAGE_DIRECT)
    g = o.readback(abi.BUF_GBUFFER0).view(np.uint32).reshape(H, W, 4)
    depth = g[..., 0].view(np.float32)
    # eye (0,1,3.4) looks down -z at the back wall z=-1; the top row of pixels sees the ceiling, none misses
    assert (depth < 1e27).all()
    c = depth[H // 2 - 4:H // 2 + 4, 2:6]                                  # left part of the centre rows: back wall or red wall
    assert np.all((c > 2.3) & (c < 5.0))
    n = g[H - 2, 6, 1]                                                     # bottom left: floor, normal +y
    out = np.zeros(3, dtype=np.float32); lib().orc_decompress_unit_vec(int(n), out.ctypes.data)
    assert np.allclose(out, [0, 1, 0], atol=1e-3)
    assert (g[..., 3] >> 24 != 0xff).all()                                 # every pixel carries a material hash
    mot = o.readback(abi.BUF_MOTION).view(np.int16).reshape(H, W, 2)
    yy, xx = np.mgrid[0:H, 0:W]
    assert np.array_equal(mot[..., 0], xx) and np.array_equal(mot[..., 1], yy)  # static camera reprojects onto itself


def test_frame_is_deterministic_and_thread_independent():
    W, H = 96, 64
    sc, env = make_scene(abi.PROC_HELMET, 0.02, env_size=(64, 32))
    st = host.default_state(W, H, sc, env)
    outs = []
    for threads in (1, 5):
        o = Oracle(threads); o.upload_scene(sc.desc(env)); o.resize(W, H)
        s2 = host.Scene().makeProcedural(abi.PROC_HELMET, 0.02, 1)
        s2.updateCamera(W, H)
        for f in range(2):
            st.time = 77 + f; s2.updateCamera(W, H); o.set_camera(s2.getCamera()); o.render_frame(st, f)
        outs.append([o.readback(b) for b in frame_buffers(1)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_pinned_workers_leave_the_callers_affinity_alone():
    """Round 6: the spawning thread used to pin worker t through its handle — and glibc's pthread_setaffinity_np on a worker that had ALREADY EXITED (more threads than
    rows, short stages) is sched_setaffinity(0, ...): it pinned the CALLER to one CPU, and every later worker with it (bench.py's 128- / 256-thread points ran on one CPU,
    profiles/r06_cpu_baseline.txt).  Workers pin themselves now: a pinned run with far more threads than rows must leave the calling thread's mask as it was."""
    import os
    if not hasattr(os, "sched_getaffinity") or len(os.sched_getaffinity(0)) < 2:
        pytest.skip("needs an affinity mask of at least two CPUs")
    before = os.sched_getaffinity(0)
    W, H = 64, 32
    sc, env = make_scene(abi.PROC_HELMET, 0.02, env_size=(64, 32))
    st = host.default_state(W, H, sc, env)
    o = Oracle(1); o.upload_scene(sc.desc(env)); o.resize(W, H)
    sc.updateCamera(W, H); o.set_camera(sc.getCamera())
    try:
        o.set_threads(64, pin=True)              # 64 workers for 32 (16 half-resolution) rows: most of them find no row and exit at once
        for f in range(3):
            st.time = 5 + f; o.render_frame(st, f)
            assert os.sched_getaffinity(0) == before, f"frame {f}: the caller's affinity mask changed from {len(before)} to {len(os.sched_getaffinity(0))} CPUs"
        for k in range(300):                     # the shortest stage there is, again and again: a worker is done before the spawning loop has reached the next one
            o.run_stage(st, 2, abi.STAGE_COMPOSE, 0, 0, H)
            assert os.sched_getaffinity(0) == before, f"compose pass {k}: the caller's affinity mask changed from {len(before)} to {len(os.sched_getaffinity(0))} CPUs"
    finally:
        os.sched_setaffinity(0, before)


def test_cpu_budget_follows_the_override_and_the_mask():
    """include/rt_cpus.h (the default thread count of the oracle, the builder and the loaders): RESTIR_CPUS overrides; otherwise never more than the affinity mask"""
    import os, subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write('#include "rt_cpus.h"\n#include <cstdio>\nint main() { std::printf("%d %d\\n", rt_cpu_budget(), rt_cpu_quota()); return 0; }\n')
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(root, "include"), src, "-o", exe])
        env = {k: v for k, v in os.environ.items() if k != "RESTIR_CPUS"}
        budget, quota = map(int, subprocess.check_output([exe], env=env).split())
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
        assert 1 <= budget <= ncpu and (quota == 0 or budget <= quota)
        assert int(subprocess.check_output([exe], env=dict(env, RESTIR_CPUS="7")).split()[0]) == 7
        if ncpu >= 2 and hasattr(os, "sched_setaffinity"):
            one = subprocess.check_output(["taskset", "-c", str(sorted(os.sched_getaffinity(0))[0]), exe], env=env).split() if subprocess.call(["which", "taskset"], stdout=subprocess.DEVNULL) == 0 else None
            if one is not None:
                assert int(one[0]) == 1


def test_reservoir_invariants_over_frames():
    W = H = 48
    sc, _ = make_scene(abi.PROC_CORNELL)
    st = host.default_state(W, H, sc, None); st.environmentProb = 0.0; st.fireflyClampThreshold = 100.0
    o = Oracle(4); o.upload_scene(sc.desc()); o.resize(W, H)
    sc.updateCamera(W, H)
    dt = np.dtype([("Li", "<f4", 3), ("wi", "<f4", 3), ("dist", "<f4"), ("num", "<u4"), ("weight", "<f4")])
    prev_num = None
    for f in range(90):
        st.time = 5 + f; sc.updateCamera(W, H); o.set_camera(sc.getCamera()); o.run_stage(st, f, abi.STAGE_DIRECT)
        resv = o.readback(abi.BUF_DIRECT_RESV0 + (f & 1)).view(dt)
        assert (resv["num"] <= st.RISSampleNum * st.reservoirClamp).all()        # resvClamp
        assert np.isfinite(resv["weight"]).all() and (resv["weight"] >= 0).all()  # resvCheckValidity
        if prev_num is not None and f < 60:
            lit = resv["num"] > 0
            assert (resv["num"][lit] >= np.minimum(prev_num[lit], 4)).all()
        prev_num = resv["num"].copy()
    assert prev_num.max() == st.RISSampleNum * st.reservoirClamp                   # history saturates at M*clamp
    lid = o.readback(abi.BUF_LIGHT_ID0 + (89 & 1)).view(np.uint32)
    kinds = set((lid[lid != 0xffffffff] >> 28).tolist())
    assert kinds <= {0x4}                                                          # only triangle lights in a Cornell box


def _bsdf(fn, m, n, wo, x):
    m, n, wo, x = (np.ascontiguousarray(v, dtype=np.float32) for v in (m, n, wo, x))
    out = np.zeros(8, dtype=np.float32)
    fn(m.ctypes.data, n.ctypes.data, wo.ctypes.data, x.ctypes.data, out.ctypes.data)
    return out


def test_bsdf_reciprocity_pdf_and_energy():
    rng = np.random.default_rng(5)
    n = np.array([0, 0, 1], dtype=np.float32)
    for trial in range(20):
        m = [*rng.uniform(0.2, 1, 3), float(rng.integers(0, 2)), rng.uniform(0.15, 1.0)]
        wo = rng.normal(size=3); wo[2] = abs(wo[2]) + 0.2; wo /= np.linalg.norm(wo)
        acc, cnt = np.zeros(3), 0
        for k in range(600):
            r = rng.uniform(0, 1, 3)
            s = _bsdf(lib().orc_bsdf_sample, m, n, wo, r)
            wi, pdf = s[0:3], s[3]
            if pdf <= 1e-8:
                cnt += 1; continue
            e = _bsdf(lib().orc_bsdf_eval, m, n, wo, wi)
            assert np.allclose(e[0:3], s[4:7], rtol=1e-5, atol=1e-7)                # Sample's f == Eval's f
            assert abs(e[3] - pdf) <= 1e-5 * max(1.0, pdf)                          # Sample's pdf == Pdf(dir)
            back = _bsdf(lib().orc_bsdf_eval, m, n, wi, wo)
            assert np.allclose(back[0:3], e[0:3], rtol=2e-3, atol=1e-6)             # reciprocity
            acc += e[0:3] * wi[2] / pdf; cnt += 1
        assert (acc / cnt < 1.6).all()   # the reference BSDF (alpha = roughness, Schlick-G) is not exactly energy conserving; it stays bounded


def test_spatial_reuse_modes():
    """ReSTIRState eSpatial / eSpatiotemporal (direct_stage.comp:224-255): the reuse step runs after every pixel cached its
    reservoir; it changes the shaded image but not the saved reservoirs, and the cached reservoirs are the pre-clamp ones."""
    from oracle.binding import Oracle
    W, H = 64, 48
    sc, env = make_scene(abi.PROC_SPONZA, 0.01, 1, (64, 32))
    desc = sc.desc(env)
    out = {}
    for mode in (abi.RESTIR_TEMPORAL, abi.RESTIR_SPATIOTEMPORAL, abi.RESTIR_RIS, abi.RESTIR_SPATIAL):
        st = host.default_state(W, H, sc, env); st.ReSTIRState = mode
        o = Oracle(0); o.upload_scene(desc); o.resize(W, H)
        sc.setCamera(*sc.cameraPose()); sc.updateCamera(W, H)
        for f in range(2):
            st.time = 500 + f; sc.updateCamera(W, H); o.set_camera(sc.getCamera())
            o.run_stage(st, f, abi.STAGE_DIRECT)
        out[mode] = (o.readback(abi.BUF_DIRECT_RESULT0 + 1).view(np.float32).reshape(H, W, 4)[..., :3].copy(),
                     o.readback(abi.BUF_DIRECT_RESV0 + 1).copy(), o.readback(abi.BUF_DIRECT_RESV_TEMP).copy())
    for a, b in ((abi.RESTIR_TEMPORAL, abi.RESTIR_SPATIOTEMPORAL), (abi.RESTIR_RIS, abi.RESTIR_SPATIAL)):
        img_a, resv_a, _ = out[a]; img_b, resv_b, temp_b = out[b]
        assert np.isfinite(img_b).all()
        assert np.array_equal(resv_a, resv_b)                      # saveNewReservoir happens before the spatial step
        assert not np.array_equal(img_a, img_b)                    # the shaded sample went through ten neighbour merges
        t = temp_b.view(np.uint8).reshape(-1, 36); s = resv_b.view(np.uint8).reshape(-1, 36)
        num_t = t[:, 28:32].copy().view(np.uint32).ravel(); num_s = s[:, 28:32].copy().view(np.uint32).ravel()
        assert (num_t >= num_s).all() and (num_t > 0).any()        # cached = un-clamped M


def test_direct_estimators_agree_in_expectation():
    """Plain next-event estimation (ReSTIRState eNone, pathtrace.glsl:205-220), RIS with M = 4 and M = 16 candidates and RIS +
    temporal reuse (direct_stage.comp:186-262) are estimators of the same integral: their per-pixel means over many seeds must
    agree.  Catches a wrong weight / normalisation in the restated estimator (`LiBsdf / lum(LiBsdf) * weight / num`), which no
    known-answer vector of a helper can see.  Measured with 300 frames: ratios 1.003-1.005 of the image sums."""
    W = H = 48
    K = 160
    sc, _ = make_scene(abi.PROC_CORNELL)
    means = {}
    for name, mode, M in (("nee", abi.RESTIR_NONE, 4), ("ris4", abi.RESTIR_RIS, 4), ("ris16", abi.RESTIR_RIS, 16), ("temporal", abi.RESTIR_TEMPORAL, 4)):
        st = host.default_state(W, H, sc, None)
        st.environmentProb = 0.0; st.fireflyClampThreshold = 1e6; st.ReSTIRState = mode; st.RISSampleNum = M; st.denoise = 0
        o = Oracle(0); o.upload_scene(sc.desc()); o.resize(W, H)
        sc.updateCamera(W, H); sc.updateCamera(W, H)
        acc = np.zeros((H, W, 3))
        for f in range(K):
            st.time = 9000 + f; sc.updateCamera(W, H); o.set_camera(sc.getCamera()); o.run_stage(st, f, abi.STAGE_DIRECT)
            y = o.readback(abi.BUF_DIRECT_RESULT0 + (f & 1)).view(np.float32).reshape(H, W, 4)[..., :3].astype(np.float64)
            acc += y / (1.0 - np.minimum(y, 0.999999))            # undo HDRToLDR (the image stays LDR-encoded with denoise == 0)
        means[name] = acc / K
    lum = lambda a: a @ np.array([0.2126, 0.7152, 0.0722])  # noqa: E731
    ref = means["nee"]
    mask = (lum(ref) > 0.02) & (lum(ref) < 5.0)               # lit surfaces, not the emitter itself
    assert mask.sum() > 1000
    for name in ("ris4", "ris16", "temporal"):
        ratio_rgb = means[name][mask].sum(0) / ref[mask].sum(0)
        assert np.all(np.abs(ratio_rgb - 1.0) < 0.02), (name, ratio_rgb)
        per_pixel = lum(means[name])[mask] / lum(ref)[mask]
        assert abs(per_pixel.mean() - 1.0) < 0.02 and per_pixel.std() < 0.25, (name, per_pixel.mean(), per_pixel.std())


def test_indirect_temporal_reuse_keeps_the_expectation():
    """ReSTIR GI with temporal reuse (indirect_stage.comp:228-268: history reservoir, `resvUpdate` of the new path, clamp 2x,
    `bigW = weight / (p_hat * num)`) against the same stage without history (ReSTIRState eRIS: one path per frame): same
    expectation.  1000 seeds at 64x64 (deterministic): image sums within 5 % (measured 0.97-0.98; 2000 seeds: 0.98-1.00)."""
    W = H = 64
    K = 1000
    sc, _ = make_scene(abi.PROC_CORNELL)
    means = {}
    for name, mode in (("single", abi.RESTIR_RIS), ("temporal", abi.RESTIR_TEMPORAL)):
        st = host.default_state(W, H, sc, None)
        st.environmentProb = 0.0; st.fireflyClampThreshold = 1e6; st.ReSTIRState = mode; st.denoise = 0
        o = Oracle(0); o.upload_scene(sc.desc()); o.resize(W, H)
        sc.updateCamera(W, H); sc.updateCamera(W, H)
        acc = np.zeros((H // 2, W // 2, 3))
        for f in range(K):
            st.time = 9000 + f; sc.updateCamera(W, H); o.set_camera(sc.getCamera())
            o.run_stage(st, f, abi.STAGE_DIRECT); o.run_stage(st, f, abi.STAGE_INDIRECT)
            y = o.readback(abi.BUF_DENOISE_IND_A).view(np.float32).reshape(H, W, 4)[:H // 2, :W // 2, :3].astype(np.float64)
            acc += y / (1.0 - np.minimum(y, 0.999999))            # undo HDRToLDR
        means[name] = acc / K
    ratio = means["temporal"].sum((0, 1)) / means["single"].sum((0, 1))
    assert means["single"].mean() > 0.1
    assert np.all(np.abs(ratio - 1.0) < 0.05), ratio


def test_direct_light_estimator_matches_area_quadrature():
    """The whole light-sampling chain (uniform id -> alias table -> uniform barycentrics -> `Li = emission/area`, pdf =
    table pdf * d^2 / (area * |cos|), times trigSampProb; pathtrace.glsl:103-139, 161-183, 205-220) must integrate to
      E[direct] = sum_tri (emission / area_tri) * integral_tri f(wo, wi) cos+ |cos_l| / d^2 V dA
    whatever the selection probabilities are.  The mean of 300 NEE frames is compared with a midpoint quadrature of that
    integral (BSDF through orc_bsdf_eval, visibility through orc_trace_any) on a grid of Cornell-box pixels."""
    W = H = 48
    K = 300
    sc, _ = make_scene(abi.PROC_CORNELL)
    desc = sc.desc()
    st = host.default_state(W, H, sc, None)
    st.environmentProb = 0.0; st.fireflyClampThreshold = 1e6; st.ReSTIRState = abi.RESTIR_NONE; st.denoise = 0
    o = Oracle(0); o.upload_scene(desc); o.resize(W, H)
    sc.updateCamera(W, H); sc.updateCamera(W, H)
    acc = np.zeros((H, W, 3))
    for f in range(K):
        st.time = 4000 + f; sc.updateCamera(W, H); cam = sc.getCamera(); o.set_camera(cam); o.run_stage(st, f, abi.STAGE_DIRECT)
        y = o.readback(abi.BUF_DIRECT_RESULT0 + (f & 1)).view(np.float32).reshape(H, W, 4)[..., :3].astype(np.float64)
        acc += y / (1.0 - np.minimum(y, 0.999999))
    mean = acc / K
    g = o.readback(abi.BUF_GBUFFER0 + ((K - 1) & 1)).view(np.uint32).reshape(H, W, 4)

    nl = desc.lightInfo.trigLightSize
    assert nl >= 2
    lights = np.frombuffer(C.string_at(desc.trigLights, nl * 96), dtype=np.uint8).reshape(nl, 96)
    mats = np.frombuffer(C.string_at(desc.materials, desc.numMaterials * 80), dtype=np.uint8).reshape(-1, 80)
    vi = np.array(list(cam.viewInverse.m), dtype=np.float64).reshape(4, 4).T
    pi = np.array(list(cam.projInverse.m), dtype=np.float64).reshape(4, 4).T
    n_q = 20                                                          # midpoints of an n_q x n_q barycentric grid, both halves
    a, b = (np.mgrid[0:n_q, 0:n_q] + 0.5) / n_q
    lower = (a + b) < 1.0
    bary = np.concatenate([np.stack([a[lower], b[lower]], -1), np.stack([1 - a[~lower], 1 - b[~lower]], -1)])  # folds the square onto the triangle
    lum = lambda c: c @ np.array([0.2126, 0.7152, 0.0722])  # noqa: E731
    est, quad = [], []
    out3, out8 = np.zeros(3, np.float32), np.zeros(8, np.float32)
    for py in range(3, H, 6):
        for px in range(3, W, 6):
            tex = g[py, px]
            if (tex[3] & 0xFF000000) == 0xFF000000 or lum(mean[py, px]) > 5.0:
                continue
            u, v = (px + 0.5) / W * 2 - 1, (py + 0.5) / H * 2 - 1
            t = pi @ np.array([u, v, 1.0, 1.0]); d = vi[:3, :3] @ (t[:3] / np.linalg.norm(t[:3]))
            pos = vi[:3, 3] + d * float(tex[0:1].view(np.float32)[0])
            lib().orc_decompress_unit_vec(int(tex[1]), out3.ctypes.data)
            n = out3.astype(np.float64).copy()
            if np.dot(n, -d) < 0: n = -n                               # ffnormal
            zb = int(tex[2])
            m = np.array([1, 1, 1, (zb & 0xff) / 255.0, ((zb >> 8) & 0xff) / 255.0], np.float32)   # albedo forced to 1 (direct_stage.comp:179)
            total = np.zeros(3)
            for L in lights:
                v0, v1, v2 = (L[8 + 12 * k: 20 + 12 * k].view(np.float32).astype(np.float64) for k in range(3))
                emission = mats[int(L[0:4].view(np.uint32)[0])][36:48].view(np.float32).astype(np.float64)
                nl_vec = np.cross(v1 - v0, v2 - v0); area = 0.5 * np.linalg.norm(nl_vec); nl_vec /= 2 * area
                ys = v0 + bary[:, :1] * (v1 - v0) + bary[:, 1:] * (v2 - v0)
                dirs = ys - pos; dist = np.linalg.norm(dirs, axis=1); wi = dirs / dist[:, None]
                rays = np.zeros((len(ys), 8), np.float32)
                rays[:, 0:3] = pos + n * 1e-3; rays[:, 3:6] = wi; rays[:, 6] = dist * 0.999
                vis = o.trace_any(rays) == 0
                cosl = np.abs(wi @ nl_vec)
                for k in np.nonzero(vis)[0]:
                    w32 = wi[k].astype(np.float32)
                    lib().orc_bsdf_eval(m.ctypes.data, n.astype(np.float32).ctypes.data, (-d).astype(np.float32).ctypes.data, w32.ctypes.data, out8.ctypes.data)
                    total += (emission / area) * out8[0:3] * max(float(np.dot(n, wi[k])), 0.0) * cosl[k] / dist[k] ** 2 * (area / len(ys))
            est.append(mean[py, px]); quad.append(total)
    est, quad = np.array(est), np.array(quad)
    assert len(est) >= 40 and lum(quad).sum() > 1.0
    ratio = est.sum(0) / quad.sum(0)
    assert np.all(np.abs(ratio - 1.0) < 0.03), ratio                  # measured: within 1 %
    lit = lum(quad) > 0.2 * lum(quad).mean()
    per_pixel = lum(est)[lit] / lum(quad)[lit]
    assert abs(np.median(per_pixel) - 1.0) < 0.05, np.median(per_pixel)


def test_sliver_triangle_cannot_hit_outside_its_box():
    """The ray behind tests/test_gpu_fuzz.py::test_regression_sliver_triangle: a shadow ray of the interior scene that ends next to
    an emissive sliver (two vertices an ulp apart, area 7e-11).  Raw Moller-Trumbore reports a hit on it at t = 4 — metres from the
    triangle.  With the hit-point-in-padded-box rule (orc_scene.cpp / csrc/traverse.h intersectTri) the ray is unoccluded up to its
    tmax, and the oracle's BVH and its brute-force tracer agree on the closest hit (the light's neighbour, just beyond tmax)."""
    import test_gpu_fuzz as F
    rng = np.random.default_rng(302)
    for _ in range(384):
        F._skip_case(rng)
    kind, scale = F.KINDS[rng.integers(len(F.KINDS))]
    rng.integers(33, 260); rng.integers(17, 150); rng.integers(3)
    sc, env = make_scene(kind, scale, int(rng.integers(1, 1000)), (64, 32))
    assert kind == abi.PROC_BISTRO_INT
    o = Oracle(1); o.upload_scene(sc.desc(env)); o.resize(16, 16)
    bits = lambda *h: np.array([int(x, 16) for x in h], dtype=np.uint32).view(np.float32)  # noqa: E731
    ray = np.zeros((1, 8), np.float32)
    ray[0, 0:3] = bits("c0882ef8", "40644ccd", "40bfff00"); ray[0, 3:6] = bits("3f4a7797", "bc115996", "bf1ca54f")
    ray[0, 6] = bits("411ce318")[0]; ray[0, 7] = bits("0d490c4e")[0]
    assert o.trace_any(ray)[0] == 0
    far = ray.copy(); far[0, 6] = 1e28
    a, b = o.trace_closest(far), o.trace_closest(far, brute=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert a[0, 0] > ray[0, 6] and abs(a[0, 0] - 9.8057) < 1e-3


def test_config1_helmet_class_full_size_cpu_plumbing():
    """BASELINE config 1 at its stated size: DamagedHelmet-class mesh (~70 k triangles, base colour / metal-rough / normal / emissive textures), 256x256, one
    primary ray + direct light per pixel, CPU only — the loader -> scene arrays -> BVH -> trace -> shade plumbing of the oracle (no GPU involved)."""
    W = H = 256
    sc, env = make_scene(abi.PROC_HELMET, 1.0, 1, (256, 128))
    tris = sc.getStat()["instancedTriangles"]
    assert 5e4 < tris < 1e5, tris
    st = host.default_state(W, H, sc, env)
    o = Oracle(0); o.upload_scene(sc.desc(env)); o.resize(W, H)
    sc.updateCamera(W, H); sc.updateCamera(W, H)
    o.set_camera(sc.getCamera())
    o.reset_counters()
    o.run_stage(st, 0, abi.STAGE_DIRECT)
    c = o.counters()
    assert c.closestHitRays == W * H and c.anyHitRays > 0
    g = o.readback(abi.BUF_GBUFFER0).view(np.uint32).reshape(H, W, 4)
    hit = g[..., 0].view(np.float32) < 1e27
    assert 0.15 < hit.mean() < 0.95                                         # the helmet fills part of the view, the rest sees the environment
    img = o.readback(abi.BUF_DIRECT_RESULT0).view(np.float32).reshape(H, W, 4)
    assert np.isfinite(img).all() and img[hit][:, :3].max() > 0.05 and img[~hit][:, :3].max() > 0.0
    assert len(np.unique(g[hit][:, 3] & 0xffffff)) > 50                    # textured albedo, not one flat colour
    # the oracle's BVH against its own brute force on this mesh (the GPU-free leg of the parity chain for config 1)
    rays = _rays(1500, [-1.5, -1.5, -1.5], [1.5, 1.5, 1.5], 9)
    a, b = o.trace_closest(rays), o.trace_closest(rays, brute=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # a second context reproduces the frame bit for bit
    o2 = Oracle(3); o2.upload_scene(sc.desc(env)); o2.resize(W, H); o2.set_camera(sc.getCamera()); o2.run_stage(st, 0, abi.STAGE_DIRECT)
    assert np.array_equal(o2.readback(abi.BUF_DIRECT_RESULT0), o.readback(abi.BUF_DIRECT_RESULT0))
