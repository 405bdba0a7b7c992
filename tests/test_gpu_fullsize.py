"""BASELINE.json full sizes on the GPU: 1920x1080 on the 2.8 M-triangle Bistro-Exterior-class scene.
The oracle cannot render full frames in seconds, so parity is checked on a row band (bit-exact) and the rest through
size-independent properties: run-to-run determinism, tiled == untiled (3 emulated ranks), ray-count bound."""
import threading
import numpy as np
import pytest
from helpers import abi, host, make_scene, frame_buffers

pytestmark = pytest.mark.gpu
W, H = 1920, 1080


@pytest.fixture(scope="module")
def bistro():
    import torch
    torch.cuda.init(); torch.zeros(1, device="cuda")   # initialise torch's HIP context on the main thread before any worker thread needs it
    sc, env = make_scene(abi.PROC_BISTRO_EXT, 1.0, 1, (2048, 1024))
    st = host.default_state(W, H, sc, env)
    sc.updateCamera(W, H); sc.updateCamera(W, H)
    return sc, env, st, sc.getCamera()


def _renderer(sc, env):
    from restir_amd.renderer import Renderer
    r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
    return r


def test_band_parity_vs_oracle_and_determinism(bistro):
    from oracle.binding import Oracle
    sc, env, st, cam = bistro
    r = _renderer(sc, env)
    assert 2.6e6 < r.accel_stats()["triangles"] < 3.0e6
    st.time = 4242
    r.set_camera(cam); r.run(st, 0)
    first = {b: r.readback(b) for b in frame_buffers(0)}
    # determinism: a second context renders the identical frame
    r2 = _renderer(sc, env); r2.set_camera(cam); r2.run(st, 0)
    for b, data in first.items():
        assert np.array_equal(r2.readback(b), data), abi.BUFFER_NAMES[b]
    # oracle on rows 560..576 (street level: props, trees, lamps): direct + indirect stage words must match bit for bit
    o = Oracle(0); o.upload_scene(sc.desc(env)); o.resize(W, H); o.set_camera(cam)
    y0, y1 = 560, 576
    o.run_stage(st, 0, abi.STAGE_DIRECT, 0, y0, y1)
    o.run_stage(st, 0, abi.STAGE_INDIRECT, 0, y0 // 2, y1 // 2)
    for buf, elem, half in [(abi.BUF_GBUFFER0, 16, False), (abi.BUF_MOTION, 4, False), (abi.BUF_DIRECT_RESV0, 36, False), (abi.BUF_LIGHT_ID0, 4, False),
                            (abi.BUF_INDIRECT_RESV0, 76, True)]:
        w, a, b = (W // 2, y0 // 2, y1 // 2) if half else (W, y0, y1)
        got = first[buf].reshape(-1, w * elem)[a:b]
        ref = o.readback(buf).reshape(-1, w * elem)[a:b]
        assert np.array_equal(got, ref), abi.BUFFER_NAMES[buf]
    # the half-res noisy indirect image lives in the top-left quarter of a full-pitch image
    ind_g = r.readback(abi.BUF_DENOISE_IND_A)  # overwritten by later denoise levels on the GPU: compare the oracle's own stage output instead
    del ind_g
    # ray budget of SURVEY.md §8d: <= 2N + 3.25 Nh
    r.set_counting(True); r.run(st, 1); c = r.counters(); r.set_counting(False)
    assert 0 < c.closestHitRays + c.anyHitRays <= 2 * W * H + 3.25 * (W // 2) * (H // 2)


def test_frames_in_flight_equal_serial_frames(bistro, monkeypatch):
    """rt_render_frame keeps consecutive frames in flight (direct(f+1) beside indirect(f) and the filters of f, rotated
    G-buffer / motion buffers).  Twelve back-to-back frames with a moving camera must leave exactly the buffers that the same
    frames leave when every stage runs alone on one stream (RESTIR_OVERLAP=0) — with two frames in flight (the default) and with three
    (RESTIR_OVERLAP=3: four G-buffers, three motion buffers and three direct images in rotation; the first three frames are the serial probe frames,
    so nine frames run through the rotation)."""
    sc, env, st, cam0 = bistro
    eye, center, up, fov = sc.cameraPose()
    cams = []
    for f in range(12):
        sc.setCamera(eye + np.array([0.05 * f, 0.01 * f, -0.04 * f], dtype=np.float32), center, up, fov)
        sc.updateCamera(W, H); cams.append(sc.getCamera())
    sc.setCamera(eye, center, up, fov); sc.updateCamera(W, H); sc.updateCamera(W, H)
    out = {}
    for mode in ("2", "3", "0"):
        monkeypatch.setenv("RESTIR_OVERLAP", mode)
        r = _renderer(sc, env)
        for f, cam in enumerate(cams):
            st.time = 7000 + f
            r.set_camera(cam); r.run(st, f)            # no readback / sync between frames
        out[mode] = {b: r.readback(b) for b in frame_buffers(11) + [abi.BUF_GBUFFER0, abi.BUF_GBUFFER1, abi.BUF_MOTION, abi.BUF_DIRECT_RESV0, abi.BUF_INDIRECT_RESV0,
                                                                     abi.BUF_DIRECT_RESULT0, abi.BUF_INDIRECT_RESULT0]}
        r.destroy()
    for mode in ("2", "3"):
        for b in out["0"]:
            assert np.array_equal(out[mode][b], out["0"][b]), (mode, abi.BUFFER_NAMES[b])


class ThreadComm:
    """In-process stand-in for torch.distributed on ONE GPU: ranks are threads, collectives are device-to-device copies
    between the ranks' own contexts (every rank has full-size private buffers, as on a real node)."""
    def __init__(self, rank, world, shared):
        from restir_amd import tiled
        self.rank, self.world, self.s = rank, world, shared
        self.rx_bytes = dict.fromkeys(tiled.HALO_KINDS, 0)
    def _sync(self):
        import torch
        self.s["renderers"][self.rank].sync(); torch.cuda.synchronize(); self.s["barrier"].wait()
    nccl = False
    def _copy_items(self, items, need_of):
        """the rows every item needs from other ranks' buffers (all ranks call with the same item list => the barriers line up); strided / partial-width
        items go through the same views the RCCL transport packs and unpacks"""
        import torch
        from restir_amd import tiled
        self.s["slot"][self.rank] = [it.tensor for it in items]
        self._sync()
        for i, it in enumerate(items):
            for (lo, hi) in need_of(it):
                for p in range(self.world):
                    if p == self.rank:
                        continue
                    a, b = max(lo, it.part[p]), min(hi, it.part[p + 1], it.limit)
                    dst = tiled.halo_view(it, a, b) if b > a else None
                    if dst is not None:
                        dst.copy_(tiled.halo_view(it, a, b, self.s["slot"][p][i]))
        torch.cuda.synchronize(); self.s["barrier"].wait()
    def all_gather_rows(self, tensor, pitch, part, async_op=False, kind="fallback"):
        from restir_amd import tiled
        self._copy_items([tiled.Halo(tensor, pitch, part, part[-1], part[-1])], lambda it: [(0, part[-1])])
        return None
    def halo_exchange(self, items, async_op=False, kind="filter"):
        self._copy_items([it for it in items], lambda it: it.need(self.rank) if it.halo > it.inner else [])
        return None
    def gather_rows_to(self, tensor, pitch, part, dst=0, async_op=False): return self.all_gather_rows(tensor, pitch, part)
    def all_gather_floats(self, values):
        self.s.setdefault("floats", [None] * self.world)[self.rank] = list(values)
        self.s["barrier"].wait()
        out = [list(v) for v in self.s["floats"]]
        self.s["barrier"].wait()
        return out
    def any_flag(self, flag):
        self.s["flags"][self.rank] = bool(flag)
        self.s["barrier"].wait()
        r = any(self.s["flags"])
        self.s["barrier"].wait()
        return r
    def wait(self, work): pass
    def barrier(self): self.s["barrier"].wait()


@pytest.mark.parametrize("mode", ["serial", "serial-fallbacks", "frames-in-flight", "frames-in-flight-fallbacks", "frames-in-flight-uneven", "frames-in-flight-rebalance", "frames-in-flight-diffuse", "frames-in-flight-spatial", "serial-spatial"])
def test_tiled_three_ranks_equals_untiled(bistro, mode, monkeypatch):
    import torch
    from restir_amd import tiled
    sc, env, st, cam = bistro
    import copy
    st = copy.copy(st)
    if mode.endswith("spatial"): st.ReSTIRState = abi.RESTIR_SPATIOTEMPORAL   # direct stage in two halves around an exchange of the cache's boundary rows
    world, frames = 3, (2 if mode == "serial" else 4)
    Frame = tiled.TiledFrame if mode.startswith("serial") else tiled.PipelinedTiledFrame
    part0 = [0, 496, 560, H] if mode.endswith("uneven") else None    # a 64-row band at the horizon: every halo spans more than one rank
    parts = [None] * world
    if mode.endswith("fallbacks"):
        monkeypatch.setattr(tiled, "HIST_HALO", 0)   # every cross-band reprojection misses: both exact fallbacks (incl. the un-rotate / re-rotate of the G-buffers) run
    ref = _renderer(sc, env)
    rs = [_renderer(sc, env) for _ in range(world)]
    shared = {"renderers": rs, "barrier": threading.Barrier(world), "slot": [None] * world, "flags": [False] * world}
    cams = []
    s2, _ = make_scene(abi.PROC_BISTRO_EXT, 1.0, 1, None)
    eye, center, up, fov = s2.cameraPose()
    s2.updateCamera(W, H)
    for f in range(frames):
        s2.setCamera(eye + np.array([0.3 * f, 0.05 * f, 0], dtype=np.float32), center, up, fov); s2.updateCamera(W, H); cams.append(s2.getCamera())
    errors = []
    fallbacks = [0] * world

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            fr = Frame(tiled.RendererTensors(rs[rank]), ThreadComm(rank, world, shared), W, H, part0)
            import copy
            st_r = copy.copy(st)
            for f in range(frames):
                st_r.time = 500 + f; rs[rank].set_camera(cams[f]); fr.render_frame(st_r, f)
                if mode.endswith("rebalance") and f == 1:      # cost feedback in the middle of the sequence: the street rows cost 5x the sky rows
                    fr.rebalance(sum(5.0 if y >= 500 else 1.0 for y in range(fr.y0, fr.y1)), smoothing=1.0, max_move=8)
                if mode.endswith("diffuse") and f >= 1:        # the diffusion phase of the balancer: a stripe per frame towards the slower neighbour
                    fr.diffuse(sum(5.0 if y >= 500 else 1.0 for y in range(fr.y0, fr.y1)))
            fr.finish(); rs[rank].sync()
            parts[rank] = list(fr.part)
            fallbacks[rank] = fr.history_fallbacks
        except Exception as e:  # pragma: no cover
            errors.append(e); shared["barrier"].abort()

    th = [threading.Thread(target=rank_main, args=(i,)) for i in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errors, errors
    # (with the adaptive history halo — 16 rows until a lookup leaves band + halo — this camera path costs the plain modes one or two exact fallbacks at the start)
    assert fallbacks[0] > 0 if mode.endswith("fallbacks") else fallbacks[0] <= 2
    for f in range(frames):
        st.time = 500 + f; ref.set_camera(cams[f]); ref.run(st, f)
    cur = (frames - 1) & 1
    for b in [abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_INDIRECT_RESULT0 + cur, abi.BUF_DIRECT_RESULT0 + (cur ^ 1), abi.BUF_INDIRECT_RESULT0 + (cur ^ 1)]:   # gathered on rank 0: last frame and the one before
        got, want = rs[0].readback(b).reshape(H, -1), ref.readback(b).reshape(H, -1)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, (abi.BUFFER_NAMES[b], "rows", int(bad.min()), int(bad.max()), int(bad.size))
    part = parts[0]; parth = tiled.half_partition(part, H)
    assert all(p == part for p in parts)
    if mode.endswith("rebalance") or mode.endswith("diffuse"): assert part != tiled.equal_partition(H, world)
    for b, elem, half in [(abi.BUF_GBUFFER0 + cur, 16, False), (abi.BUF_DIRECT_RESV0 + cur, 36, False), (abi.BUF_LIGHT_ID0 + cur, 4, False),
                          (abi.BUF_INDIRECT_RESV0 + cur, 76, True)]:
        want = ref.readback(b)
        for rank in range(world):   # the frame's history stays distributed: each rank owns its band
            w = W // 2 if half else W
            a, e = (parth[rank], parth[rank + 1]) if half else (part[rank], part[rank + 1])
            got = rs[rank].readback(b).reshape(-1, w * elem)[a:e]
            assert np.array_equal(got, want.reshape(-1, w * elem)[a:e]), (abi.BUFFER_NAMES[b], rank)


@pytest.mark.parametrize("orbit", [False, True], ids=["static-camera", "orbiting-camera"])
def test_bench_gate_on_three_thread_ranks(bistro, orbit):
    """bench.py's tiled == untiled gate (restir_amd/verify.py) with the backend and the frame class the RCCL host uses — HIP renderers, frames in flight, rotating G-buffers —
    on three thread-ranks of ONE GPU: cold history, three frames on an uneven partition, every distributed buffer gathered to rank 0, SHA-256 of all six frame buffers
    against the untiled frames.  (The transport-independent part of the first real multi-device run; the same function over gloo: tests/test_tiled_gloo.py.)"""
    import copy
    import torch
    from restir_amd import tiled, verify as V
    sc, env, st, _cam = bistro
    world, part = 3, [0, 496, 560, H]
    s2, _ = make_scene(abi.PROC_BISTRO_EXT, 1.0, 1, None)
    cams = V.verify_cameras(s2, W, H, s2.cameraPose(), orbit, 3)
    rs = [_renderer(sc, env) for _ in range(world)]
    shared = {"renderers": rs, "barrier": threading.Barrier(world), "slot": [None] * world, "flags": [False] * world}
    errors, got = [], {}

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            rs[rank].update(W, H)                                   # cold history, as the bench does before the gate
            st_r = copy.copy(st)
            fr, cur = V.render_tiled(tiled.PipelinedTiledFrame, tiled.RendererTensors(rs[rank]), ThreadComm(rank, world, shared), W, H, part, cams, st_r, rs[rank].set_camera)
            rs[rank].sync()
            if rank == 0:
                got["tiled"] = V.digests(rs[0].readback, cur); got["cur"] = cur
        except Exception as e:  # pragma: no cover
            errors.append(e); shared["barrier"].abort()

    th = [threading.Thread(target=rank_main, args=(i,)) for i in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errors, errors
    ref = _renderer(sc, env)
    cur = V.render_untiled(ref.run, ref.set_camera, cams, copy.copy(st)); ref.sync()
    assert cur == got["cur"]
    verdict = V.compare(got["tiled"], V.digests(ref.readback, cur))
    assert verdict["equal"] and len(verdict["buffers"]) == 6, verdict
    for r in rs + [ref]:
        r.destroy()


def test_nccl_single_rank_collectives_on_ctx_buffers():
    """The RCCL code path of restir_amd/tiled.py with world_size 1: in-place all-gathers on the ctx-owned HBM buffers wrapped
    as torch tensors, kernels on torch's current stream.  (Multi-rank logic is covered by tests/test_tiled_gloo.py.)"""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from restir_amd import tiled
    from restir_amd.renderer import Renderer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        Ws, Hs = 256, 144
        sc, env = make_scene(abi.PROC_SPONZA, 0.01, 1, (128, 64))
        st = host.default_state(Ws, Hs, sc, env)
        outs = []
        for tiled_mode in (False, True):
            r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(Ws, Hs)
            r.set_stream(torch.cuda.current_stream().cuda_stream)
            fr = tiled.TiledFrame(tiled.RendererTensors(r), tiled.TorchComm(), Ws, Hs) if tiled_mode else None
            s2, _ = make_scene(abi.PROC_SPONZA, 0.01, 1, None)
            s2.updateCamera(Ws, Hs)
            for f in range(3):
                st.time = 321 + f; s2.updateCamera(Ws, Hs); r.set_camera(s2.getCamera())
                if fr is None: r.run(st, f)
                else: fr.render_frame(st, f)
            if fr is not None: fr.finish()
            torch.cuda.synchronize()
            outs.append([r.readback(b) for b in frame_buffers(2)])
        for a, b in zip(*outs):
            assert np.array_equal(a, b)
    finally:
        dist.destroy_process_group()


def test_config5_4k_interior_orbiting_camera():
    """Bistro-Interior-class scene (~1.0 M triangles, ~2 k emissive triangles), 3840x2160, camera orbiting
    0.5 deg per frame.  Frames in flight must leave the buffers of the serial schedule; a 16-row band of the cold frame and of the
    third (temporal-reuse, moving-camera) frame's G-buffer / motion / direct reservoirs must equal the oracle bit for bit."""
    import os
    from oracle.binding import Oracle
    from restir_amd.renderer import Renderer
    W4, H4 = 3840, 2160
    sc, env = make_scene(abi.PROC_BISTRO_INT, 1.0, 1, (512, 256))
    st = host.default_state(W4, H4, sc, env)
    eye, center, up, fov = sc.cameraPose()
    cams = []
    rel = eye - center
    for f in range(4):
        a = np.deg2rad(0.5 * f)
        rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=np.float32)
        sc.setCamera(center + rot @ rel, center, up, fov)
        sc.updateCamera(W4, H4)
        if f > 0: cams.append(sc.getCamera())             # frame index f-1; the first update only primes the history matrices
    out = {}
    keep = frame_buffers(2) + [abi.BUF_GBUFFER0, abi.BUF_DIRECT_RESV0, abi.BUF_INDIRECT_RESV0, abi.BUF_LIGHT_ID0]
    prev = os.environ.get("RESTIR_OVERLAP")
    try:
        for mode in ("2", "0"):
            os.environ["RESTIR_OVERLAP"] = mode
            r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W4, H4)
            assert 0.8e6 < r.accel_stats()["triangles"] < 1.3e6
            for f, cam in enumerate(cams):
                st.time = 5100 + f
                r.set_camera(cam); r.run(st, f)
            out[mode] = {b: r.readback(b) for b in keep}
            if mode == "0":
                hist = {b: r.readback(b) for b in (abi.BUF_GBUFFER1, abi.BUF_DIRECT_RESV1, abi.BUF_LIGHT_ID1)}   # frame 1 = history of frame 2
            r.destroy()
    finally:
        if prev is None: os.environ.pop("RESTIR_OVERLAP", None)
        else: os.environ["RESTIR_OVERLAP"] = prev
    for b in keep:
        assert np.array_equal(out["2"][b], out["0"][b]), abi.BUFFER_NAMES[b]
    # oracle: frame 2's direct stage on rows 1200..1216 with frame 1's GPU history uploaded (temporal reuse across a camera step)
    o = Oracle(0); o.upload_scene(sc.desc(env)); o.resize(W4, H4)
    for b, data in hist.items():
        o.upload_history(b, data)
    st.time = 5100 + 2
    o.set_camera(cams[2])
    y0, y1 = 1200, 1216
    o.run_stage(st, 2, abi.STAGE_DIRECT, 0, y0, y1)
    for buf, elem in [(abi.BUF_GBUFFER0, 16), (abi.BUF_MOTION, 4), (abi.BUF_DIRECT_RESV0, 36), (abi.BUF_LIGHT_ID0, 4), (abi.BUF_DIRECT_RESULT0, 16)]:
        if buf == abi.BUF_DIRECT_RESULT0:
            continue                                       # overwritten by the filters + compose on the GPU side
        got = out["0"][buf].reshape(-1, W4 * elem)[y0:y1]
        ref = o.readback(buf).reshape(-1, W4 * elem)[y0:y1]
        assert np.array_equal(got, ref), abi.BUFFER_NAMES[buf]
    moved = out["0"][abi.BUF_MOTION].view(np.int16).reshape(H4, W4, 2)[y0:y1]
    xx = np.arange(W4)[None, :]
    assert (moved[..., 0] != xx).mean() > 0.2               # the camera really moved: reprojection is not the identity
