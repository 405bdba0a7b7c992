"""Multi-rank row tiling (restir_amd/tiled.py) on CPU: world_size 2 and 3 over gloo with the oracle as the backend must
reproduce the untiled frame bit for bit, including temporal reuse under a moving camera."""
import os
import socket
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT, abi, host, make_scene

W, H, FRAMES = 96, 208, 4  # 208 rows: bands of 112+96 (world 2) and 80+80+48 (world 3): uneven last band on purpose


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class OracleTensors:
    def __init__(self, o): self.o, self._c = o, {}
    def run_stage(self, state, frames, stage, level, r0, r1): self.o.run_stage(state, frames, stage, level, r0, r1)
    def set_history_rows(self, r0, r1): self.o.set_history_rows(r0, r1)
    def history_miss(self): return self.o.history_miss()
    def history_miss_stage(self, stage): return self.o.history_miss_stage(stage)
    def tensor(self, buf):
        if buf not in self._c:
            arr, pitch = self.o.buffer_array(buf)
            self._c[buf] = (torch.from_numpy(arr), pitch)
        return self._c[buf]


def _camera(sc, f, fast):
    eye, center, up, fov = sc.cameraPose()
    # slow: reprojection stays within the history halo; fast: the camera jumps and the worker runs without a history halo,
    # so temporal reuse leaves the valid rows and the exact fallback (full history all-gather + redo) must kick in
    dy = (0.9 if (fast and f >= 2) else 0.0)
    sc.setCamera((0.03 * f, 1.0 + 0.01 * f + dy, 3.4), (0, 1, 0), (0, 1, 0), fov)


def _setup():
    from oracle.binding import Oracle
    sc, env = make_scene(abi.PROC_CORNELL, env_size=(32, 16))
    st = host.default_state(W, H, sc, env)
    o = Oracle(1); o.upload_scene(sc.desc(env)); o.resize(W, H)
    return sc, env, st, o


def _worker(rank, world, port, outdir, fast, pipelined=False, part=None, replan=False, restir=None):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from restir_amd import tiled
    if fast == 1:
        tiled.HIST_HALO = 0   # no history halo: the first cross-band reprojection must trigger the exact fallback
    # (fast == 2: the adaptive halo — 16 rows until the camera jump makes a lookup leave band + halo, the fallback, then 32)
    sc, env, st, o = _setup()
    if restir is not None: st.ReSTIRState = restir
    frame = (tiled.PipelinedTiledFrame if pipelined else tiled.TiledFrame)(OracleTensors(o), tiled.TorchComm(), W, H, part)
    sc.updateCamera(W, H)
    for f in range(FRAMES):
        st.time = 900 + f; _camera(sc, f, fast); sc.updateCamera(W, H); o.set_camera(sc.getCamera())
        frame.render_frame(st, f)
        if replan == 1 and f == 1:     # switch to another partition in the middle of the sequence: history must follow
            frame.set_partition(REPLAN[world])
        if replan == 2:                # cost feedback: a synthetic cost of 1 per row, 9 per row below row 128 => bands shrink at the bottom
            frame.rebalance(sum(9.0 if y >= 128 else 1.0 for y in range(frame.y0, frame.y1)), smoothing=1.0, max_move=3)
        if replan == 3:                # the diffusion phase: the same synthetic cost, one 16-row stripe per frame towards the slower neighbour
            frame.diffuse(sum(9.0 if y >= 128 else 1.0 for y in range(frame.y0, frame.y1)))
    frame.finish()
    cur = (FRAMES - 1) & 1
    if rank == 0:   # both parities: the last frame and the one before it (whose second half was issued one call later)
        np.savez(os.path.join(outdir, f"tiled_{world}.npz"), fallbacks=np.array([frame.history_fallbacks]),
                 **{abi.BUFFER_NAMES[b]: o.readback(b) for b in _result_buffers(cur) + _result_buffers(cur ^ 1)})
    # every rank owns the authoritative copy of its band of the history buffers
    np.savez(os.path.join(outdir, f"band_{world}_{rank}.npz"), rows=np.array([frame.y0, frame.y1, frame.h0, frame.h1]),
             halo=np.array([frame._halo]), rx=np.array([frame.halo_bytes[k] for k in tiled.HALO_KINDS]),
             **{abi.BUFFER_NAMES[b]: o.readback(b) for b in _history_buffers(cur)})
    dist.barrier(); dist.destroy_process_group()


def _result_buffers(cur):
    return [abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_INDIRECT_RESULT0 + cur]


def _history_buffers(cur):
    return [abi.BUF_GBUFFER0 + cur, abi.BUF_DIRECT_RESV0 + cur, abi.BUF_LIGHT_ID0 + cur, abi.BUF_INDIRECT_RESV0 + cur]


# uneven partitions: bands narrower than the 32-row history halo and the 144-row G-buffer halo (a halo spans several ranks)
UNEVEN = {2: [0, 48, 208], 3: [0, 16, 176, 208]}
REPLAN = {2: [0, 160, 208], 3: [0, 96, 112, 208]}


_ELEM = {"gbuffer": 16, "direct_resv": 36, "light_id": 4, "indirect_resv": 76}


@pytest.mark.parametrize("pipelined", [False, True], ids=["serial", "frames-in-flight"])
@pytest.mark.parametrize("fast", [False, True], ids=["slow-camera", "fast-camera-fallback"])
@pytest.mark.parametrize("world", [2, 3])
def test_tiled_equals_untiled(world, fast, pipelined, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), fast, pipelined), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, f"tiled_{world}.npz"))
    sc, env, st, o = _setup()
    sc.updateCamera(W, H)
    for f in range(FRAMES):
        st.time = 900 + f; _camera(sc, f, fast); sc.updateCamera(W, H); o.set_camera(sc.getCamera()); o.render_frame(st, f)
    cur = (FRAMES - 1) & 1
    # rank 0 ends the frame with both full result images (gathered)
    for b in _result_buffers(cur) + _result_buffers(cur ^ 1):
        assert np.array_equal(got[abi.BUFFER_NAMES[b]], o.readback(b)), abi.BUFFER_NAMES[b]
    # the frame's history (G-buffer, reservoirs, light ids) is distributed: each rank's band must match the untiled frame
    for rank in range(world):
        band = np.load(os.path.join(tmp_path, f"band_{world}_{rank}.npz"))
        y0, y1, h0, h1 = (int(v) for v in band["rows"])
        for b in _history_buffers(cur):
            name = abi.BUFFER_NAMES[b]
            half = name.startswith("indirect")
            w, a, e = (W // 2, h0, h1) if half else (W, y0, y1)
            elem = _ELEM[name[:-1]]
            ref = o.readback(b).reshape(-1, w * elem)[a:e]
            assert np.array_equal(band[name].reshape(-1, w * elem)[a:e], ref), (name, rank)
    assert (int(got["fallbacks"][0]) > 0) == fast


@pytest.mark.parametrize("pipelined", [False, True], ids=["serial", "frames-in-flight"])
@pytest.mark.parametrize("mode", ["uneven", "replan", "feedback", "diffuse"])
@pytest.mark.parametrize("world", [2, 3])
def test_uneven_partitions_equal_untiled(world, mode, pipelined, tmp_path):
    """cost-weighted band heights: a fixed uneven partition, and a switch of partition in the middle of a temporal sequence"""
    part = UNEVEN[world] if mode == "uneven" else None
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), False, pipelined, part, {"uneven": 0, "replan": 1, "feedback": 2, "diffuse": 3}[mode]), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, f"tiled_{world}.npz"))
    sc, env, st, o = _setup()
    sc.updateCamera(W, H)
    for f in range(FRAMES):
        st.time = 900 + f; _camera(sc, f, False); sc.updateCamera(W, H); o.set_camera(sc.getCamera()); o.render_frame(st, f)
    cur = (FRAMES - 1) & 1
    for b in _result_buffers(cur):
        assert np.array_equal(got[abi.BUFFER_NAMES[b]], o.readback(b)), abi.BUFFER_NAMES[b]
    want_rows = {"uneven": UNEVEN[world], "replan": REPLAN[world]}.get(mode)
    ends = []
    for rank in range(world):
        band = np.load(os.path.join(tmp_path, f"band_{world}_{rank}.npz"))
        y0, y1, h0, h1 = (int(v) for v in band["rows"])
        ends.append(y1)
        if want_rows:
            assert (y0, y1) == (want_rows[rank], want_rows[rank + 1])
        for b in _history_buffers(cur):
            name = abi.BUFFER_NAMES[b]
            half = name.startswith("indirect")
            w, a, e = (W // 2, h0, h1) if half else (W, y0, y1)
            elem = _ELEM[name[:-1]]
            assert np.array_equal(band[name].reshape(-1, w * elem)[a:e], o.readback(b).reshape(-1, w * elem)[a:e]), (name, rank)
    if mode == "diffuse":    # the last boundary moved down a stripe per frame (the bottom band is the expensive one) until the two times were within 6 %
        from restir_amd import tiled
        eq = tiled.equal_partition(H, world)
        assert ends[-1] == H and eq[-2] < ends[-2] <= min(H - 16, eq[-2] + 16 * FRAMES) and ends[-2] % 16 == 0
    if mode == "feedback":   # the expensive bottom rows ended up in a shorter band than the equal split's
        from restir_amd import tiled
        eq = tiled.equal_partition(H, world)
        assert ends[-1] == H and (ends[-1] - ends[-2]) < (eq[-1] - eq[-2])


@pytest.mark.parametrize("pipelined", [False, True], ids=["serial", "frames-in-flight"])
@pytest.mark.parametrize("restir", [abi.RESTIR_SPATIAL, abi.RESTIR_SPATIOTEMPORAL], ids=["spatial", "spatiotemporal"])
def test_spatial_reuse_tiled_equals_untiled(restir, pipelined, tmp_path):
    """the spatial-reuse modes across bands: direct stage in two halves (rt_run_stage levels 1 / 2) around an exchange of the cached
    reservoirs' boundary rows; a 16-row band in the middle makes both of its neighbours' picks cross a boundary"""
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), False, pipelined, UNEVEN[world], 0, restir), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, f"tiled_{world}.npz"))
    sc, env, st, o = _setup()
    st.ReSTIRState = restir
    sc.updateCamera(W, H)
    for f in range(FRAMES):
        st.time = 900 + f; _camera(sc, f, False); sc.updateCamera(W, H); o.set_camera(sc.getCamera()); o.render_frame(st, f)
    cur = (FRAMES - 1) & 1
    for b in _result_buffers(cur) + _result_buffers(cur ^ 1):
        assert np.array_equal(got[abi.BUFFER_NAMES[b]], o.readback(b)), abi.BUFFER_NAMES[b]
    for rank in range(world):
        band = np.load(os.path.join(tmp_path, f"band_{world}_{rank}.npz"))
        y0, y1, h0, h1 = (int(v) for v in band["rows"])
        for b in _history_buffers(cur):
            name = abi.BUFFER_NAMES[b]
            half = name.startswith("indirect")
            w, a, e = (W // 2, h0, h1) if half else (W, y0, y1)
            elem = _ELEM[name[:-1]]
            assert np.array_equal(band[name].reshape(-1, w * elem)[a:e], o.readback(b).reshape(-1, w * elem)[a:e]), (name, rank)


def test_band_partition():
    from restir_amd import tiled
    for Hh, world in [(1080, 8), (1080, 4), (1080, 2), (2160, 8), (80, 3), (40, 8)]:
        part = tiled.equal_partition(Hh, world)
        assert part[0] == 0 and part[-1] == Hh and all(p % 16 == 0 or p == Hh for p in part) and all(part[i] <= part[i + 1] for i in range(world))
        assert tiled.half_partition(part, Hh)[-1] == Hh // 2


def test_plan_bands_balances_cost():
    from restir_amd import tiled
    H, world = 1080, 8
    stripes = (H + 15) // 16
    cost = [1.0] * stripes
    for s_ in range(30, 38):
        cost[s_] = 12.0                       # an expensive horizon
    part = tiled.plan_bands(H, world, cost)
    assert part[0] == 0 and part[-1] == H and all(p % 16 == 0 for p in part[:-1]) and all(part[i + 1] > part[i] for i in range(world))
    sums = [sum(cost[part[r] // 16:(part[r + 1] + 15) // 16]) for r in range(world)]
    eq = tiled.equal_partition(H, world)
    sums_eq = [sum(cost[eq[r] // 16:(eq[r + 1] + 15) // 16]) for r in range(world)]
    assert max(sums) < 0.7 * max(sums_eq)      # the heaviest band got lighter
    assert min(part[r + 1] - part[r] for r in range(world)) < 64 < max(part[r + 1] - part[r] for r in range(world))
    moved = tiled.plan_bands(H, world, cost, prev=eq, max_move=2)
    assert all(abs(moved[k] - eq[k]) <= 32 for k in range(world + 1))
    with pytest.raises(AssertionError):
        tiled.plan_bands(32, 4, [1.0, 1.0])    # fewer stripes than ranks


@pytest.mark.parametrize("pipelined", [False, True], ids=["serial", "frames-in-flight"])
def test_adaptive_history_halo(pipelined, tmp_path):
    """16 rows of history until a camera jump makes a temporal lookup leave band + halo: exact fallback for that frame, 32 rows from then on (the rule of
    csrc/mgpu.cpp); the frames stay bit-identical to the untiled ones"""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), 2, pipelined), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, f"tiled_{world}.npz"))
    sc, env, st, o = _setup()
    sc.updateCamera(W, H)
    for f in range(FRAMES):
        st.time = 900 + f; _camera(sc, f, True); sc.updateCamera(W, H); o.set_camera(sc.getCamera()); o.render_frame(st, f)
    cur = (FRAMES - 1) & 1
    for b in _result_buffers(cur) + _result_buffers(cur ^ 1):
        assert np.array_equal(got[abi.BUFFER_NAMES[b]], o.readback(b)), abi.BUFFER_NAMES[b]
    assert int(got["fallbacks"][0]) > 0
    for rank in range(world):
        # (frames in flight: the direct and the indirect stage of the jump frame miss in two consecutive calls — two doublings)
        assert int(np.load(os.path.join(tmp_path, f"band_{world}_{rank}.npz"))["halo"][0]) == (64 if pipelined else 32)


@pytest.mark.parametrize("pipelined", [False, True], ids=["serial", "frames-in-flight"])
@pytest.mark.parametrize("world", [2, 3])
def test_halo_bytes_match_the_plan(world, pipelined, tmp_path):
    """what a rank really receives per steady-state frame (even-row / partial-width items included) is what the transport-free plan (CountingComm) prices"""
    from restir_amd import tiled
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), False, pipelined, UNEVEN[world]), nprocs=world, join=True)
    for rank in range(world):
        rx = np.load(os.path.join(tmp_path, f"band_{world}_{rank}.npz"))["rx"]
        want = tiled.steady_halo_bytes(W, H, world, rank, part=UNEVEN[world], pipelined=pipelined, frames=FRAMES)
        assert {k: int(v) for k, v in zip(tiled.HALO_KINDS, rx)} == want, rank


def test_steady_halo_bytes_budget():
    """SURVEY 8(e) / round-3 verdict: <= 15 MB pulled per rank and frame at 1080p on 8 ranks (history halo 16 rows, G-buffer halo 40 full + even rows to 144,
    noisy indirect colour at its real width); the round-2 exchange (32-row history, 144 full G-buffer rows, full-pitch colour) was 23.0 MB"""
    from restir_amd import tiled
    for pipelined in (True, False):
        per = [tiled.steady_halo_bytes(1920, 1080, 8, r, pipelined=pipelined) for r in range(8)]
        steady = [sum(v for k, v in b.items() if k not in ("gather", "fallback")) for b in per]
        assert max(steady) <= 15e6 and min(steady) > 5e6, steady
        assert all(b["fallback"] == 0 and b["moved"] == 0 for b in per)
    # N = 2: one neighbour each
    b = tiled.steady_halo_bytes(1920, 1080, 2, 0)
    assert b["history"] + b["filter"] <= 7.5e6 and b["gather"] == 32 * 1920 * (1080 - 544)   # rank 0 also receives the other band of the two result images (display)


def test_diffuse_bands_rule():
    """tiled.diffuse_bands: one stripe towards the slower neighbour, only beyond the tolerance, a band keeps one stripe, and a rank whose band changed in this
    step keeps its other boundary (it is measured again first), the slowest rank's boundaries go first — the rule of csrc/mgpu.cpp rebalance()'s second phase."""
    from restir_amd import tiled
    assert tiled.diffuse_bands([0, 64, 128], [1.0, 1.05]) == [0, 64, 128]                 # inside the tolerance
    assert tiled.diffuse_bands([0, 64, 128], [1.0, 1.2]) == [0, 80, 128]                  # rank 1 slower: its band shrinks
    assert tiled.diffuse_bands([0, 64, 128], [1.2, 1.0]) == [0, 48, 128]
    assert tiled.diffuse_bands([0, 112, 128], [1.0, 2.0]) == [0, 112, 128]                # ... but never below one stripe
    assert tiled.diffuse_bands([0, 64, 128, 192], [1.0, 2.0, 1.0]) == [0, 80, 128, 192]   # rank 1 changed at its upper boundary: the lower one waits
    assert tiled.diffuse_bands([0, 64, 128, 192, 256], [1.0, 2.0, 1.0, 2.0]) == [0, 80, 128, 208, 256]
    assert tiled.diffuse_bands([0, 80, 160, 208], [80.0, 336.0, 432.0]) == [0, 80, 176, 208]                          # the slowest rank's boundary goes first
    # times that are a function of the band: repeated steps end in a small limit cycle around the balance point, never worse than where they started
    def ms(part): return [sum(5.0 if y >= 512 else 1.0 for y in range(part[r], part[r + 1])) + (300.0 if part[r] <= 520 < part[r + 1] else 0.0) for r in range(len(part) - 1)]
    part = tiled.equal_partition(1080, 4); start = max(ms(part)); best = start
    for _ in range(40):
        part = tiled.diffuse_bands(part, ms(part)); best = min(best, max(ms(part)))
        assert part[0] == 0 and part[-1] == 1080 and all(b - a >= 16 and a % 16 == 0 for a, b in zip(part[:-1], part[1:-1] + [1088]))
    assert best < 0.8 * start


# ---- the tiled == untiled gate of bench.py --gpus N (restir_amd/verify.py) over gloo, oracle as the backend ---------------------------------------------------
def _verify_worker(rank, world, port, outdir, pipelined, corrupt):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if corrupt:
        os.environ["RESTIR_TEST_CORRUPT_HALO"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import json
    from restir_amd import tiled, verify as V
    sc, env, st, o = _setup()
    comm = tiled.TorchComm()
    res = {}
    for name, orbit in (("workload", False), ("moving_camera", True)):
        cams = V.verify_cameras(sc, W, H, sc.cameraPose(), orbit, 3)
        o.resize(W, H)                                                    # cold history, like Renderer.update in the bench
        fr, cur = V.render_tiled(tiled.PipelinedTiledFrame if pipelined else tiled.TiledFrame, OracleTensors(o), comm, W, H, UNEVEN[world], cams, st, o.set_camera)
        if rank == 0:
            td = V.digests(o.readback, cur)
            from oracle.binding import Oracle
            ref = Oracle(1); ref.upload_scene(sc.desc(env)); ref.resize(W, H)
            V.render_untiled(ref.render_frame, ref.set_camera, cams, st)
            res[name] = V.compare(td, V.digests(ref.readback, cur))
        dist.barrier()
    if rank == 0:
        with open(os.path.join(outdir, "verdict.json"), "w") as fh:
            json.dump(res, fh)
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("pipelined", [False, True], ids=["serial", "frames-in-flight"])
def test_verify_gate(pipelined, tmp_path):
    """bench.py's gate: digests of all six frame buffers of a tiled sequence equal the untiled ones (world 2, uneven bands, static and orbiting camera) ..."""
    import json
    mp.spawn(_verify_worker, args=(2, _free_port(), str(tmp_path), pipelined, False), nprocs=2, join=True)
    res = json.load(open(os.path.join(tmp_path, "verdict.json")))
    for name in ("workload", "moving_camera"):
        assert res[name]["equal"] and len(res[name]["buffers"]) == 6, res[name]


def test_verify_gate_catches_a_corrupted_halo(tmp_path):
    """... and a damaged filter halo (the RESTIR_TEST_CORRUPT_HALO hook of tiled.TorchComm) is reported as a mismatch of the filtered images"""
    import json
    mp.spawn(_verify_worker, args=(2, _free_port(), str(tmp_path), False, True), nprocs=2, join=True)
    res = json.load(open(os.path.join(tmp_path, "verdict.json")))
    assert not res["workload"]["equal"]
    assert not res["workload"]["buffers"]["direct_result0"]["equal"]
