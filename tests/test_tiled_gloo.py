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


def _worker(rank, world, port, outdir, fast, pipelined=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from restir_amd import tiled
    if fast:
        tiled.HIST_HALO = 0   # no history halo: the first cross-band reprojection must trigger the exact fallback
    sc, env, st, o = _setup()
    frame = (tiled.PipelinedTiledFrame if pipelined else tiled.TiledFrame)(OracleTensors(o), tiled.TorchComm(), W, H)
    sc.updateCamera(W, H)
    for f in range(FRAMES):
        st.time = 900 + f; _camera(sc, f, fast); sc.updateCamera(W, H); o.set_camera(sc.getCamera())
        frame.render_frame(st, f)
    frame.finish()
    cur = (FRAMES - 1) & 1
    if rank == 0:   # both parities: the last frame and the one before it (whose second half was issued one call later)
        np.savez(os.path.join(outdir, f"tiled_{world}.npz"), fallbacks=np.array([frame.history_fallbacks]),
                 **{abi.BUFFER_NAMES[b]: o.readback(b) for b in _result_buffers(cur) + _result_buffers(cur ^ 1)})
    # every rank owns the authoritative copy of its band of the history buffers
    np.savez(os.path.join(outdir, f"band_{world}_{rank}.npz"), rows=np.array([frame.y0, frame.y1, frame.h0, frame.h1]),
             **{abi.BUFFER_NAMES[b]: o.readback(b) for b in _history_buffers(cur)})
    dist.barrier(); dist.destroy_process_group()


def _result_buffers(cur):
    return [abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_INDIRECT_RESULT0 + cur]


def _history_buffers(cur):
    return [abi.BUF_GBUFFER0 + cur, abi.BUF_DIRECT_RESV0 + cur, abi.BUF_LIGHT_ID0 + cur, abi.BUF_INDIRECT_RESV0 + cur]


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


def test_band_partition():
    from restir_amd import tiled
    for Hh, world in [(1080, 8), (1080, 4), (1080, 2), (2160, 8), (80, 3), (40, 8)]:
        B = tiled.band_height(Hh, world)
        assert B % 16 == 0 and world * B >= Hh and world * B - Hh <= 128
        rows = [tiled.band_rows(Hh, world, r) for r in range(world)]
        assert rows[0][0] == 0 and max(r[1] for r in rows) == Hh
        assert all(rows[i][1] == rows[i + 1][0] or rows[i + 1][0] == Hh for i in range(world - 1))
