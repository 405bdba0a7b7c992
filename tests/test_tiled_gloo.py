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

W, H, FRAMES = 96, 80, 3   # 80 rows: bands of 48+32 (world 2) and 32+32+16 (world 3): uneven last band on purpose


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class OracleTensors:
    def __init__(self, o): self.o, self._c = o, {}
    def run_stage(self, state, frames, stage, level, r0, r1): self.o.run_stage(state, frames, stage, level, r0, r1)
    def tensor(self, buf):
        if buf not in self._c:
            arr, pitch = self.o.buffer_array(buf)
            self._c[buf] = (torch.from_numpy(arr), pitch)
        return self._c[buf]


def _camera(sc, f):
    eye, center, up, fov = sc.cameraPose()
    sc.setCamera((0.05 * f, 1.0 + 0.02 * f, 3.4), (0, 1, 0), (0, 1, 0), fov)


def _setup():
    from oracle.binding import Oracle
    sc, env = make_scene(abi.PROC_CORNELL, env_size=(32, 16))
    st = host.default_state(W, H, sc, env)
    o = Oracle(1); o.upload_scene(sc.desc(env)); o.resize(W, H)
    return sc, env, st, o


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from restir_amd import tiled
    sc, env, st, o = _setup()
    frame = tiled.TiledFrame(OracleTensors(o), tiled.TorchComm(), W, H)
    sc.updateCamera(W, H)
    for f in range(FRAMES):
        st.time = 900 + f; _camera(sc, f); sc.updateCamera(W, H); o.set_camera(sc.getCamera())
        frame.render_frame(st, f)
    frame.finish()
    cur = (FRAMES - 1) & 1
    if rank == 0:
        np.savez(os.path.join(outdir, f"tiled_{world}.npz"), **{abi.BUFFER_NAMES[b]: o.readback(b) for b in _final_buffers(cur)})
    dist.barrier(); dist.destroy_process_group()


def _final_buffers(cur):
    return [abi.BUF_GBUFFER0 + cur, abi.BUF_DIRECT_RESV0 + cur, abi.BUF_LIGHT_ID0 + cur, abi.BUF_INDIRECT_RESV0 + cur,
            abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_INDIRECT_RESULT0 + cur]


@pytest.mark.parametrize("world", [2, 3])
def test_tiled_equals_untiled(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, f"tiled_{world}.npz"))
    sc, env, st, o = _setup()
    sc.updateCamera(W, H)
    for f in range(FRAMES):
        st.time = 900 + f; _camera(sc, f); sc.updateCamera(W, H); o.set_camera(sc.getCamera()); o.render_frame(st, f)
    cur = (FRAMES - 1) & 1
    for b in _final_buffers(cur):
        assert np.array_equal(got[abi.BUFFER_NAMES[b]], o.readback(b)), abi.BUFFER_NAMES[b]


def test_band_partition():
    from restir_amd import tiled
    for Hh, world in [(1080, 8), (1080, 4), (1080, 2), (2160, 8), (80, 3), (40, 8)]:
        B = tiled.band_height(Hh, world)
        assert B % 16 == 0 and world * B >= Hh and world * B - Hh <= 128 + 16 * world
        rows = [tiled.band_rows(Hh, world, r) for r in range(world)]
        assert rows[0][0] == 0 and max(r[1] for r in rows) == Hh
        assert all(rows[i][1] == rows[i + 1][0] or rows[i + 1][0] == Hh for i in range(world - 1))
