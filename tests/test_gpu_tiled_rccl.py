"""restir_amd/tiled.py over REAL RCCL: one process per GPU (the way `bench.py --gpus N` runs), world_size 2 when two devices are
visible — bit-identical to the untiled frame.  On a one-GPU box the same worker runs with world_size 1 (process spawn, NCCL
bootstrap, in-place collectives on the ctx-owned HBM buffers, frames in flight); the multi-rank logic itself is also covered on
CPU over gloo (tests/test_tiled_gloo.py) and natively by tests/test_gpu_mgpu.py."""
import os
import socket
import sys
import numpy as np
import pytest

from helpers import ROOT, abi, host, make_scene, frame_buffers

pytestmark = pytest.mark.gpu
W, H, FRAMES = 320, 208, 4


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _scene():
    sc, env = make_scene(abi.PROC_SPONZA, 0.01, 1, (128, 64))
    return sc, env, host.default_state(W, H, sc, env)


def _cameras(sc):
    eye, center, up, fov = sc.cameraPose()
    cams = []
    sc.updateCamera(W, H)
    for f in range(FRAMES):
        sc.setCamera(eye + np.array([0.05 * f, 0.02 * f, -0.03 * f], dtype=np.float32), center, up, fov)
        sc.updateCamera(W, H); cams.append(sc.getCamera())
    return cams


def _worker(rank, world, port, outdir, pipelined):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    from restir_amd import tiled
    from restir_amd.renderer import Renderer
    r = Renderer().setup(rank)
    ptrs = tiled.RendererTensors.create_streams(r)    # the rank renders on its context's three streams, created before the process group (round 6; bench.py does the same)
    lay = r.stream_layout()["creation_index"]
    assert lay["main"] < lay["side"] < lay["ind"]
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    try:
        sc, env, st = _scene()
        r.load_scene(sc.desc(env)); r.update(W, H)
        stream = torch.cuda.ExternalStream(ptrs["main"]); torch.cuda.set_stream(stream); r.set_stream(stream.cuda_stream)
        Frame = tiled.PipelinedTiledFrame if pipelined else tiled.TiledFrame
        fr = Frame(tiled.RendererTensors(r), tiled.TorchComm(), W, H)
        for f, cam in enumerate(_cameras(sc)):
            st.time = 700 + f; r.set_camera(cam); fr.render_frame(st, f)
        fr.finish(); torch.cuda.synchronize()
        cur = (FRAMES - 1) & 1
        out = {"rows": np.array([fr.y0, fr.y1, fr.h0, fr.h1])}
        for b in (abi.BUF_GBUFFER0 + cur, abi.BUF_DIRECT_RESV0 + cur, abi.BUF_LIGHT_ID0 + cur, abi.BUF_INDIRECT_RESV0 + cur):
            out[abi.BUFFER_NAMES[b]] = r.readback(b)
        if rank == 0:
            for b in (abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_INDIRECT_RESULT0 + cur):
                out[abi.BUFFER_NAMES[b]] = r.readback(b)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pipelined", [False, True], ids=["serial", "frames-in-flight"])
def test_tiled_over_rccl_equals_untiled(pipelined, tmp_path):
    import torch
    import torch.multiprocessing as mp
    from restir_amd.renderer import Renderer
    world = 2 if torch.cuda.device_count() >= 2 else 1
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), pipelined), nprocs=world, join=True)
    sc, env, st = _scene()
    ref = Renderer().setup(0); ref.load_scene(sc.desc(env)); ref.update(W, H)
    for f, cam in enumerate(_cameras(sc)):
        st.time = 700 + f; ref.set_camera(cam); ref.run(st, f)
    cur = (FRAMES - 1) & 1
    elem = {"gbuffer": 16, "direct_resv": 36, "light_id": 4, "indirect_resv": 76}
    for rank in range(world):
        got = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        y0, y1, h0, h1 = (int(v) for v in got["rows"])
        for b in (abi.BUF_GBUFFER0 + cur, abi.BUF_DIRECT_RESV0 + cur, abi.BUF_LIGHT_ID0 + cur, abi.BUF_INDIRECT_RESV0 + cur):
            name = abi.BUFFER_NAMES[b]
            half = name.startswith("indirect")
            w, a, e = (W // 2, h0, h1) if half else (W, y0, y1)
            want = ref.readback(b).reshape(-1, w * elem[name[:-1]])[a:e]
            assert np.array_equal(got[name].reshape(-1, w * elem[name[:-1]])[a:e], want), (name, rank)
        if rank == 0:
            for b in (abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_INDIRECT_RESULT0 + cur):
                assert np.array_equal(got[abi.BUFFER_NAMES[b]], ref.readback(b)), abi.BUFFER_NAMES[b]
