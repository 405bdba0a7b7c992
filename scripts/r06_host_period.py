#!/usr/bin/env python3
"""Round 6, verdict item 1: ONE band (the whole frame) through each host, every host in a fresh process — what the host and its stream layout cost, no halos involved.

  python scripts/r06_host_period.py [--frames K] [--footprint real|lite]          (parent: runs every mode below in a child process, prints a table + one JSON line)
  python scripts/r06_host_period.py --mode <mode>                                   (child)

modes
  single           rt_render_frame, frames in flight (the context's own schedule: what the single-GPU line times)
  native1          rt_mgpu with one rank (csrc/mgpu.cpp: worker thread + the context's streams, created lazily)
  rccl1            NCCL world 1 + tiled.PipelinedTiledFrame (its frames-in-flight schedule forced: RESTIR_TILED_FORCE_PIPELINE=1) on the context's streams (rt_get_streams, created BEFORE init_process_group) — the host the driver launches
  rccl1_torchpool  the same on three streams of torch's pool created after the process group (RESTIR_TILED_TORCH_STREAMS=1): the layout of rounds 1-5
  rccl1_late       the context's streams, but created AFTER init_process_group + a first collective (what "create them before" is worth)
"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODES = ["single", "native1", "rccl1", "rccl1_torchpool", "rccl1_late"]


def child(a):
    import numpy as np
    import torch
    import restir_amd  # noqa: F401
    import bench
    from restir_amd import abi, host, tiled
    from restir_amd.renderer import Renderer, MultiGpuRenderer
    cfg = bench.CONFIGS[4]
    W, H = cfg["size"]
    torch.cuda.set_device(0)
    r = m = None
    if a.mode == "native1":
        m = MultiGpuRenderer().setup([0])
    else:
        r = Renderer().setup(0)
    dist = None
    if a.mode.startswith("rccl1"):
        import torch.distributed as dist
        if a.mode == "rccl1":
            ptrs = tiled.RendererTensors.create_streams(r)
        os.environ["RESTIR_TILED_FORCE_PIPELINE"] = "1"     # world 1 would otherwise fall back to the serial TiledFrame
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29000 + os.getpid() % 2000), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        probe = torch.ones(1, device="cuda"); dist.all_reduce(probe)
        if a.mode == "rccl1_torchpool":
            os.environ["RESTIR_TILED_TORCH_STREAMS"] = "1"
        if a.mode == "rccl1_late":
            ptrs = tiled.RendererTensors.create_streams(r)
    kind = getattr(abi, "PROC_BISTRO_EXT_REAL" if a.footprint == "real" else "PROC_BISTRO_EXT")
    scene = host.Scene().makeProcedural(kind, 1.0, 1)
    env = host.HdrSampling(); env.makeSyntheticSky(cfg["env"][0], cfg["env"][1], 5e4, 7)
    st = host.default_state(W, H, scene, env)
    for k, v in cfg.get("state", {}).items():
        setattr(st, k, v)
    desc = scene.desc(env)
    (m or r).load_scene(desc); (m or r).update(W, H)
    scene.updateCamera(W, H)
    frame = None
    if a.mode.startswith("rccl1"):
        stream = torch.cuda.Stream() if a.mode == "rccl1_torchpool" else torch.cuda.ExternalStream(ptrs["main"])
        torch.cuda.set_stream(stream); r.set_stream(stream.cuda_stream)
        frame = tiled.PipelinedTiledFrame(tiled.RendererTensors(r), tiled.TorchComm(), W, H)
    f = 0

    def step():
        nonlocal f
        st.time = 1000 + f
        scene.updateCamera(W, H)
        cam = scene.getCamera()
        if m is not None:
            m.set_camera(cam); m.run(st, f)
        elif frame is not None:
            r.set_camera(cam); frame.render_frame(st, f)
        else:
            r.set_camera(cam); r.run(st, f)
        f += 1

    def fence():
        if frame is not None:
            frame.finish()
        if m is not None:
            m.sync()
        torch.cuda.synchronize()
    for _ in range(30):
        step()
    fence()
    per = []
    for _rep in range(3):
        t0 = time.perf_counter()
        for _ in range(a.frames):
            step()
        fence()
        per.append((time.perf_counter() - t0) / a.frames * 1e3)
    lay = (m or r).stream_layout()
    print(json.dumps({"mode": a.mode, "ms_per_frame": round(min(per), 4), "passes_ms": [round(x, 4) for x in per], "stream_layout": lay}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--footprint", default="real")
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    if a.mode:
        return child(a)
    rows = []
    for rep in range(a.reps):
        for mode in MODES:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--mode", mode, "--frames", str(a.frames), "--footprint", a.footprint], capture_output=True, text=True, timeout=900)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            d = json.loads(lines[-1]) if lines else {"mode": mode, "error": (p.stderr or "")[-300:]}
            d["rep"] = rep
            rows.append(d)
            print("%-16s rep %d  %s ms/frame   passes %s   streams %s" % (mode, rep, d.get("ms_per_frame"), d.get("passes_ms"), json.dumps(d.get("stream_layout"))), flush=True)
    print(json.dumps({"host_period": rows}))


if __name__ == "__main__":
    main()
