#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: Mrays/s and ms/frame of the ReSTIR DI+GI+denoise frame at 1920x1080 on the
Bistro-Exterior-class scene (BASELINE.json configs[3]; the real asset is absent => seeded procedural stand-in, 2.8 M
triangles, alpha-masked foliage, emissive lamps, synthetic HDR sky: `data: synthetic`).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one frame = Renderer::run's 12 dispatches (renderer.cpp:154-206) at the reference defaults
(sample_example.hpp:154-184: maxDepth 4, temporal ReSTIR, M=4, MIS, denoise on) with a fresh RNG seed per frame.
N>1 row-tiles the frame (restir_amd/tiled.py): total work is fixed => "strong" scaling.

The JSON line carries
  roofline      dominant kernel: algorithmic bytes per launch (screen traffic of SURVEY.md §8d + counted BVH8 node / triangle
                / hit / RIS-candidate gathers x declared sizes) / its mean launch time from HIP events on the launch stream
  cpu_baseline  the CPU oracle (oracle/, "port") timed on a bounded band of the same frame on the host cores (N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

STAGE_NAMES = ["direct_stage", "indirect_stage", "denoise_direct", "denoise_indirect", "compose", "direct_gen", "direct_reuse"]
# closed-form screen traffic per stage at the reference layouts (bytes per stage-grid pixel)
SCREEN_BYTES = {0: 124.0, 1: 204.0, 2: 192.0, 3: 240.0, 4: 68.0}
NODE_B, TRI_B, HIT_B, RIS_B = 80, 64, 12 + 96 + 80, 16 + 96  # bvh8.h node / triangle record; hit gathers; RIS candidate gathers
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scale", type=float, default=1.0, help="scene tessellation scale (1.0 = the 2.8 M-triangle benchmark scene)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-world", type=int, default=0, help="(1 GPU) time ONE rank of an N-way row-tiled frame with communication stubbed out")
    ap.add_argument("--emulate-rank", type=int, default=-1)
    ap.add_argument("--cpu-rows", type=int, default=256, help="height of the row band the CPU baseline renders")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import restir_amd  # noqa: F401
    from restir_amd import abi, host
    from restir_amd.renderer import Renderer
    from restir_amd import tiled

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        import datetime
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver
        # a stuck exchange should end the run with an error instead of hanging it
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), timeout=datetime.timedelta(seconds=300))

    W, H = args.width, args.height
    scene = host.Scene().makeProcedural(abi.PROC_BISTRO_EXT, args.scale, 1)
    env = host.HdrSampling()
    env.makeSyntheticSky(2048, 1024, 5e4, 7)
    st = host.default_state(W, H, scene, env)
    desc = scene.desc(env)
    r = Renderer().setup(local_rank)
    t0 = time.time()
    r.load_scene(desc)
    build_s = time.time() - t0
    r.update(W, H)
    # kernels and RCCL ops share one HIP runtime (torch's) and one stream: ordered by torch's stream semantics, no host syncs
    stream = torch.cuda.Stream() if world > 1 else None
    if stream is not None:
        torch.cuda.set_stream(stream)
        r.set_stream(stream.cuda_stream)
    comm = tiled.TorchComm() if world > 1 else tiled.LocalComm()
    Frame = tiled.TiledFrame if os.environ.get("RESTIR_TILED") == "serial" else tiled.PipelinedTiledFrame
    frame = Frame(tiled.RendererTensors(r), comm, W, H) if world > 1 else None
    if world == 1 and args.emulate_world > 1:
        class StubComm(tiled.LocalComm):   # per-rank compute + host overhead of the tiled schedule, no real peers
            world = args.emulate_world
            rank = args.emulate_rank if args.emulate_rank >= 0 else args.emulate_world // 2
            def all_gather_rows(self, *a, **k): return None
            def halo_exchange(self, items, async_op=False): return []
            def gather_rows_to(self, *a, **k): return None
            def any_flag(self, flag): return False
        frame = Frame(tiled.RendererTensors(r), StubComm(), W, H)

    scene.updateCamera(W, H)  # prime the camera history (static camera: SURVEY.md §8d)

    def step(f):
        st.time = 1000 + f
        scene.updateCamera(W, H)
        r.set_camera(scene.getCamera())
        if frame is None:
            r.run(st, f)
        else:
            frame.render_frame(st, f)

    def fence():
        if frame is not None:
            frame.finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    f = 0
    for _ in range(args.warmup):
        step(f); f += 1
    fence()
    r.set_counting(False)  # resets the accumulated stage timings
    first_timed = f
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(f); f += 1
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timing = r.counters()

    # ---- rays / traversal counts of the same frames (instrumented kernels, outside the timed region) -------------
    n_count = min(4, args.steps)
    r.set_counting(True)
    for k in range(n_count):
        step(first_timed + k)
    fence()
    cnt = r.counters()
    r.set_counting(False)
    vals = np.array([cnt.closestHitRays, cnt.anyHitRays, cnt.nodesVisited, cnt.trisTested, cnt.hitsShaded, cnt.risCandidates], dtype=np.float64) / n_count
    if dist is not None:
        t = torch.tensor(vals, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        vals = t.cpu().numpy()
    rays_per_frame = float(vals[0] + vals[1])

    # per-stage counts for the roofline: re-run the dominant stage alone with counting (N=1 only: stages timed in-library)
    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rays_per_frame * args.steps / elapsed / 1e6
        out = {
            "metric": "Mrays/s (ClosestHit+AnyHit ray queries per second) of the 1080p ReSTIR DI+GI+denoise frame; ms_per_step = ms/frame",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"bistro-exterior-class procedural scene, {scene.getStat()['instancedTriangles']} triangles, {W}x{H}, "
                                   "ReSTIR DI (temporal, M=4) + GI (maxDepth 4, MIS) + A-Trous 4+5 levels + compose, static camera, "
                                   "2048x1024 synthetic HDR sky", "width": W, "height": H, "scene_scale": args.scale,
                       "parallelism": ("single GPU" if frame is None else f"ONE rank of an emulated {args.emulate_world}-way row tiling, communication stubbed (not a benchmark result)") if world == 1 else f"row-tiled x{world}: frames in flight on 3 streams per rank, neighbour halo exchanges over RCCL (restir_amd/tiled.py PipelinedTiledFrame)",
                       "rays_per_frame": round(rays_per_frame), "fps": round(1e3 / ms_per_step, 2), "bvh8_build_s": round(build_s, 2),
                       "accel": r.accel_stats()},
        }
    if world == 1 and timing.framesTimed > 0:
        stage_ms = [timing.stageMs[i] / max(1, timing.framesTimed) for i in range(5)]
        frame_latency_ms = timing.frameMs / max(1, timing.framesTimed)
        # The timed region runs with frames in flight (rt_set_overlap mode 2): kernels of consecutive frames share the chip,
        # so each launch is stretched while the frame rate goes up.  A short extra pass with every launch alone on one
        # stream gives the un-overlapped duration of the same kernels on the same frames.
        serial_ms = None
        if frame is None:
            r.set_overlap(0)
            for k in range(2):
                step(first_timed + k)
            r.sync(); r.set_counting(False)
            ns = min(10, args.steps)
            for k in range(ns):
                step(first_timed + k)
            r.sync()
            ts = r.counters()
            serial_ms = [ts.stageMs[i] / max(1, ts.framesTimed) for i in range(5)]
            r.set_overlap(2 if os.environ.get("RESTIR_OVERLAP") is None else int(os.environ["RESTIR_OVERLAP"]))
        dom = int(np.argmax(stage_ms))
        launches = {0: 1, 1: 1, 2: 4, 3: 5, 4: 1}[dom]
        # counts of the dominant stage alone
        r.set_counting(True)
        for k in range(n_count):
            st.time = 1000 + first_timed + k
            levels = range(launches) if dom in (2, 3) else [0]
            for lv in levels:
                r.run_stage(st, first_timed + k, dom, lv)
        r.sync()
        c2 = r.counters()
        r.set_counting(False)
        grid_px = (W // 2) * (H // 2) if dom in (1, 3) else W * H
        b_screen = SCREEN_BYTES[dom] * grid_px / launches
        b_trav = (c2.nodesVisited * NODE_B + c2.trisTested * TRI_B + c2.hitsShaded * HIT_B + c2.risCandidates * RIS_B) / float(n_count) / launches
        dur_ms = stage_ms[dom] / launches
        achieved = (b_screen + b_trav) / (dur_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": STAGE_NAMES[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                           "algorithmic_bytes_per_launch": round(b_screen + b_trav), "screen_bytes": round(b_screen), "traversal_bytes": round(b_trav),
                           "launch_ms": round(dur_ms, 4), "stage_ms_per_frame": {STAGE_NAMES[i]: round(stage_ms[i], 4) for i in range(5)},
                           "frame_latency_ms": round(frame_latency_ms, 4)}
        # HBM traffic of the same kernel from the PMC passes (scripts/pmc.sh: separate rocprofv3 --pmc runs of this command;
        # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950's FETCH_SIZE counts wide reads at half their size — MI355X_MICROARCH.md,
        # HBM section — so it is doubled; WRITE_SIZE is taken as reported).  The file is refreshed with the profiles.
        t = None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as fh:
                t = json.load(fh).get({0: "k_direct_stage", 1: "k_indirect_stage", 2: "k_denoise", 3: "k_denoise", 4: "k_compose"}[dom])
            if t and dom in (0, 1, 4):
                out["roofline"]["traffic"] = round(2 * t["FETCH_SIZE_KB"] * 1024 + t["WRITE_SIZE_KB"] * 1024)
                out["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch, separate run)"
        except (OSError, ValueError, KeyError):
            pass
        if serial_ms is not None:  # same kernel, same bytes, launched alone (no other frame's kernels beside it)
            sdur = serial_ms[dom] / launches
            out["roofline"]["serial"] = {"launch_ms": round(sdur, 4), "achieved": round((b_screen + b_trav) / (sdur * 1e-3) / 1e9, 2),
                                         "frac": round((b_screen + b_trav) / (sdur * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                         "stage_ms_per_frame": {STAGE_NAMES[i]: round(serial_ms[i], 4) for i in range(5)},
                                         "frame_ms": round(sum(serial_ms), 4)}
            # What actually bounds the kernel: VALU issue.  Wave-level VALU instructions per launch from the same PMC
            # passes (SQ_INSTS_VALU; a wave64 instruction occupies its SIMD16 for 4 cycles) against 1024 SIMDs x 2.4 GHz, and the
            # fraction of lanes active in those instructions (SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)).
            try:
                if t and t.get("INSTS_VALU"):
                    issue_ms = t["INSTS_VALU"] * 4.0 / (1024 * 2.4e9) * 1e3
                    out["roofline"]["valu"] = {"wave_insts_per_launch": round(t["INSTS_VALU"]), "issue_ms_at_peak": round(issue_ms, 4),
                                               "issue_frac_serial": round(issue_ms / sdur, 4), "lane_utilisation": round(t["THREAD_CYCLES_VALU"] / (64.0 * t["INSTS_VALU"]), 4),
                                               "source": "profiles/pmc_traffic.json (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU per launch)"}
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as fh:
                    fr = json.load(fh).get("_frame")
                if fr and "valu" in out["roofline"] and world == 1 and args.emulate_world <= 1:
                    f_ms = fr["INSTS_VALU"] * 4.0 / (1024 * 2.4e9) * 1e3   # every kernel of one frame
                    out["roofline"]["valu"].update({"frame_wave_insts": round(fr["INSTS_VALU"]), "frame_issue_ms_at_peak": round(f_ms, 4),
                                                    "frame_issue_frac": round(f_ms / out["ms_per_step"], 4)})
            except (NameError, KeyError, ZeroDivisionError, OSError, ValueError):
                pass
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(abi, host, scene, env, st, desc, W, H, args.cpu_rows, first_timed)
    if rank == 0 and frame is not None and out is not None and "roofline" not in out:
        # Row-tiled run (or its one-GPU emulation): the dominant kernel of THIS rank — the direct stage on the rank's band — launched
        # alone after the timed region, timed with HIP events on the stream it runs on; algorithmic bytes from the instrumented
        # kernels on the same band (same per-unit figures as the N = 1 line).
        try:
            y0, y1 = frame.y0, frame.y1
            fdom = first_timed + min(2, args.steps - 1)
            st.time = 1000 + fdom
            cur = torch.cuda.Stream()                     # (the default stream's handle 0 would select the ctx-owned stream)
            r.set_stream(cur.cuda_stream)
            r.run_stage(st, fdom, abi.STAGE_DIRECT, 0, y0, y1); torch.cuda.synchronize()
            reps = 5
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            for _ in range(reps):
                r.run_stage(st, fdom, abi.STAGE_DIRECT, 0, y0, y1)
            e1.record(cur); torch.cuda.synchronize()
            dur_ms = e0.elapsed_time(e1) / reps
            r.set_counting(True)
            r.run_stage(st, fdom, abi.STAGE_DIRECT, 0, y0, y1); r.sync()
            cb = r.counters(); r.set_counting(False)
            b_screen = SCREEN_BYTES[0] * W * (y1 - y0)
            b_trav = cb.nodesVisited * NODE_B + cb.trisTested * TRI_B + cb.hitsShaded * HIT_B + cb.risCandidates * RIS_B
            ach = (b_screen + b_trav) / (dur_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "direct_stage (this rank's band, rows %d..%d, launched alone)" % (y0, y1), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes_per_launch": round(b_screen + b_trav),
                               "screen_bytes": round(b_screen), "traversal_bytes": round(b_trav), "launch_ms": round(dur_ms, 4)}
        except Exception as e:  # the headline number must not depend on this extra pass
            out["roofline"] = {"bound": "hbm", "error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(abi, host, scene, env, st, desc, W, H, rows, frame0):
    """The CPU oracle (naive binary BVH + scalar C++, std::thread over all host cores) on a bounded sample: the full
    12-dispatch frame restricted to a horizontal band of `rows` full-res rows around the image centre."""
    from oracle.binding import Oracle
    o = Oracle(0)
    o.upload_scene(desc)
    o.resize(W, H)
    y0 = (H // 2 // 16) * 16
    y1 = min(H, y0 + rows)
    h0, h1 = y0 // 2, y1 // 2
    o.set_camera(scene.getCamera())
    st.time = 1000 + frame0
    o.reset_counters()
    t0 = time.perf_counter()
    o.run_stage(st, frame0, abi.STAGE_DIRECT, 0, y0, y1)
    o.run_stage(st, frame0, abi.STAGE_INDIRECT, 0, h0, h1)
    for l in range(4):
        o.run_stage(st, frame0, abi.STAGE_DENOISE_DIRECT, l, y0, y1)
    for l in range(5):
        o.run_stage(st, frame0, abi.STAGE_DENOISE_INDIRECT, l, h0, h1)
    o.run_stage(st, frame0, abi.STAGE_COMPOSE, 0, y0, y1)
    dt = time.perf_counter() - t0
    c = o.counters()
    rays = c.closestHitRays + c.anyHitRays
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": o.threads, "kind": "port",
            "sample": f"rows {y0}..{y1} of one {W}x{H} frame (all 12 dispatches, cold temporal history), {rays} rays in {dt:.2f} s "
                      f"=> {dt * H / max(1, y1 - y0) * 1e3:.0f} ms/frame extrapolated"}


if __name__ == "__main__":
    main()
