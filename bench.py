#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: Mrays/s and ms/frame of the ReSTIR DI+GI+denoise frame at 1920x1080 on the
Bistro-Exterior-class scene (BASELINE.json configs[3]; the real asset is absent => seeded procedural stand-in, 2.8 M
triangles, alpha-masked foliage, emissive lamps, synthetic HDR sky: `data: synthetic`; since round 5 with the asset's memory
footprint — 147 materials, 3.4 GB of full-size BGRA8 textures, 16 distinct cut-out cards, long thin triangles; `--scene-footprint
lite` is the cache-resident scene of rounds 1-4).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one frame = Renderer::run's 12 dispatches (renderer.cpp:154-206) at the reference defaults
(sample_example.hpp:154-184: maxDepth 4, temporal ReSTIR, M=4, MIS, denoise on) with a fresh RNG seed per frame.
N>1 row-tiles the frame (restir_amd/tiled.py): total work is fixed => "strong" scaling.

The JSON line carries
  roofline      dominant kernel: algorithmic bytes per launch (screen traffic of SURVEY.md §8d + counted BVH8 node / triangle
                / hit / RIS-candidate gathers x declared sizes) / its mean launch time from HIP events on the launch stream
  cpu_baseline  the CPU oracle (oracle/, "port") timed on a bounded band of the same frame on the host cores (N=1 only)
"""
import time as _time_mod
_T_START = _time_mod.time()   # wall clock of this process: `wall_s` of the N > 1 lines (the driver's limit for the command is 1 800 s)
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
# SURVEY.md §8(d) per-unit figures: 80 B per node visit, 48 B per triangle test (v0, e1, e2, id, flags — the record in HBM is 64 B
# because it also carries the opacity micro-map, csrc/bvh8.h; the algorithmic figure stays the survey's), hit gathers, RIS candidate gathers
NODE_B, TRI_B, HIT_B, RIS_B = 80, 48, 12 + 96 + 80, 16 + 96
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E
# Round 5: the `real` footprint scenes are the default.  The headline moved by 52 % when the exterior stand-in got the memory footprint and triangle shapes of the
# asset it stands in for (2.88 -> 4.39 ms per frame on one box, profiles/r05_real_vs_lite.txt): the number that describes a Bistro-class workload is the latter.
DEFAULT_FOOTPRINT = "real"


# SURVEY.md 8(d) configurations 2-5 (config 1 is the CPU-only plumbing case: tests/test_oracle.py).  The real assets are absent from the image: seeded procedural
# stand-ins of the same class (host/scene_gen.cpp).
CONFIGS = {
    2: {"name": "Cornell box", "kind": "PROC_CORNELL", "size": (512, 512), "env": None, "di_only": True,
        "state": {"environmentProb": 0.0, "fireflyClampThreshold": 100.0}, "pipeline": "ReSTIR DI only (temporal, M=4, clamp 80): direct stage alone"},
    3: {"name": "Sponza-class", "kind": "PROC_SPONZA", "size": (1920, 1080), "env": (2048, 1024), "state": {"maxDepth": 2},
        "pipeline": "ReSTIR DI (temporal, M=4) + GI (maxDepth 2 = one indirect bounce, MIS) + A-Trous 4+5 levels + compose"},
    4: {"name": "bistro-exterior-class", "kind": "PROC_BISTRO_EXT", "size": (1920, 1080), "env": (2048, 1024),
        "pipeline": "ReSTIR DI (temporal, M=4) + GI (maxDepth 4, MIS) + A-Trous 4+5 levels + compose"},
    5: {"name": "bistro-interior-class", "kind": "PROC_BISTRO_INT", "size": (3840, 2160), "env": (512, 256), "orbit": True,
        "pipeline": "ReSTIR DI (temporal, M=4) + GI (maxDepth 4, MIS) + A-Trous 4+5 levels + compose, temporal reuse under motion"},
}


def lib_sha256_16():
    import hashlib
    from restir_amd.renderer import HIP_LIB_PATH
    with open(HIP_LIB_PATH, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


REAL_KINDS = {4: "PROC_BISTRO_EXT_REAL", 3: "PROC_SPONZA_1K"}   # host/scene.hpp: the same classes with the reference's texture upload rules (full size, BGRA8, no mips)


def scene_kind(abi, config, footprint):
    """ProcScene of a configuration.  `real` (configs 3 and 4): the memory footprint of the asset the scene stands in for (config 4: 128 materials with 2k^2 texture
    sets = 3.3 GB of texels, 16 distinct 1k^2 foliage cards, long thin triangles; config 3: SURVEY 8(d)'s 1k^2 textures); `lite`: the rounds 1-4 scenes (60 MB of 512^2
    textures, one 256^2 cut-out — a working set that fits the Infinity Cache)."""
    if footprint == "real" and config in REAL_KINDS:
        return getattr(abi, REAL_KINDS[config])
    return getattr(abi, CONFIGS[config]["kind"])


def footprint_of(args):
    return args.scene_footprint if args.config in REAL_KINDS else "lite"


def workload_key(config, orbit, footprint="lite", pose=0):
    return f"config{config}" + ("_moving" if orbit and config != 5 else "") + ("_real" if footprint == "real" and config in REAL_KINDS else "") + (f"_pose{pose}" if pose else "")


# --pose N (config 4): the headline scene from fixed camera poses.  Round-5 verdict, missing 3: every conclusion of round 5 was a statement about ONE view — the street of
# trees seen end-on, whose horizon tiles carry the frame (the same scene 15 degrees off axis renders 2.5 x faster).  0 = the scene's own camera (end-on down the street),
# 1 = the same eye orbited 15 degrees about its centre of interest (what the orbiting camera of --moving-camera sees after 30 frames), 2 = an elevated view across the street
# (eye at balcony height on one side, looking down at the other side's facades, awnings, chairs and tree crowns from above: no horizon, no end-on row of trees).
POSES = {0: "end-on down the street (the scene's camera)", 1: "orbited 15 degrees about the centre of interest", 2: "elevated, across the street"}


def apply_pose(args, scene):
    pose = getattr(args, "pose", 0)
    if not pose:
        return
    eye, center, up, fov = scene.cameraPose()
    if pose == 1:
        a = np.deg2rad(15.0)
        rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=np.float32)
        scene.setCamera(center + rot @ (eye - center), center, up, fov)
    elif pose == 2:
        scene.setCamera(np.array([-30.0, 13.0, 10.5], dtype=np.float32), np.array([-4.0, 2.0, -9.0], dtype=np.float32), up, fov)
    else:
        raise SystemExit(f"--pose {pose}: 0, 1 or 2")


def texture_bytes(desc):
    """BGRA8 bytes of every image of the upload payload (rt_texture: pointer, width, height, ... = 32 B, include/rt_abi.h)"""
    import ctypes
    n = int(desc.numTextures)
    if not desc.textures or n == 0:
        return 0
    t = np.frombuffer((ctypes.c_char * (n * 32)).from_address(desc.textures), dtype=np.dtype([("ptr", "<u8"), ("w", "<i4"), ("h", "<i4"), ("rest", "<i4", 4)]))
    return int((t["w"].astype(np.int64) * t["h"] * 4).sum())


def pmc_entry(key, kname):
    """(per-launch counters of kernel `kname` for workload `key`, the whole file) from profiles/pmc_traffic.json — only if they were collected on THIS build of
    the library (scripts/pmc.sh stamps the file with the library hash); otherwise (None, file-or-None): counters of another build are history, not measurement."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
    except (OSError, ValueError):
        return None, None
    if pmc.get("_lib_sha256_16") != lib_sha256_16():
        return None, pmc
    w = pmc.get("workloads", {}).get(key)
    if w is None and key == "config4":
        w = pmc          # (the layout of rounds 2-3: the headline workload at top level)
    return (w.get(kname) if w else None), pmc


def label_bound(roof, t, dur_ms, peak_mix, alg_bytes):
    """`bound` follows the counters of THIS workload on THIS build, or is null: the three fractions of the dominant kernel side by side at top level."""
    roof["frac_algorithmic"] = roof.get("frac")
    roof["frac_hbm_counter"] = roof["frac_valu"] = None
    if not t:
        roof["bound"] = None
        roof["bound_evidence"] = "no counter pass of this workload on this build (scripts/pmc.sh <tag> <key> <bench args>): the label is withheld rather than defaulted"
        return
    traffic = 2 * t["FETCH_SIZE_KB"] * 1024 + t["WRITE_SIZE_KB"] * 1024
    roof["traffic"] = round(traffic)
    roof["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch, separate run)"
    hb = traffic / (dur_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    roof["hbm_counter"] = {"bytes_per_launch": round(traffic), "achieved": round(traffic / (dur_ms * 1e-3) / 1e9, 2), "frac": round(hb, 5),
                           "counter_over_algorithmic": round(traffic / max(1.0, alg_bytes), 4)}
    roof["frac_hbm_counter"] = round(hb, 5)
    if t.get("INSTS_VALU") and peak_mix:
        roof["frac_valu"] = round(t["INSTS_VALU"] / (dur_ms * 1e-3) / peak_mix, 4)
        roof["bound"] = "valu" if roof["frac_valu"] > hb else "hbm"
        roof["bound_evidence"] = (f"VALU issue at {roof['frac_valu']:.2f} of the ceiling measured in this run vs HBM at {hb:.3f} of 8 TB/s (counter bytes of this workload); "
                                  "no MFMA on this path; frac / frac_algorithmic count per-lane touches that L1 / L2 mostly serve")
    else:
        roof["bound"] = None
        roof["bound_evidence"] = "HBM counters present, VALU counters or ceiling missing"


def di_only_roofline(r, abi, st, W, H, first_timed, n_count, ms_per_step):
    """config 2: the direct stage is the whole step.  Algorithmic bytes as for the headline line; the launch time is the step time (one launch per step)."""
    r.set_counting(True)
    for k in range(n_count):
        st.time = 1000 + first_timed + k
        r.run_stage(st, first_timed + k, abi.STAGE_DIRECT, 0)
    r.sync()
    c2 = r.counters(); r.set_counting(False)
    b_screen = SCREEN_BYTES[0] * W * H
    b_trav = (c2.nodesVisited * NODE_B + c2.trisTested * TRI_B + c2.hitsShaded * HIT_B + c2.risCandidates * RIS_B) / float(n_count)
    ach = (b_screen + b_trav) / (ms_per_step * 1e-3) / 1e9
    roof = {"bound": None, "kernel": "direct_stage", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
            "algorithmic_bytes_per_launch": round(b_screen + b_trav), "screen_bytes": round(b_screen), "traversal_bytes": round(b_trav), "launch_ms": round(ms_per_step, 4),
            "note": "launch_ms = wall clock per step (launch-to-launch, one direct-stage launch per step): a 512x512 launch is bound by launch rate and latency, not by bandwidth "
                    "(a 36-triangle scene never leaves the caches: `frac` counts per-lane touches, not bytes that crossed the fabric)"}
    t, _ = pmc_entry("config2", "k_direct_stage")
    peak = None
    try:
        peak = r.measure_valu_peak(0, 8)
    except Exception:
        pass
    label_bound(roof, t, ms_per_step, peak, b_screen + b_trav)
    return roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=4, choices=[2, 3, 4, 5], help="BASELINE.json / SURVEY.md 8(d) configuration (4 = the headline workload)")
    ap.add_argument("--moving-camera", action="store_true", help="the camera orbits its centre of interest by 0.5 degrees per frame (always on for --config 5)")
    ap.add_argument("--via-gltf", action="store_true", help="N = 1: the scene goes through the reference's entry point — it is written as a glTF asset (Scene::saveGltf: .gltf + .bin + "
                    "one PNG per image), read back by Scene::load (host/gltf_loader.cpp <-> src/scene.cpp:57-125), the digest of everything rt_upload_scene reads is compared with the "
                    "procedural scene's, the LOADED scene is what the line times, and three cold-history frames of both are compared buffer by buffer; load seconds and peak RSS in the line")
    ap.add_argument("--pose", type=int, default=0, choices=[0, 1, 2], help="config 4: fixed camera pose (0 = the scene's own, end-on down the street; 1 = orbited 15 degrees; 2 = elevated, across the street)")
    ap.add_argument("--scene-footprint", choices=["lite", "real"], default=os.environ.get("RESTIR_SCENE_FOOTPRINT", DEFAULT_FOOTPRINT),
                    help="configs 3 / 4: `real` = the texture / material / triangle-shape footprint of the asset the procedural scene stands in for (see scene_kind); "
                         "`lite` = the cache-resident scenes of rounds 1-4")
    ap.add_argument("--scale", type=float, default=1.0, help="scene tessellation scale (1.0 = the configuration's triangle count)")
    ap.add_argument("--width", type=int, default=0, help="default: the configuration's width")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-run", action="store_true", help="only the warm-up + K timed frames + the counting pass (for rocprofv3 --pmc / --kernel-trace runs)")
    ap.add_argument("--emulate-world", type=int, default=0, help="(1 GPU) N ranks of the native row-tiled frame (rt_mgpu_*) on this device, taking turns: per-rank stage times of a chip to itself")
    ap.add_argument("--equal-bands", action="store_true", help="N > 1 (and --emulate-world): equal-height row bands instead of cost-weighted ones")
    ap.add_argument("--band-rounds", type=int, default=8, help="N > 1: planning rounds of the cost-weighted band heights before the warm-up")
    ap.add_argument("--diffuse-rounds", type=int, default=10, help="N > 1: after the cost-model rounds the band boundaries diffuse one 16-row stripe per round towards the slower "
                                                                  "neighbour (restir_amd/tiled.py diffuse_bands); the best partition seen is kept")
    ap.add_argument("--no-period", action="store_true", help="--emulate-world: skip the frames-in-flight period pass")
    ap.add_argument("--solo-fresh", type=int, default=0, help="--emulate-world: re-measure the period of the K slowest ranks, each in a FRESH child process in which only that rank ever "
                    "creates its two extra streams (round 6: a stream's worth depends on how many streams the process created before it; in-process, every rank measured earlier has left its streams behind)")
    ap.add_argument("--emulate-child", type=str, default="", help="(internal, --solo-fresh) `rank:b0,b1,...,bN`: measure that rank's period on that partition and print one JSON line")
    ap.add_argument("--period-rounds", type=int, default=4, help="--emulate-world: re-planning rounds of the band heights on the measured per-rank periods")
    ap.add_argument("--native", action="store_true", help="N > 1: ONE process drives the N devices through the native context (rt_mgpu_*, csrc/mgpu.cpp: frames in flight per rank, "
                                                         "event-ordered peer pulls) instead of one process per GPU over RCCL; same JSON line plus the measured link time of every pull group")
    ap.add_argument("--devices", type=str, default="", help="--native: explicit device list, e.g. 0,1,2,3; a list with repeats (0,0) runs several ranks on one device — a functional check "
                                                            "of the host (tests/test_gpu_bench_cli.py), reported with n_gpus = the number of DISTINCT devices")
    ap.add_argument("--stream-priorities", type=str, default="auto", help="N = 1: `auto` (default) = the library decides at its first frame (rt_render_frame; reported in the "
                    "line), `default` = the levels of rounds 2-4 (+1, 0), or `i,f` with levels -1 / 0 / +1")
    ap.add_argument("--verify-frames", type=int, default=3, help="N > 1: frames of the tiled == untiled gate after the timed region (0 = skip the gate; the line then says so)")
    ap.add_argument("--single-host", action="store_true", help="N > 1 without --native: time the RCCL host only (default: rank 0 then also runs the native host in a child process and "
                                                                "reports both, the faster as `value`)")
    ap.add_argument("--print-workload-key", action="store_true", help="print the key of this workload in profiles/pmc_traffic.json and exit (scripts/pmc.sh)")
    args = ap.parse_args()
    if args.print_workload_key:
        print(workload_key(args.config, args.moving_camera or CONFIGS[args.config].get("orbit", False), footprint_of(args), args.pose))
        return None

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1 and not args.native and not os.environ.get("RESTIR_CPUS"):
        # one process per GPU on ONE host: the ranks generate their scenes and build their trees at the same time, so each takes its share of the CPUs the container may use
        # (include/rt_cpus.h reads RESTIR_CPUS; round 6: eight ranks x 512 builder threads on a 16-CPU quota)
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        os.environ["RESTIR_CPUS"] = str(max(2, min(ncpu, cpu_quota() or ncpu) // max(1, local_world)))
        os.environ["RESTIR_CPUS_SHARED"] = "1"      # (ours, not the user's: the native host rank 0 starts afterwards in a child process gets the whole budget again)

    import torch
    import restir_amd  # noqa: F401
    from restir_amd import abi, host
    from restir_amd.renderer import Renderer
    from restir_amd import tiled

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # --gpus N means N devices, whoever launched this process.  A plain `python bench.py --gpus 8` (no WORLD_SIZE) used to render on ONE device and print
    # "n_gpus": 1 (round-3 verdict): it now refuses when fewer than N devices are visible and otherwise launches the N ranks itself.
    # TEST HOOK (tests/test_gpu_bench_cli.py): RESTIR_BENCH_SHARE_DEVICE=1 lets the ranks of a `--gpus N` run share the devices there are (rank r on device r % count) so
    # that the N > 1 code path of THIS file — band planning, the gate, the rank report, both hosts — executes on a one-GPU box; with RESTIR_DIST_BACKEND=gloo as the
    # transport (RCCL refuses two ranks on one device).  The line then says n_gpus = the DISTINCT devices and carries a note: a functional check, never a result.
    share = os.environ.get("RESTIR_BENCH_SHARE_DEVICE") == "1"
    if share and args.native and not args.devices:
        args.devices = ",".join(str(q % torch.cuda.device_count()) for q in range(args.gpus))
    if args.gpus > 1 and torch.cuda.device_count() < args.gpus and not (args.native and args.devices) and not share:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} device(s) visible on this host — refusing to report a {args.gpus}-GPU number "
                         f"from fewer devices (use --emulate-world {args.gpus} for the one-device per-rank instrument)")
    if args.gpus > 1 and args.native:
        if rank != 0:
            return None                      # (launched under torch.distributed.run: the native host is ONE process, rank 0's)
        return native_world(args, abi, host, Renderer, torch)
    if args.gpus > 1 and world == 1:
        return self_launch(args)
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    r = Renderer().setup(local_rank)
    ctx_streams = None
    if world > 1:
        # Round 6: the rank renders on its CONTEXT's three streams (rt_get_streams: created here, filter stream first, then the indirect stream — the layout the schedule
        # was tuned on) instead of on three streams of torch's pool, and they exist BEFORE the process group does: RCCL's communicator brings streams of its own, and a
        # stream's worth depends on how many the process created before it (profiles/r05_prio_by_config_ab.txt section 2; profiles/r06_mgpu_streams_ab.txt).
        ctx_streams = tiled.RendererTensors.create_streams(r)
    if world > 1:
        import torch.distributed as dist
        import datetime
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver
        # a stuck exchange should end the run with an error instead of hanging it
        try:
            backend = os.environ.get("RESTIR_DIST_BACKEND", "nccl")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), timeout=datetime.timedelta(seconds=300))
            else:
                dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300))
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe)                       # the first collective builds the communicator: a broken fabric / IPC setup fails HERE, not in the timed region
            if int(probe.item()) != world:
                raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
        except Exception as e:   # noqa: BLE001 — whatever RCCL raises: the run must still produce a line
            # RCCL could not be brought up: say so with what a maintainer needs (devices, peer access) and fall through to the native host, which needs no
            # process group (one process, hipMemcpyPeerAsync).  Ranks other than 0 leave: the native host is rank 0's process.
            devs = [torch.cuda.get_device_name(i) for i in range(torch.cuda.device_count())]
            print(f"[bench rank {rank}] RCCL init failed: {e!r}; visible devices: {devs}; peer access: {peer_matrix(torch, args.gpus)}", file=sys.stderr, flush=True)
            try:
                dist.destroy_process_group()
            except Exception:   # noqa: BLE001
                pass
            if rank != 0:
                return None
            return native_world(args, abi, host, Renderer, torch, extra={"rccl": f"failed: {e!r}"[:300]})

    cfg = CONFIGS[args.config]
    W, H = args.width or cfg["size"][0], args.height or cfg["size"][1]
    orbit = args.moving_camera or cfg.get("orbit", False)
    footprint = footprint_of(args)
    scene = host.Scene().makeProcedural(scene_kind(abi, args.config, footprint), args.scale, 1)
    apply_pose(args, scene)
    env = None
    if cfg["env"]:
        env = host.HdrSampling()
        env.makeSyntheticSky(cfg["env"][0], cfg["env"][1], 5e4, 7)
    st = host.default_state(W, H, scene, env)
    for k, v in cfg.get("state", {}).items():
        setattr(st, k, v)
    di_only = cfg.get("di_only", False)
    desc = scene.desc(env)
    via = None
    if args.via_gltf and world == 1 and args.emulate_world <= 1:
        via, scene, desc = via_gltf(args, abi, host, Renderer, scene, env, st, W, H)
    if world == 1 and args.emulate_world > 1 and args.emulate_child:
        return emulate_child(args, abi, host, scene, env, st, desc, r, W, H, local_rank)
    t0 = time.time()
    r.load_scene(desc)
    build_s = time.time() - t0
    r.update(W, H)
    # kernels and RCCL ops share one HIP runtime (torch's) and one stream: ordered by torch's stream semantics, no host syncs
    stream = torch.cuda.ExternalStream(ctx_streams["main"]) if world > 1 else None
    if stream is not None:
        torch.cuda.set_stream(stream)
        r.set_stream(stream.cuda_stream)
    comm = tiled.TorchComm() if world > 1 else tiled.LocalComm()
    Frame = tiled.TiledFrame if os.environ.get("RESTIR_TILED") == "serial" else tiled.PipelinedTiledFrame
    frame = Frame(tiled.RendererTensors(r), comm, W, H) if world > 1 else None
    if world == 1 and args.emulate_world > 1:
        return emulate_world(args, abi, host, scene, env, st, desc, r, W, H, local_rank)

    eye0, center0, up0, fov0 = scene.cameraPose()
    scene.updateCamera(W, H)  # prime the camera history (static camera unless --moving-camera / config 5: SURVEY.md §8d)
    prio_explicit = None
    if args.stream_priorities == "default":
        args.stream_priorities = "1,0"
    if world == 1 and not di_only and "," in args.stream_priorities:
        # (unset, the context decides at its first frame — rt_render_frame: filter stream high next to the indirect stream when the filter chain is >= 14 % of the traced
        #  stages' time; profiles/r05_prio_by_config_ab.txt.  The choice is reported in the line: `stream_priorities`.)
        prio_explicit = [int(x) for x in args.stream_priorities.split(",")]
        r.set_stream_priorities(prio_explicit[0], prio_explicit[1])

    def step(f):
        st.time = 1000 + f
        if orbit:   # the camera orbits its centre of interest, 0.5 degrees per frame (src/scene.cpp:777-826 keeps the history matrices)
            a = np.deg2rad(0.5 * (f + 1))
            rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=np.float32)
            scene.setCamera(center0 + rot @ (eye0 - center0), center0, up0, fov0)
        scene.updateCamera(W, H)
        r.set_camera(scene.getCamera())
        if di_only:
            r.run_stage(st, f, abi.STAGE_DIRECT, 0)        # config 2: ReSTIR DI only, the indirect stage and the filters are skipped
        elif frame is None:
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
    band_plan = None
    last_band_ms = -1.0
    if world > 1 and not args.equal_bands and os.environ.get("RESTIR_EQUAL_BANDS", "0") != "1":
        # Cost-weighted band heights (SURVEY 8(e) "expected scaling limit"), planned before the warm-up: a few rounds of {two real
        # frames, every rank times its band's two traced stages launched alone, the times are gathered, the boundaries move}.  The
        # partition is then fixed: nothing of this runs in the warm-up or in the timed region.
        tstream = stream   # (the stages of a band launched alone: the rank's main stream, idle after the fence — no fourth stream, round 6)

        def band_ms():
            """two real frames on the current partition, then this rank's two traced stages launched alone on its band (ms per pair)"""
            nonlocal f
            for _k in range(2):
                step(f); f += 1
            fence()
            r.set_stream(tstream.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.time = 1000 + f - 1
            r.run_stage(st, f - 1, abi.STAGE_DIRECT, 0, frame.y0, frame.y1); r.run_stage(st, f - 1, abi.STAGE_INDIRECT, 0, frame.h0, frame.h1)   # warm
            e0.record(tstream)
            for _k in range(2):
                r.run_stage(st, f - 1, abi.STAGE_DIRECT, 0, frame.y0, frame.y1); r.run_stage(st, f - 1, abi.STAGE_INDIRECT, 0, frame.h0, frame.h1)
            e1.record(tstream); torch.cuda.synchronize()
            r.history_miss()                                   # the re-runs may have raised the flag again: clear it
            r.set_stream(stream.cuda_stream)
            return e0.elapsed_time(e1) / 2.0

        for _ in range(args.band_rounds):
            last_band_ms = band_ms()
            band_plan = frame.rebalance(last_band_ms, smoothing=0.6, max_move=6)
        # A band's time is not the sum of its stripes' costs (a band with horizon rows takes what its slowest tile takes): the cost model settles with the
        # slowest rank ~1.35x the fastest.  From there the boundaries diffuse, one stripe per round; the partition with the shortest slowest rank is kept.
        seen = []
        for _ in range(args.diffuse_rounds if args.band_rounds > 0 else 0):
            ms = last_band_ms = band_ms()
            worst = max(a[0] for a in frame.comm.all_gather_floats([ms]))   # (the same gathered numbers on every rank)
            seen.append((worst, list(frame.part)))
            new, _spread = frame.diffuse(ms)
            band_plan = new
            if new == seen[-1][1]:
                break
        if seen:
            ms = last_band_ms = band_ms()
            worst = max(a[0] for a in frame.comm.all_gather_floats([ms]))
            seen.append((worst, list(frame.part)))
            best = min(seen, key=lambda t: t[0])[1]
            if best != list(frame.part):
                frame.set_partition(best)
            band_plan = best
        fence()
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
    elapsed_local = elapsed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timing = r.counters()

    # a sustained pass of >= 1 s of back-to-back frames (the K timed steps above are what `value` reports; at the driver's K = 20 they
    # last 70 ms, too short for any sampling monitor to see the GPU busy): same frames-in-flight schedule, reported beside it
    sustained = None
    if world == 1 and frame is None and not args.profile_run and not di_only:
        n_s = max(args.steps, int(1.0 / max(1e-4, elapsed / args.steps)) + 1)
        t1 = time.perf_counter()
        for _ in range(n_s):
            step(f); f += 1
        fence()
        sustained = {"frames": n_s, "ms_per_frame": round((time.perf_counter() - t1) / n_s * 1e3, 4)}
        r.set_counting(False)

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
            "metric": "Mrays/s (ClosestHit+AnyHit ray queries per second) of the " + ("1080p ReSTIR DI+GI+denoise frame" if args.config == 4 else f"config-{args.config} frame") + "; ms_per_step = ms/frame",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config {args.config}: {cfg['name']} procedural scene ({footprint} footprint: {scene.getStat()['materials']} materials, {texture_bytes(desc) / 1e6:.0f} MB of BGRA8 texels), "
                                   f"{scene.getStat()['instancedTriangles']} triangles, {W}x{H}, {cfg['pipeline']}, "
                                   + ("camera orbiting 0.5 deg / frame" if orbit else "static camera") + (f", pose {args.pose}: {POSES[args.pose]}" if getattr(args, "pose", 0) else "") + (f", {cfg['env'][0]}x{cfg['env'][1]} synthetic HDR sky" if cfg["env"] else ", no environment"),
                       "baseline_config": args.config, "scene_footprint": footprint, "texture_bytes": texture_bytes(desc), "width": W, "height": H, "scene_scale": args.scale,
                       "parallelism": ("single GPU" if frame is None else f"ONE rank of an emulated {args.emulate_world}-way row tiling, communication stubbed (not a benchmark result)") if world == 1 else f"row-tiled x{world}: frames in flight on 3 streams per rank, halo exchanges over RCCL (restir_amd/tiled.py PipelinedTiledFrame), " + ("equal-height bands" if band_plan is None else f"cost-weighted bands {band_plan}"),
                       "rays_per_frame": round(rays_per_frame), "fps": round(1e3 / ms_per_step, 2), "bvh8_build_s": round(build_s, 2),
                       "accel": r.accel_stats()},
        }
        out["timing_mode"] = ("ms_per_step = wall clock over K back-to-back frames with frames in flight (throughput; rt_set_overlap(2), the library default); "
                              "ms_per_frame_serial = SURVEY 8(d)'s metric: sum of the frame's kernels, HIP events, every launch alone on one stream, "
                              "median over the frames of a separate pass; frame_latency_ms = first launch to last launch of one frame while frames are in flight")
        if sustained:
            out["sustained"] = sustained
        if via is not None:
            # the frames of the loaded scene against the procedural scene's (reference digests taken before the run): three frames from a cold history, all six buffers
            from restir_amd import verify as V
            r.sync(); r.update(W, H)
            V.render_untiled(r.run, r.set_camera, via.pop("_cams"), st); r.sync()
            got = V.digests(r.readback, 0)
            ref = via.pop("_frame_digests")
            via["frames_equal"] = got == ref
            via["frame_digests"] = {"procedural": ref, "loaded": got}
            out["via_gltf"] = via
        if world == 1 and not di_only:
            sp = r.stream_priorities()
            from restir_amd.renderer import PRIO_FILTER_SHARE
            sp["how"] = "given on the command line" if prio_explicit else ("RESTIR_PRIO" if os.environ.get("RESTIR_PRIO") else
                                                                         f"rt_render_frame's rule on the third (warm) probe frame's stage times (filter stream high when filter_share >= {PRIO_FILTER_SHARE})")
            if sp.get("filter_share") is not None:   # how far the measured share is from the rule's threshold: a decision within a few percent of it can flip between runs
                sp["threshold"] = PRIO_FILTER_SHARE
                sp["margin_rel"] = round(sp["filter_share"] / PRIO_FILTER_SHARE - 1.0, 3)
            out["stream_priorities"] = sp
        if frame is not None:   # what rank 0 received for the last timed frame, by purpose (restir_amd/tiled.py accounting), and the exact fallbacks of the run
            out["halo_bytes_rank0"] = dict(frame.halo_bytes)
            out["history_fallbacks"] = int(frame.history_fallbacks)
    if world == 1 and di_only and not args.profile_run:
        out["roofline"], out["cpu_baseline"] = di_only_roofline(r, abi, st, W, H, first_timed, n_count, elapsed / args.steps * 1e3), None
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(abi, host, scene, env, st, desc, W, H, first_timed, di_only=True)
    if world == 1 and timing.framesTimed > 0 and args.profile_run and not di_only:
        # (round 6) the profiled command's own HIP-event figures: `rocprofv3 --kernel-trace --stats` and the events then describe the SAME launches of the SAME process —
        # the profiler shifts how the kernels of the frames in flight share the chip, so a trace of one process and the events of another differ by more than their error
        out["events_in_this_run"] = {"stage_ms_per_frame": {n: round(timing.stageMs[i] / max(1, timing.framesTimed), 4) for i, n in enumerate(STAGE_NAMES[:5])},
                                     "frames": int(timing.framesTimed), "note": "HIP events on the stream of each launch over the timed frames (rt_get_counters); "
                                     "direct_stage = one k_direct_stage launch: compare with the AverageNs of the kernel stats of this run"}
    if world == 1 and timing.framesTimed > 0 and not args.profile_run and not di_only:
        stage_ms = [timing.stageMs[i] / max(1, timing.framesTimed) for i in range(5)]
        frame_latency_ms = timing.frameMs / max(1, timing.framesTimed)
        # The timed region runs with frames in flight (rt_set_overlap mode 2): kernels of consecutive frames share the chip,
        # so each launch is stretched while the frame rate goes up.  A short extra pass with every launch alone on one
        # stream gives the un-overlapped duration of the same kernels on the same frames.
        serial_ms = None
        if frame is None and not di_only:
            r.set_overlap(0)
            for k in range(2):
                step(first_timed + k)
            r.sync(); r.set_counting(False)
            ns = max(100, min(200, args.steps))             # median of >= 100 frames
            per_frame, prev = [], [0.0] * 5
            for k in range(ns):
                step(first_timed + k)
                ts = r.counters()                            # waits for the frame; stage times accumulate
                per_frame.append([ts.stageMs[i] - prev[i] for i in range(5)])
                prev = [ts.stageMs[i] for i in range(5)]
            pf = np.array(per_frame)
            serial_ms = [float(x) for x in np.median(pf, axis=0)]
            out["ms_per_frame_serial"] = round(float(np.median(pf.sum(axis=1))), 4)
            out["ms_per_frame_serial_frames"] = ns
            r.set_overlap(2 if os.environ.get("RESTIR_OVERLAP") is None else int(os.environ["RESTIR_OVERLAP"]))
        out["frame_latency_ms"] = round(frame_latency_ms, 4)
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
        kname = {0: "k_direct_stage", 1: "k_indirect_stage", 2: "k_denoise_lds<false, true>", 3: "k_denoise_lds<true, true>", 4: "k_compose"}[dom]
        out["roofline"] = {"bound": None, "kernel": STAGE_NAMES[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                           "algorithmic_bytes_per_launch": round(b_screen + b_trav), "screen_bytes": round(b_screen), "traversal_bytes": round(b_trav),
                           "launch_ms": round(dur_ms, 4), "stage_ms_per_frame": {STAGE_NAMES[i]: round(stage_ms[i], 4) for i in range(5)},
                           "note": "achieved / frac: ALGORITHMIC bytes (per-lane node / triangle / hit / candidate touches x SURVEY 8(d) sizes + screen traffic) "
                                   "over the launch time in the timed region; most of those touches are served by L1 / L2 — `hbm_counter` is what crossed the fabric"}
        # HBM traffic and VALU instructions of the same kernel from the PMC passes of THIS workload (scripts/pmc.sh: separate rocprofv3 --pmc runs of this
        # command; FETCH_SIZE / WRITE_SIZE are in KiB; gfx950's FETCH_SIZE counts wide reads at half their size — MI355X_MICROARCH.md, HBM section — so it is
        # doubled; WRITE_SIZE is taken as reported).  Counters of another build or another workload are not used: the label is then null.
        t, pmc = pmc_entry(workload_key(args.config, orbit, footprint, args.pose), kname)
        frames_note = None
        if t and orbit:
            # a moving camera renders other work every frame: counters describe this line only if they were collected over the same frame indices
            # (scripts/pmc.sh records them: `_frames`; entries of round 4 have none)
            wf = ((pmc.get("workloads", {}).get(workload_key(args.config, orbit, footprint, args.pose)) or {}).get("_frames") or {})
            if wf.get("warmup") != args.warmup or wf.get("steps") != args.steps:
                frames_note = f"the counter pass of this workload covers frames {wf.get('warmup')}..+{wf.get('steps')}, this line times frames {args.warmup}..+{args.steps} of a moving camera: counters withheld"
                t = None
        out["roofline"]["traffic_lib"] = pmc.get("_lib_sha256_16") if pmc else None
        out["roofline"]["traffic_stale"] = bool(pmc) and pmc.get("_lib_sha256_16") != lib_sha256_16()
        sdur = None
        if serial_ms is not None:  # same kernel, same bytes, launched alone (no other frame's kernels beside it)
            sdur = serial_ms[dom] / launches
            out["roofline"]["serial"] = {"launch_ms": round(sdur, 4), "achieved": round((b_screen + b_trav) / (sdur * 1e-3) / 1e9, 2),
                                         "frac": round((b_screen + b_trav) / (sdur * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                         "stage_ms_per_frame": {STAGE_NAMES[i]: round(serial_ms[i], 4) for i in range(5)},
                                         "frame_ms": round(sum(serial_ms), 4)}
        # What the kernel is actually close to: VALU instruction issue.  The ceiling is MEASURED in this run on this device
        # (rt_measure_valu_peak, csrc/microbench.hip: a chain-free loop of the kernels' instruction mix, 8 waves per SIMD), the
        # kernel's wave-level instruction count comes from the same PMC passes (SQ_INSTS_VALU per launch).
        peak_mix = None
        try:
            peak_mix = r.measure_valu_peak(0, 8)
            peak_fma = r.measure_valu_peak(1, 8)
            valu = {"peak_wave_inst_per_s": round(peak_mix), "peak_fma_only_wave_inst_per_s": round(peak_fma),
                    "peak_source": "rt_measure_valu_peak in this run (csrc/microbench.hip): kernel instruction mix / v_fma_f32 only, chain-free, 8 waves per SIMD",
                    "fma_only_tflops": round(peak_fma * 128 / 1e12, 1)}
            if t and t.get("INSTS_VALU"):
                rate = t["INSTS_VALU"] / (dur_ms * 1e-3)
                valu.update({"wave_insts_per_launch": round(t["INSTS_VALU"]), "achieved_wave_inst_per_s": round(rate), "frac": round(rate / peak_mix, 4),
                             "lane_utilisation": round(t["THREAD_CYCLES_VALU"] / (64.0 * t["INSTS_VALU"]), 4),
                             "source": "profiles/pmc_traffic.json (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU per launch)"})
                if sdur:
                    valu["serial_frac"] = round(t["INSTS_VALU"] / (sdur * 1e-3) / peak_mix, 4)
                w = (pmc.get("workloads", {}).get(workload_key(args.config, orbit, footprint, args.pose)) or pmc) if pmc else None
                fr = w.get("_frame") if w else None
                if fr and args.emulate_world <= 1:
                    valu.update({"frame_wave_insts": round(fr["INSTS_VALU"]), "frame_frac": round(fr["INSTS_VALU"] / (out["ms_per_step"] * 1e-3) / peak_mix, 4)})
            out["roofline"]["valu"] = valu
        except Exception as e:  # the headline number must not depend on this extra pass
            out["roofline"]["valu"] = {"error": repr(e)}
        label_bound(out["roofline"], t, dur_ms, peak_mix, b_screen + b_trav)
        if frames_note:
            out["roofline"]["bound_evidence"] = frames_note
        ff = (out["roofline"].get("valu") or {}).get("frame_frac")
        if ff is not None and ff > 1.0:
            # more instructions per second than the chip can issue: the counted instructions are not the timed work (round-4 verdict, weak 6) — no label
            out["roofline"]["bound"] = None
            out["roofline"]["bound_evidence"] = f"valu.frame_frac = {ff} > 1: the counter pass does not describe the timed frames; label withheld"
        if t and sdur:
            out["roofline"]["hbm_counter"]["serial_frac"] = round(out["roofline"]["traffic"] / (sdur * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            if t.get("WRITE_SIZE_KB_full_lds_stack"):
                out["roofline"]["hbm_counter"]["write_bytes"] = round(t["WRITE_SIZE_KB"] * 1024)
                out["roofline"]["hbm_counter"]["write_bytes_with_whole_stack_in_lds"] = round(t["WRITE_SIZE_KB_full_lds_stack"] * 1024)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(abi, host, scene, env, st, desc, W, H, first_timed)
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
            out["roofline"] = {"bound": None, "bound_evidence": "no counter pass exists for a row band; a band-sized launch is bound by the dependent steps of its slowest rays (DESIGN.md 7)",
                               "kernel": "direct_stage (this rank's band, rows %d..%d, launched alone)" % (y0, y1), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes_per_launch": round(b_screen + b_trav),
                               "screen_bytes": round(b_screen), "traversal_bytes": round(b_trav), "launch_ms": round(dur_ms, 4)}
        except Exception as e:  # the headline number must not depend on this extra pass
            out["roofline"] = {"bound": None, "error": repr(e)}
    if world > 1 and frame is not None:
        # ---- what the first real multi-device run should tell (it runs once, on a node nobody sees): per-rank bands and times, the peer-access matrix, and the
        #      tiled == untiled gate (restir_amd/verify.py): a short sequence rendered twice from a cold history, tiled on the timed partition and untiled on rank 0
        per_rank = comm.all_gather_floats([float(frame.y0), float(frame.y1), float(last_band_ms), elapsed_local / args.steps * 1e3, float(frame.history_fallbacks)])
        ver = None
        if args.verify_frames > 0 and not args.profile_run:
            ver = verify_rccl(args, abi, tiled, torch, Renderer, comm, Frame, r, list(frame.part), scene, (eye0, center0, up0, fov0), st, desc, W, H, orbit, rank, local_rank)
        if rank == 0 and out is not None:
            out["rccl_ranks"] = world
            out["rccl"] = "ok" if comm.nccl else f"not used: RESTIR_DIST_BACKEND={os.environ.get('RESTIR_DIST_BACKEND')} (test hook)"
            if share:
                out["n_gpus"] = min(world, torch.cuda.device_count())
                out["note"] = f"{world} ranks on {out['n_gpus']} device(s) (RESTIR_BENCH_SHARE_DEVICE): a functional check of the one-process-per-GPU host, NOT a benchmark result"
            out["peer_access"] = peer_matrix(torch, world)
            # where this rank's streams sit in the process's creation order (round-5 finding: that order is worth 15-75 % of a frame): the three context streams were
            # created before the process group (rt_get_streams), nothing in the timed path takes a stream from torch's pool
            out["stream_layout"] = dict(r.stream_layout(), torch_pool_streams_in_timed_path=0, created_before_process_group=True, levels=r.stream_priorities()["chosen"],
                                        note="library-created streams only; RCCL's internal streams are created at init_process_group, after the context's")
            out["rank_report"] = [{"rank": q, "rows": [int(v[0]), int(v[1])], "traced_stages_alone_ms": (round(v[2], 4) if v[2] >= 0 else None),
                                   "ms_per_step_local": round(v[3], 4), "history_fallbacks": int(v[4])} for q, v in enumerate(per_rank)]
            attach_verdict(out, ver)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and not args.single_host and not args.profile_run and os.environ.get("RESTIR_BENCH_BOTH", "1") != "0":
            out = both_hosts(args, out)     # the driver never passes --native: time that host too (child process, bounded by the time left) and report it beside this one
        if world > 1:
            out["wall_s"] = round(_time_mod.time() - _T_START, 1)   # this process, start to line (scene generation, BVH build, band planning, timed region, gate, second host)
        print(json.dumps(out), flush=True)
        if world > 1 and out.get("tiled_equals_untiled") is False:
            raise SystemExit(3)             # a number for a wrong image is not a result


def via_gltf(args, abi, host, Renderer, scene, env, st, W, H):
    """--via-gltf: write the procedural scene as a glTF asset, read it back through Scene::load, hold the loaded scene to the procedural one (digest of every array
    rt_upload_scene reads; reference frame digests rendered here from the procedural scene, compared after the run with the loaded scene's).  Returns (report, loaded scene,
    its description): the bench line then times the LOADED scene."""
    import resource
    import shutil
    import tempfile
    from restir_amd import verify as V
    d = tempfile.mkdtemp(prefix="restir_gltf_")
    path = os.path.join(d, "scene.gltf")
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    t0 = time.time()
    if not scene.saveGltf(path):
        raise SystemExit("--via-gltf: Scene::saveGltf failed")
    save_s = time.time() - t0
    files = os.listdir(d)
    disk = sum(os.path.getsize(os.path.join(d, f)) for f in files)
    ref_digest = host.scene_digest(scene.desc(env))
    # reference frames from the procedural scene: three frames, cold history, the run's first camera
    pose = scene.cameraPose()
    scene.updateCamera(W, H)
    cams = V.verify_cameras(scene, W, H, pose, False, 3)
    r0 = Renderer().setup(0); r0.load_scene(scene.desc(env)); r0.update(W, H)
    st0 = type(st).from_buffer_copy(st)
    V.render_untiled(r0.run, r0.set_camera, cams, st0); r0.sync()
    frame_ref = V.digests(r0.readback, 0)
    r0.destroy()
    stat_ref = scene.getStat()
    loaded = host.Scene()
    t0 = time.time()
    if not loaded.load(path):
        raise SystemExit("--via-gltf: Scene::load failed")
    load_s = time.time() - t0
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    shutil.rmtree(d, ignore_errors=True)
    loaded.setCamera(*pose)      # (a glTF camera node stores a matrix, not eye / centre of interest: the pose is carried over so that the frames can be compared bit for bit)
    desc = loaded.desc(env)
    dig = host.scene_digest(desc)
    rep = {"path": "Scene::saveGltf -> .gltf + .bin + one PNG per image -> Scene::load (host/gltf_loader.cpp)", "files": len(files), "bytes_on_disk": disk,
           "save_s": round(save_s, 2), "load_s": round(load_s, 2), "peak_rss_gb_before_save": round(rss0, 2), "peak_rss_gb_after_load": round(rss1, 2),
           "stats_equal": loaded.getStat() == stat_ref, "scene_digest_equal": dig == ref_digest, "scene_digest": {"procedural": ref_digest["all"], "loaded": dig["all"]},
           "parts_differing": [k for k in ref_digest if ref_digest[k] != dig[k]], "_frame_digests": frame_ref, "_cams": cams}
    return rep, loaded, desc


def peer_matrix(torch, n):
    """torch.cuda.can_device_access_peer for the first n devices (1 on the diagonal)"""
    k = min(n, torch.cuda.device_count())
    return [[1 if a == b else int(torch.cuda.can_device_access_peer(a, b)) for b in range(k)] for a in range(k)]


def attach_verdict(out, ver):
    """top-level verdict fields of the tiled == untiled gate"""
    if not ver:
        out["tiled_equals_untiled"] = None
        out["verify"] = {"skipped": "--verify-frames 0 or --profile-run"}
        return
    out["verify"] = ver
    out["tiled_equals_untiled"] = bool(ver["workload"]["equal"])
    if "moving_camera" in ver:     # informational: the history halos / the exact fallback under camera motion (the timed workload is the static camera)
        out["tiled_equals_untiled_moving_camera"] = bool(ver["moving_camera"]["equal"])


def verify_rccl(args, abi, tiled, torch, Renderer, comm, Frame, r, part, scene, pose, st, desc, W, H, orbit, rank, local_rank):
    """COLLECTIVE (every rank calls it, outside the timed region).  For the workload's camera path — and, when that is static, also for the orbiting camera, which
    exercises the history halos — render --verify-frames frames from a cold history row-tiled on the timed partition, gather every distributed buffer to rank 0,
    render the same frames untiled on rank 0's device (a second context) and compare digests of all six buffers of the last frame."""
    from restir_amd import verify as V
    res = {}
    modes = [("workload", orbit)] + ([] if orbit else [("moving_camera", True)])
    for name, orb in modes:
        cams = V.verify_cameras(scene, W, H, pose, orb, args.verify_frames)
        torch.cuda.synchronize(); comm.barrier()
        r.update(W, H)                                   # re-allocates and clears every screen buffer: cold history on every rank
        be = tiled.RendererTensors(r)
        fr, cur = V.render_tiled(Frame, be, comm, W, H, part, cams, st, r.set_camera)
        torch.cuda.synchronize()
        if rank == 0:
            td = V.digests(r.readback, cur)
            ref = Renderer().setup(local_rank); ref.load_scene(desc); ref.update(W, H)
            V.render_untiled(ref.run, ref.set_camera, cams, st); ref.sync()
            ud = V.digests(ref.readback, cur); ref.destroy()
            res[name] = dict(V.compare(td, ud), frames=len(cams), partition=list(part), history_fallbacks=int(fr.history_fallbacks),
                             camera="orbiting 0.5 deg / frame" if orb else "static")
        comm.barrier()
    return res


def verify_native(args, abi, m, r, bands, scene, pose, st, W, H, orbit):
    """the same gate for the native context: m = MultiGpuRenderer (its readback assembles a buffer from the ranks that own its rows), r = a single-device Renderer"""
    from restir_amd import verify as V
    res = {}
    part = [b[0] for b in bands] + [bands[-1][1]]
    modes = [("workload", orbit)] + ([] if orbit else [("moving_camera", True)])
    for name, orb in modes:
        cams = V.verify_cameras(scene, W, H, pose, orb, args.verify_frames)
        m.sync(); m.update(W, H); m.set_bands(part)     # cold history; rt_mgpu_resize falls back to equal bands: the timed partition again (frozen)
        fb0 = int(m.stats().historyFallbacks)
        cur = V.render_untiled(m.run, m.set_camera, cams, st); m.sync()
        td = V.digests(m.readback, cur)
        r.set_stream(0)                                  # (the roofline pass above bound the context to a torch stream: back to its own)
        r.sync(); r.update(W, H)
        V.render_untiled(r.run, r.set_camera, cams, st); r.sync()
        ud = V.digests(r.readback, cur)
        res[name] = dict(V.compare(td, ud), frames=len(cams), partition=part, history_fallbacks=int(m.stats().historyFallbacks) - fb0,
                         camera="orbiting 0.5 deg / frame" if orb else "static")
    return res


def both_hosts(args, rccl_line):
    """rank 0, after the RCCL host's run and the end of its process group: the native host (one process, hipMemcpyPeerAsync pulls, no host sync per stage) on the
    same devices in a CHILD process (this one has RCCL and N idle peers' contexts around), both results in one line — `value` stays the RCCL host's (the host the driver launched) as long as its tiled == untiled
    gate passed; `faster_host` names the faster verified one."""
    import subprocess
    keep = ["--gpus", str(args.gpus), "--native", "--steps", str(args.steps), "--warmup", str(args.warmup), "--config", str(args.config), "--scene-footprint", args.scene_footprint,
            "--scale", str(args.scale), "--verify-frames", str(args.verify_frames), "--band-rounds", str(args.band_rounds), "--diffuse-rounds", str(args.diffuse_rounds),
            "--period-rounds", str(args.period_rounds)]
    keep += (["--moving-camera"] if args.moving_camera else []) + (["--equal-bands"] if args.equal_bands else [])
    keep += (["--width", str(args.width)] if args.width else []) + (["--height", str(args.height)] if args.height else [])
    keep += (["--devices", args.devices] if getattr(args, "devices", "") else [])
    env = {k: v for k, v in os.environ.items() if not (k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
                                                               "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS", "RESTIR_CPUS_SHARED") or k.startswith(("TORCHELASTIC_", "NCCL_ASYNC", "TORCH_NCCL"))
                                                          or (k == "RESTIR_CPUS" and os.environ.get("RESTIR_CPUS_SHARED") == "1"))}
    nat, err = None, None
    try:
        time.sleep(2.0)     # the other ranks are leaving: let their contexts go before the native host times anything
        # bounded by what is left of the driver's limit for the whole command (1 800 s, measured from this process's start), with a margin for the line itself
        budget = float(os.environ.get("RESTIR_BENCH_WALL_LIMIT", "1500")) - (_time_mod.time() - _T_START)
        if budget < 90:
            raise TimeoutError(f"second host skipped: {budget:.0f} s left of the command's wall budget")
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + keep, env=env, capture_output=True, text=True, timeout=budget)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if lines:
            nat = json.loads(lines[-1])
        if p.returncode != 0 or nat is None:
            err = f"exit {p.returncode}: {(p.stderr or '').strip()[-400:]}"
    except Exception as e:   # noqa: BLE001 — the RCCL line must survive whatever the child does
        err = repr(e)

    def summary(d):
        return None if d is None else {"value": d.get("value"), "ms_per_step": d.get("ms_per_step"), "tiled_equals_untiled": d.get("tiled_equals_untiled"),
                                       "tiled_equals_untiled_moving_camera": d.get("tiled_equals_untiled_moving_camera"), "history_fallbacks": d.get("history_fallbacks")}
    hosts = {"rccl": summary(rccl_line), "native": summary(nat)}
    if err:
        hosts["native_error"] = err
    # `value` is tied to ONE named host — the one-process-per-GPU RCCL host the driver launched — so that the headline does not switch implementation between runs
    # (advisor finding of round 5); the native host's numbers stand beside it under `hosts`, and `faster_host` names the faster verified one.  Only when the RCCL line has
    # no verified value does the native one take its place (the line says so: `host`).
    ok = {k: d for k, d in (("rccl", rccl_line), ("native", nat)) if d is not None and d.get("value") and d.get("tiled_equals_untiled") is not False}
    name = "rccl" if ("rccl" in ok or not ok) else "native"
    out = dict(ok.get(name, rccl_line)); out["host"] = name
    out["faster_host"] = max(ok, key=lambda k: ok[k]["value"]) if ok else None
    out["hosts"] = hosts
    out["rccl_ranks"] = rccl_line.get("rccl_ranks"); out["rccl"] = rccl_line.get("rccl", "ok")
    out.setdefault("peer_access", rccl_line.get("peer_access"))
    if out.get("rank_report") is None:      # (the native line won: the one-process-per-GPU host's per-rank report and digests stay in the line)
        out["rank_report"] = rccl_line.get("rank_report")
        out["hosts"]["rccl"]["verify"] = rccl_line.get("verify")
    out["hosts_all_verified"] = all(h is not None and h.get("tiled_equals_untiled") is True for h in (hosts["rccl"], hosts["native"]))
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks the way the driver does (torch.distributed.run, one rank per GPU, RCCL) and pass
    rank 0's JSON line through."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def native_world(args, abi, host, Renderer, torch, extra=None):
    """N > 1 through the native context: ONE process, one worker thread + rt_ctx + three streams per device, frames in flight per rank, every exchange a
    hipMemcpyPeerAsync pull ordered by events (csrc/mgpu.cpp) — the host round 3 optimised, on N REAL devices.  Same metric and JSON as the RCCL host."""
    from restir_amd.renderer import MultiGpuRenderer, LINK_GROUPS
    n = args.gpus
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(n))
    if len(devices) != n or any(d < 0 or d >= torch.cuda.device_count() for d in devices):
        raise SystemExit(f"--devices {args.devices}: needs {n} indices below {torch.cuda.device_count()}")
    distinct = len(set(devices))
    cfg = CONFIGS[args.config]
    W, H = args.width or cfg["size"][0], args.height or cfg["size"][1]
    orbit = args.moving_camera or cfg.get("orbit", False)
    if cfg.get("di_only"):
        raise SystemExit("--native: config 2 is the direct stage alone (one launch per step); it has no row-tiled form")
    footprint = footprint_of(args)
    scene = host.Scene().makeProcedural(scene_kind(abi, args.config, footprint), args.scale, 1)
    apply_pose(args, scene)
    env = None
    if cfg["env"]:
        env = host.HdrSampling(); env.makeSyntheticSky(cfg["env"][0], cfg["env"][1], 5e4, 7)
    st = host.default_state(W, H, scene, env)
    for k, v in cfg.get("state", {}).items():
        setattr(st, k, v)
    desc = scene.desc(env)
    m = MultiGpuRenderer().setup(devices)
    t0 = time.time(); m.load_scene(desc); build_s = time.time() - t0
    m.update(W, H)
    m.set_balance(0 if args.equal_bands else 1)
    eye0, center0, up0, fov0 = scene.cameraPose()
    scene.updateCamera(W, H)

    def camera(f):
        st.time = 1000 + f
        if orbit:
            a = np.deg2rad(0.5 * (f + 1))
            rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=np.float32)
            scene.setCamera(center0 + rot @ (eye0 - center0), center0, up0, fov0)
        scene.updateCamera(W, H)
        return scene.getCamera()

    f = 0
    for _ in range(args.warmup + 8 * 2):      # the balancer moves a boundary by at most two stripes per frame: give it the frames the RCCL host's planning rounds get
        m.set_camera(camera(f)); m.run(st, f); f += 1
    m.sync()
    if not args.equal_bands:
        m.set_balance(2)                       # the partition is then fixed, like the RCCL host's: nothing of the planning runs in the timed region
    for _ in range(4):
        m.set_camera(camera(f)); m.run(st, f); f += 1
    m.sync()
    first_timed = f
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.set_camera(camera(f)); m.run(st, f); f += 1
    m.sync()
    elapsed = time.perf_counter() - t0
    stats, links = m.stats(), m.link_stats()
    bands = [(int(stats.bandBegin[r]), int(stats.bandEnd[r])) for r in range(n)]
    # rays of the same frames: the tiled frame is bit-identical to the single-GPU one, so its rays are counted there (instrumented kernels, device 0, untimed)
    r = Renderer().setup(0); r.load_scene(desc); r.update(W, H)
    n_count = min(4, args.steps)
    for k in range(2):
        r.set_camera(camera(first_timed - 2 + k)); r.run(st, first_timed - 2 + k)
    r.set_counting(True)
    for k in range(n_count):
        r.set_camera(camera(first_timed + k)); r.run(st, first_timed + k)
    r.sync(); cnt = r.counters(); r.set_counting(False)
    rays_per_frame = float(cnt.closestHitRays + cnt.anyHitRays) / n_count
    ms_per_step = elapsed / args.steps * 1e3
    out = {"metric": "Mrays/s (ClosestHit+AnyHit ray queries per second) of the " + ("1080p ReSTIR DI+GI+denoise frame" if args.config == 4 else f"config-{args.config} frame") + "; ms_per_step = ms/frame",
           "value": round(rays_per_frame * args.steps / elapsed / 1e6, 2), "unit": "Mrays/s", "n_gpus": distinct, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"config {args.config}: {cfg['name']} procedural scene ({footprint} footprint: {scene.getStat()['materials']} materials, {texture_bytes(desc) / 1e6:.0f} MB of BGRA8 texels), "
                                  f"{scene.getStat()['instancedTriangles']} triangles, {W}x{H}, {cfg['pipeline']}, "
                                  + ("camera orbiting 0.5 deg / frame" if orbit else "static camera") + (f", pose {args.pose}: {POSES[args.pose]}" if getattr(args, "pose", 0) else ""),
                      "baseline_config": args.config, "scene_footprint": footprint, "texture_bytes": texture_bytes(desc), "width": W, "height": H, "scene_scale": args.scale,
                      "parallelism": f"row-tiled x{n}: ONE process, native context (rt_mgpu_*): a worker thread, an rt_ctx and three streams per device, frames in flight per rank, "
                                     f"event-ordered hipMemcpyPeerAsync pulls; " + ("equal-height bands" if args.equal_bands else f"cost-weighted bands {bands}"),
                      "rays_per_frame": round(rays_per_frame), "fps": round(1e3 / ms_per_step, 2), "bvh8_build_s": round(build_s, 2)},
           "ranks": n, "history_fallbacks": int(stats.historyFallbacks),
           "halo_bytes_per_rank": [int(sum(stats.haloBytesRankKind[q][k] for k in range(4))) for q in range(n)]}
    # the links, measured: every rank's four pull groups of the last timed frame, HIP events on the stream that carried them
    lk = {"devices": [int(links.devices[q]) for q in range(n)], "peer_access": [[int(links.peerAccess[a][b]) for b in range(n)] for a in range(n)], "groups": list(LINK_GROUPS),
          "pull_ms": [[round(float(links.pullMs[q][g]), 4) for g in range(4)] for q in range(n)], "pull_bytes": [[int(links.pullBytes[q][g]) for g in range(4)] for q in range(n)]}
    tot_b = sum(sum(x) for x in lk["pull_bytes"]); tot_ms = sum(sum(x) for x in lk["pull_ms"])
    lk["GBps_while_pulling"] = round(tot_b / max(1e-9, tot_ms * 1e-3) / 1e9, 2) if tot_ms > 0 else None
    lk["slowest_rank_pull_ms"] = round(max(sum(x) for x in lk["pull_ms"]), 4)
    out["links"] = lk
    if distinct < n:
        out["note"] = f"{n} ranks on {distinct} device(s): a functional check of the native host, NOT a benchmark result"
    out["host"] = "native"
    out["stream_layout"] = dict(m.stream_layout(), note="every rank renders on its context's streams: main created at rt_mgpu_create, filter then indirect stream on the rank's first "
                                                        "frame in flight; rank 0 adds a copy stream for the gather (csrc/mgpu.cpp ensurePipeStreams)")
    out["peer_access_runtime"] = peer_matrix(torch, torch.cuda.device_count())
    if extra:
        out.update(extra)
    # roofline of the dominant kernel of rank 0's band, launched alone on device 0 (same per-unit figures as the N = 1 line)
    try:
        y0, y1 = bands[0]
        fdom = first_timed + min(2, args.steps - 1)
        r.set_camera(camera(fdom))
        cur = torch.cuda.Stream(); r.set_stream(cur.cuda_stream)
        r.run_stage(st, fdom, abi.STAGE_DIRECT, 0, y0, y1); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(5):
            r.run_stage(st, fdom, abi.STAGE_DIRECT, 0, y0, y1)
        e1.record(cur); torch.cuda.synchronize()
        dur_ms = e0.elapsed_time(e1) / 5
        r.set_counting(True); r.run_stage(st, fdom, abi.STAGE_DIRECT, 0, y0, y1); r.sync()
        cb = r.counters(); r.set_counting(False)
        b_screen = SCREEN_BYTES[0] * W * (y1 - y0)
        b_trav = cb.nodesVisited * NODE_B + cb.trisTested * TRI_B + cb.hitsShaded * HIT_B + cb.risCandidates * RIS_B
        ach = (b_screen + b_trav) / (dur_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": None, "bound_note": "no counter pass exists for a row band: a band-sized launch is bound by the dependent steps of its slowest rays (DESIGN.md 7), neither by bytes nor by issue slots",
                           "kernel": "direct_stage (rank 0's band, rows %d..%d, launched alone)" % (y0, y1), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes_per_launch": round(b_screen + b_trav),
                           "screen_bytes": round(b_screen), "traversal_bytes": round(b_trav), "launch_ms": round(dur_ms, 4)}
    except Exception as e:  # the headline number must not depend on this extra pass
        out["roofline"] = {"bound": None, "error": repr(e)}
    # the tiled == untiled gate (restir_amd/verify.py), outside the timed region, on the partition that was timed
    ver = None
    if args.verify_frames > 0:
        try:
            ver = verify_native(args, abi, m, r, bands, scene, (eye0, center0, up0, fov0), st, W, H, orbit)
        except Exception as e:   # noqa: BLE001
            ver = {"workload": {"equal": False, "error": repr(e)}}
    attach_verdict(out, ver)
    out["wall_s"] = round(_time_mod.time() - _T_START, 1)
    print(json.dumps(out), flush=True)
    m.destroy(); r.destroy()
    if out.get("tiled_equals_untiled") is False:
        raise SystemExit(3)                 # a number for a wrong image is not a result
    return None


# xGMI model for the emulation (the link time of a pull is not measurable on one device): one link per neighbour, 153.6 GB/s bidirectional per link
# (task brief / AMD MI355X platform figure) = 76.8 GB/s per direction, of which a DMA copy reaches ~85 %; plus a fixed cost per hipMemcpyPeerAsync
XGMI_GBS_PER_DIRECTION = 76.8 * 0.85
XGMI_COPY_LATENCY_US = 8.0
HALO_KINDS = ["history", "filter", "spatial", "moved", "gather", "fallback"]


def emulate_world(args, abi, host, scene, env, st, desc, single, W, H, device):
    """Per-rank compute of the N-way row-tiled frame on ONE GPU with the native multi-GPU context (every rank on this device).
      serial   the ranks take turns (rt_mgpu_set_serialize): per-rank traced + filter time of a GPU to itself, barrier schedule — the critical path of ONE frame
      period   frames in flight (rt_mgpu's default schedule), one rank at a time (rt_mgpu_set_solo): wall clock per frame of that rank — the rate an N-GPU node sustains
    Exchanges run as device-to-device copies: their volume is reported per purpose and priced with the xGMI model above.  NOT a benchmark result — the
    driver measures real scaling; this is the load-balance / critical-path instrument DESIGN.md §7 quotes."""
    from restir_amd.renderer import MultiGpuRenderer
    from restir_amd.tiled import diffuse_bands
    n = args.emulate_world
    single.update(W, H)
    scene.updateCamera(W, H)
    orbit = args.moving_camera or CONFIGS[args.config].get("orbit", False)
    eye0, center0, up0, fov0 = scene.cameraPose()

    def pose(sc, f):   # SURVEY 8(d) config 5 / --moving-camera: the camera orbits its centre of interest, 0.5 degrees per frame (same path as the N = 1 line)
        if orbit:
            a = np.deg2rad(0.5 * (f + 1))
            rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=np.float32)
            sc.setCamera(center0 + rot @ (eye0 - center0), center0, up0, fov0)

    def cam(f):
        st.time = 1000 + f
        pose(scene, f)
        scene.updateCamera(W, H)
        return scene.getCamera()
    # single-GPU references on the same frames: serial sum of kernels and the frames-in-flight rate
    single.set_overlap(0)
    for f in range(args.warmup):
        single.set_camera(cam(f)); single.run(st, f)
    single.sync(); single.set_counting(False)
    for f in range(args.warmup, args.warmup + args.steps):
        single.set_camera(cam(f)); single.run(st, f)
    c = single.counters()
    one = sum(c.stageMs[i] for i in range(5)) / max(1, c.framesTimed)
    single.set_overlap(2)
    f0 = args.warmup + args.steps
    for f in range(f0, f0 + 8):
        single.set_camera(cam(f)); single.run(st, f)
    single.sync(); t0 = time.perf_counter()
    for f in range(f0 + 8, f0 + 8 + args.steps):
        single.set_camera(cam(f)); single.run(st, f)
    single.sync(); one_flight = (time.perf_counter() - t0) / args.steps * 1e3
    single.destroy()
    m = MultiGpuRenderer().setup([device] * n)
    m.load_scene(desc); m.update(W, H)
    m.set_serialize(True); m.set_balance(0 if args.equal_bands else 1)
    scene2 = host.Scene().makeProcedural(scene_kind(abi, args.config, footprint_of(args)), args.scale, 1)   # (a second instance: its camera history is the tiled run's own)
    apply_pose(args, scene2)
    scene2.updateCamera(W, H)
    acc = np.zeros((n, 2)); halo = 0; kinds = np.zeros(6); bands = None
    f = 0
    def frame():
        nonlocal f
        st.time = 1000 + f
        pose(scene2, f)
        scene2.updateCamera(W, H)
        m.set_camera(scene2.getCamera()); m.run(st, f)
        f += 1
    for k in range(args.warmup + args.steps):
        frame()
        if k >= args.warmup:
            s = m.stats()
            acc += np.array([[s.tracedMs[r], s.filterMs[r]] for r in range(n)])
            halo += s.haloBytes; kinds += np.array([s.haloBytesKind[i] for i in range(6)])
            bands = [(s.bandBegin[r], s.bandEnd[r]) for r in range(n)]
    acc /= args.steps
    tot = acc.sum(axis=1)
    kinds /= args.steps
    out = {"metric": "per-rank ms/frame of the N-way row-tiled frame, ranks emulated one at a time on ONE GPU (not a benchmark result)",
           "n_ranks": n, "steps": args.steps, "warmup": args.warmup, "balance": "equal bands" if args.equal_bands else "cost-weighted bands",
           "size": [W, H], "single_gpu_serial_ms": round(one, 4), "single_gpu_frames_in_flight_ms": round(one_flight, 4),
           "rank_ms": [round(float(x), 4) for x in tot], "rank_traced_ms": [round(float(x), 4) for x in acc[:, 0]],
           "rank_filter_ms": [round(float(x), 4) for x in acc[:, 1]], "bands_last_frame": bands, "slowest_rank_ms": round(float(tot.max()), 4),
           "max_over_min": round(float(tot.max() / max(1e-6, tot.min())), 3), "projected_speedup_compute_only": round(one / float(tot.max()), 2),
           "halo_bytes_per_frame": int(halo / args.steps), "halo_bytes_per_frame_by_kind": {HALO_KINDS[i]: int(kinds[i]) for i in range(6)},
           "fallbacks": m.stats().historyFallbacks}
    # ---- frames in flight per rank: the PERIOD of every rank alone on the GPU, on the partition found above -------------------------------------------
    if not args.no_period:
        m.set_balance(2); m.set_serialize(False); m.set_pipeline(True)
        if os.environ.get("RESTIR_EMULATE_GATHER") == "0":
            m.set_gather(False)
        only = [int(x) for x in os.environ["RESTIR_EMULATE_RANKS"].split(",")] if os.environ.get("RESTIR_EMULATE_RANKS") else range(n)

        def solo_period(r, k):
            m.set_solo(r)
            for _ in range(10):
                frame()
            m.sync(); t0 = time.perf_counter()
            for _ in range(k):
                frame()
            m.sync(); return (time.perf_counter() - t0) / k * 1e3

        # The partition above equalises the SUM of a rank's stage times; with frames in flight a rank's period is closer to the longest of them.  A few
        # rounds of {measure every rank's period alone, spread it over the rank's 16-row stripes, re-plan} equalise the periods instead.
        cur = [b[0] for b in bands] + [bands[-1][1]]
        stripes = (H + 15) // 16
        cost = np.ones(stripes)
        history = []
        for rnd in range(0 if os.environ.get("RESTIR_EMULATE_RANKS") else args.period_rounds):
            per = [solo_period(r, max(40, args.steps)) for r in range(n)]
            history.append({"bands": cur, "period_ms": [round(float(x), 3) for x in per]})
            for r in range(n):
                a, b = cur[r] // 16, (cur[r + 1] + 15) // 16
                cost[a:b] = 0.5 * cost[a:b] + 0.5 * per[r] / max(1, b - a) if rnd else per[r] / max(1, b - a)
            cur = MultiGpuRenderer.plan_bands(H, n, cost, cur, 4)
            m.set_solo(-1); m.set_bands(cur)
            for _ in range(3):
                frame()      # the rows that moved travel with the history pulls
        # ... then the boundaries diffuse (restir_amd/tiled.py diffuse_bands = the rule of csrc/mgpu.cpp rebalance()): one stripe towards the slower of two neighbours whose
        # periods differ by > 6 %, the slowest rank's boundaries first
        for rnd in range(0 if os.environ.get("RESTIR_EMULATE_RANKS") else 2 * args.period_rounds):
            per = [solo_period(r, max(40, args.steps)) for r in range(n)]
            history.append({"bands": cur, "period_ms": [round(float(x), 3) for x in per]})
            nxt = diffuse_bands(cur, per)
            if nxt == cur:
                break
            cur = nxt
            m.set_solo(-1); m.set_bands(cur)
            for _ in range(3):
                frame()
        if history:
            # the re-planning is a noisy fixed-point iteration: measure the last plan as well and keep the best partition seen
            per = [solo_period(r, max(40, args.steps)) for r in range(n)]
            history.append({"bands": cur, "period_ms": [round(float(x), 3) for x in per]})
            best = min(history, key=lambda h: max(h["period_ms"]))
            if best["bands"] != cur:
                cur = best["bands"]
                m.set_solo(-1); m.set_bands(cur)
                for _ in range(3):
                    frame()
            out["period_balancing_rounds"] = history
            out["bands_period_balanced"] = [(cur[r], cur[r + 1]) for r in range(n)]
        periods, pk = [], np.zeros(6)
        for r in only:
            periods.append(solo_period(r, max(60, args.steps)))
            s = m.stats()
            mine = np.array([s.haloBytesRankKind[r][i] for i in range(6)], dtype=np.float64)
            if mine[:4].sum() >= pk[:4].sum():
                pk = mine
            if r == 0:
                out["rank0_bytes_by_kind"] = {HALO_KINDS[i]: int(mine[i]) for i in range(6)}
            out.setdefault("rank_bytes_steady", []).append(int(mine[:4].sum()))
        m.set_solo(-1)
        out["rank_period_ms"] = [round(float(x), 4) for x in periods]
        out["slowest_rank_period_ms"] = round(float(max(periods)), 4)
        out["projected_speedup_period_vs_single_gpu_frames_in_flight"] = round(one_flight / float(max(periods)), 2)
        out["projected_speedup_period_vs_single_gpu_serial"] = round(one / float(max(periods)), 2)
        # modelled xGMI time of the steady-state pulls of the busiest rank (history before its direct stage, filter halos before its filters; two
        # neighbour links used one after the other on the rank's stream: the pessimistic end)
        worst = {k: float(pk[i]) for i, k in enumerate(HALO_KINDS)}
        out["xgmi_model"] = {"GBs_per_direction": XGMI_GBS_PER_DIRECTION, "copy_latency_us": XGMI_COPY_LATENCY_US,
                              "worst_rank_bytes_by_kind": {k: int(v) for k, v in worst.items()},
                              "modelled_ms": {k: round(v / (XGMI_GBS_PER_DIRECTION * 1e9) * 1e3 + (XGMI_COPY_LATENCY_US * 1e-3 * (6 if v > 0 else 0)), 4) for k, v in worst.items()}}
        mm = out["xgmi_model"]["modelled_ms"]
        out["xgmi_model"]["on_frame_path_ms"] = round(mm["history"] + mm["filter"] + mm["moved"], 4)
        out["slowest_rank_period_with_modelled_xgmi_ms"] = round(float(max(periods)) + out["xgmi_model"]["on_frame_path_ms"], 4)
    out["stream_layout_in_process"] = m.stream_layout()
    m.destroy()
    if args.solo_fresh > 0 and not args.no_period and "rank_period_ms" in out and len(out["rank_period_ms"]) == n:
        out["solo_fresh"] = solo_fresh(args, n, cur, out["rank_period_ms"])
    print(json.dumps(out), flush=True)
    return None


def solo_fresh(args, n, part, periods):
    """the K slowest ranks of the in-process period pass, each measured again in a fresh child process (emulate_child): same partition, same frames, but the process holds the
    N contexts' main streams and ONLY that rank's filter / indirect stream (+ rank 0's copy stream) — the in-process figure was taken with every earlier rank's streams alive"""
    import subprocess
    order = sorted(range(n), key=lambda q: -periods[q])[:args.solo_fresh]
    res = []
    for q in order:
        cmd = [sys.executable, os.path.abspath(__file__), "--emulate-world", str(n), "--config", str(args.config), "--scene-footprint", args.scene_footprint, "--scale", str(args.scale),
               "--steps", str(args.steps), "--warmup", str(args.warmup), "--emulate-child", f"{q}:" + ",".join(str(int(b)) for b in part)]
        cmd += (["--moving-camera"] if args.moving_camera else []) + (["--width", str(args.width)] if args.width else []) + (["--height", str(args.height)] if args.height else [])
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            d = json.loads(lines[-1]) if lines else {"error": f"exit {p.returncode}: {(p.stderr or '').strip()[-300:]}"}
        except Exception as e:   # noqa: BLE001
            d = {"error": repr(e)}
        d["rank"] = q; d["period_in_process_ms"] = periods[q]
        res.append(d)
    return res


def emulate_child(args, abi, host, scene, env, st, desc, single, W, H, device):
    """--emulate-child rank:bands — one rank's period in a process of its own (see solo_fresh)"""
    from restir_amd.renderer import MultiGpuRenderer
    single.destroy()                                    # (its one stream was created first and stays counted: `library_streams_created`)
    n = args.emulate_world
    q, bands = args.emulate_child.split(":")
    q, part = int(q), [int(x) for x in bands.split(",")]
    orbit = args.moving_camera or CONFIGS[args.config].get("orbit", False)
    eye0, center0, up0, fov0 = scene.cameraPose()
    m = MultiGpuRenderer().setup([device] * n)
    m.load_scene(desc); m.update(W, H)
    m.set_bands(part); m.set_serialize(False); m.set_pipeline(True)     # (an explicit partition implies "freeze")
    if os.environ.get("RESTIR_EMULATE_GATHER") == "0":
        m.set_gather(False)
    scene.updateCamera(W, H)
    f = 0

    def frame():
        nonlocal f
        st.time = 1000 + f
        if orbit:
            a = np.deg2rad(0.5 * (f + 1))
            rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=np.float32)
            scene.setCamera(center0 + rot @ (eye0 - center0), center0, up0, fov0)
        scene.updateCamera(W, H)
        m.set_camera(scene.getCamera()); m.run(st, f)
        f += 1
    m.set_solo(q)
    for _ in range(max(20, args.warmup)):
        frame()
    per = []
    for _rep in range(2):
        m.sync(); t0 = time.perf_counter()
        k = max(60, args.steps)
        for _ in range(k):
            frame()
        m.sync(); per.append((time.perf_counter() - t0) / k * 1e3)
    out = {"period_fresh_process_ms": round(min(per), 4), "passes_ms": [round(x, 4) for x in per], "stream_layout": m.stream_layout(), "mgpu_prio": os.environ.get("RESTIR_MGPU_PRIO", "default")}
    print(json.dumps(out), flush=True)
    m.destroy()
    return None


def cpu_quota():
    """CPUs the container's cgroup quota allows (cgroup v2 cpu.max / v1 cfs quota, rounded up), or None — include/rt_cpus.h is the C side of this"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, -(-int(q) // int(p)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return max(1, -(-q // p)) if q > 0 and p > 0 else None
    except (OSError, ValueError):
        return None


def cpu_baseline(abi, host, scene, env, st, desc, W, H, frame0, di_only=False):
    """The CPU oracle (naive binary BVH + scalar C++, std::thread workers) on bounded samples of the SAME frame: rows around the image centre, cold temporal history, the full
    12-dispatch frame (config 2: its direct stage), every point the BEST of its passes (a pass can only be slowed by whatever else the host runs), the host's load average
    beside it.
    Round 6 found what had shaped this leg since round 4 (profiles/r06_cpu_baseline.txt).  (1) The box's container may use SIXTEEN of the machine's 256 hardware threads
    (cgroup cpu.max): 16-19 CPUs were busy whatever the thread count, so "cores: 128" described the request, not the resource.  (2) The oracle pinned a worker through its
    handle after creating it, and doing that to a worker that had already exited pinned the CALLER: from some dispatch on, every worker of the 128- and 256-thread points ran on
    one CPU (fixed in oracle/orc_stages.h).  Now: `cores` = the CPUs the process can keep busy (affinity mask and cgroup quota), thread counts 1 / cores / 2 x cores / 4 x cores,
    workers pinned only where there is no quota (a pinned worker cannot move off a CPU another tenant is using, and under a quota the scheduler's choice measured 40 % better);
    `value` is the best point."""
    from oracle.binding import Oracle
    o = Oracle(0)
    o.upload_scene(desc)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cpu_quota()
    cores = max(1, min(ncpu, quota) if quota else ncpu)
    pin = quota is None or quota >= ncpu
    load0 = os.getloadavg()
    st.time = 1000 + frame0

    def band(y0, y1):
        h0, h1 = y0 // 2, y1 // 2
        o.resize(W, H); o.set_camera(scene.getCamera())          # (re-allocates the screen buffers: cold history every pass)
        o.reset_counters()
        t0, c0 = time.perf_counter(), time.process_time()
        o.run_stage(st, frame0, abi.STAGE_DIRECT, 0, y0, y1)
        if not di_only:
            o.run_stage(st, frame0, abi.STAGE_INDIRECT, 0, h0, h1)
            for l in range(4):
                o.run_stage(st, frame0, abi.STAGE_DENOISE_DIRECT, l, y0, y1)
            for l in range(5):
                o.run_stage(st, frame0, abi.STAGE_DENOISE_INDIRECT, l, h0, h1)
            o.run_stage(st, frame0, abi.STAGE_COMPOSE, 0, y0, y1)
        dt, cpu = time.perf_counter() - t0, time.process_time() - c0
        c = o.counters()
        return c.closestHitRays + c.anyHitRays, dt, cpu

    mid = (H // 2 // 16) * 16

    def point(threads, rows, passes, budget_s):
        """best of up to `passes` passes (at least 2, until budget_s is spent) over `rows` rows around the image centre"""
        o.set_threads(threads, pin=pin)
        y0, y1 = max(0, mid - rows // 2), min(H, mid + rows // 2)
        got, spent = [], 0.0
        while len(got) < passes and (len(got) < 2 or spent < budget_s):
            rays, dt, cpu = band(y0, y1)
            got.append((rays / dt, rays, dt, cpu)); spent += dt
        best = max(got)
        return {"threads": threads, "rows": [y0, y1], "mrays_s": round(best[0] / 1e6, 4), "per_thread_mrays_s": round(best[0] / 1e6 / threads, 5), "passes": len(got),
                "spread": [round(min(g[0] for g in got) / 1e6, 4), round(best[0] / 1e6, 4)], "rays": int(best[1]), "seconds_best": round(best[2], 3),
                "cpus_busy": round(best[3] / max(best[2], 1e-9), 1), "wall_s_all_passes": round(spent, 1)}

    o.set_threads(cores, pin=pin)
    band(max(0, mid - 16), min(H, mid + 16))                      # warm-up (page cache, BVH in the host caches), not reported
    # Every point but the single thread runs the SAME band (comparable rays), >= 4 rows per thread of the largest count (the row scheduler is dynamic: with one row per thread a
    # pass lasts as long as its slowest row); the single thread takes 8 rows of its middle — the horizon rows, the dearest rays of the frame: its rate is a lower bound, which is
    # why no "parallel efficiency" is derived from it any more.
    counts = sorted({t for t in (cores, 2 * cores, 4 * cores) if t <= max(cores, 1024)})
    rows_all = min(H - H % 16, max(16, (4 * counts[-1] + 15) // 16 * 16, 512 if not di_only else 64))
    pts = [point(1, 8, 2, 0.0)] + [point(t, rows_all, 3, 3.0) for t in counts]
    allp = max(pts[1:], key=lambda q: q["mrays_s"])
    # the best thread count once more at the end of the leg (the neighbours' load moves within seconds): the better of its two visits is the point
    again = point(allp["threads"], allp["rows"][1] - allp["rows"][0], 4, 4.0)
    if again["mrays_s"] > allp["mrays_s"]:
        again["passes"] += allp["passes"]; again["spread"] = [min(again["spread"][0], allp["spread"][0]), again["spread"][1]]
        pts[pts.index(allp)] = again; allp = again
    else:
        allp["passes"] += again["passes"]; allp["spread"] = [min(again["spread"][0], allp["spread"][0]), allp["spread"][1]]
    load1 = os.getloadavg()
    y0, y1 = allp["rows"]
    return {"value": allp["mrays_s"], "unit": "Mrays/s", "cores": min(cores, allp["threads"]), "host_cpus": ncpu, "cpu_quota": quota, "threads_used": allp["threads"],
            "cpus_busy": allp["cpus_busy"], "per_core": round(allp["mrays_s"] / max(1, min(cores, allp["threads"])), 5), "kind": "port",
            "single_thread": pts[0]["mrays_s"], "single_thread_rows": pts[0]["rows"], "pinned": pin, "statistic": "best pass of each point; value = the best point", "scaling": pts,
            "host_loadavg_before_after": [round(load0[0], 2), round(load1[0], 2)],
            "sample": f"rows {y0}..{y1} of one {W}x{H} frame ({'direct stage' if di_only else 'all 12 dispatches'}, cold temporal history) on {allp['threads']} threads"
                      f"{' (pinned)' if pin else ''} of a process that may use {cores} of the host's {ncpu} CPUs ({'cgroup quota' if quota and quota < ncpu else 'affinity mask'}; "
                      f"{allp['cpus_busy']} busy on average): best of {allp['passes']} passes ({allp['spread'][0]}..{allp['spread'][1]} Mrays/s), {allp['rays']} rays in "
                      f"{allp['seconds_best']:.2f} s => {allp['seconds_best'] * H / max(1, y1 - y0) * 1e3:.0f} ms/frame extrapolated; the other thread counts "
                      f"({', '.join(str(q['threads']) for q in pts)}) in `scaling`"}


if __name__ == "__main__":
    main()
