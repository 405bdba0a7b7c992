"""bench.py's multi-GPU command line (round-3 verdict, item 3): `--gpus N` never reports an N-GPU number from fewer devices; `--native` drives the one-process
context and prints the same JSON line plus the measured link statistics."""
import json
import os
import subprocess
import sys
import pytest
from helpers import ROOT

pytestmark = pytest.mark.gpu


def _run(*args, timeout=600):
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_gpus_n_refuses_fewer_devices():
    import torch
    have = torch.cuda.device_count()
    p = _run("--gpus", str(have + 1), "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert p.returncode != 0
    assert "device(s) visible" in p.stderr and '"n_gpus"' not in p.stdout
    p = _run("--gpus", str(have + 1), "--native", "--steps", "2", "--warmup", "1")
    assert p.returncode != 0 and '"n_gpus"' not in p.stdout


def test_native_host_line():
    """two ranks of the native context (both on device 0 when the box has one GPU: a functional check), config 3 at a small size"""
    import torch
    devs = "0,1" if torch.cuda.device_count() >= 2 else "0,0"
    p = _run("--gpus", "2", "--native", "--devices", devs, "--config", "3", "--scale", "0.05", "--width", "480", "--height", "272", "--steps", "6", "--warmup", "3")
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["ranks"] == 2 and d["n_gpus"] == len(set(devs.split(","))) and d["value"] > 0 and d["ms_per_step"] > 0
    lk = d["links"]
    assert lk["groups"] == ["history", "history_indirect", "filter_direct", "filter_indirect"] and len(lk["pull_ms"]) == 2
    assert all(sum(b) > 0 for b in lk["pull_bytes"]) and all(sum(t) > 0 for t in lk["pull_ms"])      # every rank pulled, and the pulls were timed
    assert lk["peer_access"][0][0] == 1 and d["halo_bytes_per_rank"][0] > 0
    assert ("note" in d) == (d["n_gpus"] < 2)


SMALL = ["--config", "3", "--scale", "0.05", "--width", "480", "--height", "272", "--steps", "6", "--warmup", "3"]


def test_native_host_gate_passes_and_catches_a_corrupted_halo():
    """round-4 verdict, item 2: the N > 1 bench checks what it times.  After the timed region the tiled frame sequence is compared with the untiled one on
    digests of all six frame buffers (restir_amd/verify.py); a deliberately damaged filter halo (RESTIR_TEST_CORRUPT_HALO, csrc/mgpu.cpp) must fail the gate
    and the process."""
    import torch
    devs = "0,1" if torch.cuda.device_count() >= 2 else "0,0"
    p = _run("--gpus", "2", "--native", "--devices", devs, *SMALL)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["tiled_equals_untiled"] is True and d["tiled_equals_untiled_moving_camera"] is True and d["host"] == "native"
    v = d["verify"]["workload"]
    assert v["frames"] == 3 and set(v["buffers"]) == {"gbuffer0", "direct_resv0", "light_id0", "indirect_resv0", "direct_result0", "indirect_result0"}
    assert all(b["equal"] and b["tiled"] == b["untiled"] and len(b["tiled"]) == 16 for b in v["buffers"].values())
    # the hook lives in a TEST build of the library only (-DRT_TEST_HOOKS=1, built here: measurement builds do not travel to the GPU box): the product library ignores the variable
    from restir_amd import build as b
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env["RESTIR_TEST_CORRUPT_HALO"] = "1"
    q0 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--native", "--devices", devs] + SMALL, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert q0.returncode == 0 and json.loads(q0.stdout.strip().splitlines()[-1])["tiled_equals_untiled"] is True
    env["RESTIR_HIP_LIB"] = b.build_hip(variant="testhooks", extra_flags=["-DRT_TEST_HOOKS=1"])
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--native", "--devices", devs] + SMALL, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert q.returncode == 3, (q.returncode, q.stderr[-1500:])
    e = json.loads(q.stdout.strip().splitlines()[-1])
    assert e["tiled_equals_untiled"] is False
    bad = [k for k, b in e["verify"]["workload"]["buffers"].items() if not b["equal"]]
    assert "direct_result0" in bad and "gbuffer0" not in bad        # the damaged rows are noisy direct colour: the filtered image differs, the traced buffers do not


def test_rccl_host_line_carries_the_gate():
    """the RCCL host the driver launches (one process per GPU); world 2 when the box has two devices, else world 1 is refused by --gpus and this test runs the
    ranks through torch.distributed.run itself on the devices there are (NCCL world 1: the gate's collectives, the digests and the JSON fields)."""
    import socket
    import torch
    n = 2 if torch.cuda.device_count() >= 2 else 1
    if n == 1:
        pytest.skip("one device: the RCCL gate needs world > 1 (covered over gloo on the CPU, tests/test_tiled_gloo.py::test_verify_gate)")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + SMALL, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["tiled_equals_untiled"] is True and d["rccl_ranks"] == n and d["host"] in ("rccl", "native")
    assert d["hosts"]["rccl"]["tiled_equals_untiled"] is True and d["hosts"]["native"]["tiled_equals_untiled"] is True and d["hosts_all_verified"] is True


def test_one_process_per_gpu_code_path_on_shared_device():
    """Everything `bench.py --gpus 2` executes as the driver launches it (torch.distributed.run, one process per rank: band planning, timed region, rank report, the gate on
    the RendererTensors backend, then the native host in a child process and both hosts in one line) on the devices there are: two ranks share device 0 through the
    RESTIR_BENCH_SHARE_DEVICE hook, gloo carries the exchanges (RCCL refuses two ranks on one device).  The first real multi-device run must not be the first run of this code."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", RESTIR_BENCH_SHARE_DEVICE="1", RESTIR_DIST_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--band-rounds", "2", "--diffuse-rounds", "2"] + SMALL, capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["tiled_equals_untiled"] is True and d["rccl_ranks"] == 2 and d["host"] in ("rccl", "native") and "note" in d
    h = d["hosts"]
    assert h["rccl"]["tiled_equals_untiled"] is True and h["rccl"]["tiled_equals_untiled_moving_camera"] is True and h["rccl"]["value"] > 0
    assert h["native"] is not None and h["native"]["tiled_equals_untiled"] is True and d["hosts_all_verified"] is True


def test_via_gltf_the_loaded_scene_is_the_procedural_one():
    """round-5 verdict, missing 4: the reference's entry point is Scene::load on a glTF file (src/scene.cpp:57-125).  `bench.py --via-gltf` writes the workload's scene as
    an asset (.gltf + .bin + one PNG per image), reads it back through Scene::load, times the LOADED scene and holds it to the procedural one: same digest of every array
    rt_upload_scene reads, same six frame buffers after three cold-history frames.  Here at scale 0.05 (the full-size run is profiles/r06z_bench_via_gltf.json)."""
    os.environ["RESTIR_GLTF_EXTERNAL"] = "1"      # (a scene this small would be written self-contained: force the asset form — .gltf + .bin + one PNG per image)
    try:
        p = _run("--via-gltf", "--scale", "0.05", "--width", "640", "--height", "368", "--steps", "4", "--warmup", "4", "--no-cpu-baseline")
    finally:
        os.environ.pop("RESTIR_GLTF_EXTERNAL", None)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    v = d["via_gltf"]
    assert v["stats_equal"] and v["scene_digest_equal"] and v["parts_differing"] == [] and v["frames_equal"] is True, v
    assert v["files"] >= 3 and v["bytes_on_disk"] > 1e6 and v["load_s"] > 0 and d["value"] > 0
    assert v["frame_digests"]["procedural"] == v["frame_digests"]["loaded"] and len(v["frame_digests"]["loaded"]) == 6


def test_pose_changes_the_view_not_the_scene():
    """`--pose N`: the headline scene from the fixed poses of round 6 (0 = the scene's camera); the key of its counter pass carries the pose"""
    p = _run("--print-workload-key", "--pose", "2")
    assert p.returncode == 0 and p.stdout.strip() == "config4_real_pose2"
    a = json.loads(_run("--scale", "0.05", "--width", "480", "--height", "272", "--steps", "3", "--warmup", "4", "--no-cpu-baseline").stdout.strip().splitlines()[-1])
    b = json.loads(_run("--pose", "2", "--scale", "0.05", "--width", "480", "--height", "272", "--steps", "3", "--warmup", "4", "--no-cpu-baseline").stdout.strip().splitlines()[-1])
    assert ", pose 2:" in b["config"]["workload"] and ", pose " not in a["config"]["workload"]      # ("compose" is part of every workload string)
    assert a["config"]["rays_per_frame"] != b["config"]["rays_per_frame"] and a["config"]["accel"] == b["config"]["accel"]
