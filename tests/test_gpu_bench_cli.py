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
