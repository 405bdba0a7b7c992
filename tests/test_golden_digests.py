"""Committed golden fixtures (tests/golden/stage_digests.json, minted by tests/golden/make_stage_digests.py from the oracle):
SHA-256 of every screen-space buffer of small seeded frames.  The CPU test holds the oracle to them, the GPU test holds the HIP
path to them WITHOUT running the oracle — so a change made to both sides at once (numerics contract, RNG order, a stage's
control flow) cannot pass unnoticed, and the parity of the HIP path is checked against data that travels with the repository."""
import importlib.util
import json
import os
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_stage_digests", os.path.join(HERE, "golden", "make_stage_digests.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)
GOLD = json.load(open(os.path.join(HERE, "golden", "stage_digests.json")))


def _check(case, got):
    want = GOLD[case]
    assert got["scene"] == want["scene"], "procedural scene generation drifted (not a renderer mismatch): regenerate the fixture knowingly"
    assert len(got["frames"]) == len(want["frames"])
    for f, (g, w) in enumerate(zip(got["frames"], want["frames"])):
        assert g.keys() == w.keys()
        bad = [k for k in w if g[k] != w[k]]
        assert not bad, f"{case} frame {f}: buffers differ from the golden digests: {bad}"


def test_fixture_covers_the_cases():
    assert set(GOLD) == set(gen.CASES)
    assert all(len(v["frames"]) >= 3 for v in GOLD.values())


@pytest.mark.parametrize("case", sorted(gen.CASES))
def test_oracle_matches_golden_digests(case):
    _check(case, gen.run_case(gen.oracle_factory, case))


def _hip_factory(latency):
    def make(desc, W, H):
        from helpers import RendererBackend
        from restir_amd.renderer import Renderer
        r = Renderer().setup(0); r.load_scene(desc); r.update(W, H); r.set_traversal(2 if latency else 1)
        return RendererBackend(r)
    return make


@pytest.mark.gpu
@pytest.mark.parametrize("latency", [False, True], ids=["throughput", "latency"])
@pytest.mark.parametrize("case", sorted(gen.CASES))
def test_hip_path_matches_golden_digests(case, latency):
    _check(case, gen.run_case(_hip_factory(latency), case))
