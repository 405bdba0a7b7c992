"""Known-answer vectors minted from the REFERENCE's own sources (oracle/kat/mint_kat.sh -> tests/golden/kat_reference.json):
random.glsl tea/pcg/rand, compress.glsl (C++ branch), common.glsl OffsetRay/hash8bit, src/alias_table.hpp.
They pin the oracle's (and the host library's) integer / bit-exact pieces to the reference."""
import ctypes as C
import json
import os
import numpy as np
from oracle.binding import lib
from restir_amd import host

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_reference.json")))
f32 = lambda bits: np.array(bits, dtype=np.uint32).view(np.float32)  # noqa: E731


def test_tea():
    for a, b, want in KAT["tea"]:
        assert lib().orc_tea(a, b) == want
    assert lib().orc_tea(1920 * 540 + 960, 12345) == KAT["tea_1920x540_960_t12345"] == 0x6230029F  # SURVEY.md §8c probe


def test_pcg_and_rand():
    for s0, s1, out in KAT["pcg"]:
        s = C.c_uint32(s0)
        assert lib().orc_pcg(C.byref(s)) == out and s.value == s1
    for s0, s1, bits in KAT["rand"]:
        s = C.c_uint32(s0)
        r = np.float32(lib().orc_rand(C.byref(s)))
        assert r.view(np.uint32) == bits and s.value == s1 and 0.0 <= r < 1.0


def test_compress_unit_vec():
    for vx, vy, vz, packed, dx, dy, dz in KAT["compress_unit_vec"]:
        v = f32([vx, vy, vz])
        assert lib().orc_compress_unit_vec(float(v[0]), float(v[1]), float(v[2])) == packed
        # host-side packer used by Scene::createVertexBuffer must agree bit for bit as well
        out = np.zeros(3, dtype=np.float32)
        lib().orc_decompress_unit_vec(packed, out.ctypes.data)
        # decompress ends in normalize(): GLSL builtin in the reference, v*(1/sqrt) here => compare within 2 ulp
        assert np.allclose(out, f32([dx, dy, dz]), rtol=3e-7, atol=3e-7)
        assert abs(float(np.dot(out.astype(np.float64), v.astype(np.float64))) - 1.0) < 2e-4  # 16+16-bit oct precision


def test_pack_unorm4x8():
    for x, y, z, w, want in KAT["pack_unorm4x8"]:
        v = f32([x, y, z, w])
        assert lib().orc_pack_unorm4x8(*[float(t) for t in v]) == want


def test_hash8bit_and_offset_ray():
    for a, want in KAT["hash8bit"]:
        assert lib().orc_hash8bit(a) == want
    for row in KAT["offset_ray"]:
        p, n, want = f32(row[0:3]), f32(row[3:6]), np.array(row[6:9], dtype=np.uint32)
        out = np.zeros(3, dtype=np.float32)
        lib().orc_offset_ray(p.ctypes.data, n.ctypes.data, out.ctypes.data)
        assert np.array_equal(out.view(np.uint32), want)


def test_alias_table_matches_reference_header():
    for case in KAT["alias_table"]:
        w = f32(case["w"])
        n = w.size
        prob = np.zeros(n, dtype=np.float32)
        fail = np.zeros(n, dtype=np.int32)
        host.host_lib().rth_alias_table(n, w.ctypes.data, prob.ctypes.data, fail.ctypes.data)
        assert np.array_equal(prob.view(np.uint32), np.array(case["prob"], dtype=np.uint32))
        assert np.array_equal(fail, np.array(case["fail"], dtype=np.int32))
        # marginals: P(i) = (prob_i + sum_{j: fail_j = i} (1 - prob_j)) / n == w_i / sum(w)
        marg = prob.astype(np.float64).copy()
        for j in range(n):
            if fail[j] != j:
                marg[fail[j]] += 1.0 - prob[j]
            # self-aliased buckets keep their whole column
            else:
                marg[j] += 1.0 - prob[j]
        assert np.allclose(marg / n, w.astype(np.float64) / w.astype(np.float64).sum(), atol=2e-5)
