"""include/rt_detmath.h: the deterministic transcendentals stay within a few ulp of the real functions."""
import ctypes as C
import numpy as np
from oracle.binding import lib

OPS = {"exp": 0, "log": 1, "pow": 2, "sin": 3, "cos": 4, "asin": 5, "acos": 6, "atan2": 7, "tan": 8, "div_uniform": 9, "exp_nonpos": 10, "ftoi": 11, "ftou": 12, "unorm8": 13}


def det(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float32)
    out = np.empty_like(a)
    lib().orc_detmath(OPS[op], a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / np.maximum(ulp, 1e-45)


def test_exp_log():
    x = np.linspace(-87, 88, 400001, dtype=np.float32)
    assert ulp_err(det("exp", x), np.exp(x.astype(np.float64))).max() <= 2.0
    x = np.geomspace(1e-30, 1e30, 400001).astype(np.float32)
    assert ulp_err(det("log", x), np.log(x.astype(np.float64))).max() <= 2.0
    assert det("exp", [-200.0])[0] == 0.0 and np.isinf(det("exp", [100.0])[0]) and np.isnan(det("exp", [np.nan])[0])
    assert det("log", [0.0])[0] == -np.inf and np.isnan(det("log", [-1.0])[0])


def test_trig():
    x = np.linspace(-7, 7, 400001, dtype=np.float32)
    assert ulp_err(det("sin", x), np.sin(x.astype(np.float64)))[np.abs(np.sin(x)) > 1e-3].max() <= 4.0
    assert ulp_err(det("cos", x), np.cos(x.astype(np.float64)))[np.abs(np.cos(x)) > 1e-3].max() <= 4.0
    assert np.abs(det("sin", x) - np.sin(x.astype(np.float64))).max() < 2e-7
    x = np.linspace(-1, 1, 400001, dtype=np.float32)
    assert ulp_err(det("asin", x), np.arcsin(x.astype(np.float64)))[np.abs(x) > 1e-3].max() <= 4.0
    assert ulp_err(det("acos", x), np.arccos(x.astype(np.float64)))[x < 0.999].max() <= 4.0
    y = np.random.default_rng(1).uniform(-5, 5, 200000).astype(np.float32)
    xx = np.random.default_rng(2).uniform(-5, 5, 200000).astype(np.float32)
    ref = np.arctan2(y.astype(np.float64), xx.astype(np.float64))
    assert np.abs(det("atan2", y, xx) - ref).max() < 1e-6
    assert det("atan2", [0.0], [0.0])[0] == 0.0


def test_pow_srgb():
    x = np.linspace(0, 1, 65537, dtype=np.float32)
    got = det("pow", x, np.full_like(x, 2.2))
    ref = np.power(x.astype(np.float64), np.float64(np.float32(2.2)))
    rel = np.abs(got[1:] - ref[1:]) / ref[1:]
    assert rel.max() < 4e-6 and got[0] == 0.0


def test_uniform_divisor_shortcut_is_ieee_division():
    """csrc/stages.hip divUniform (A-Trous weights: distance / sigma): q' = fma(a - b*q, 1/b, q) equals RN(a/b) for every tested
    numerator, for divisors in the range launchStage accepts (1e-6 .. 1e6)."""
    rng = np.random.default_rng(5)
    sig = np.concatenate([[0.4, 0.1, 0.02, 4.0, 1.0, 1e-6, 1e6, 0.3333333, 0.99999994, 1.9999999], np.exp(rng.uniform(np.log(1e-6), np.log(1e6), 90))]).astype(np.float32)
    for b in sig:
        a = np.concatenate([np.exp(rng.uniform(np.log(2.0 ** -25) + np.log(b), 40.0, 60000)), rng.random(20000), [0.0, b, 3 * b, b / 3]]).astype(np.float32)
        got = det("div_uniform", a, np.full_like(a, b))
        want = a / b                                    # numpy float32 division = IEEE
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (b, int((got != want).sum()))
    # below 2^-25 * sigma the quotient may differ in the last bit (subnormal residual) but exp(-q) is exactly 1 for both
    a = (np.float32(0.4) * np.exp(rng.uniform(-80, np.log(2.0 ** -25), 20000))).astype(np.float32)
    q1 = det("div_uniform", a, np.full_like(a, 0.4)); q2 = a / np.float32(0.4)
    assert (det("exp", -q1) == 1.0).all() and (det("exp", -q2) == 1.0).all()
    # above 1e30 (where a * (1/b) could overflow for the smallest divisor) and for non-finite numerators the plain division runs
    big = np.concatenate([np.exp(rng.uniform(np.log(1e29), np.log(3.4e38), 5000)), [1e30, 1.0000001e30, 3.4028235e38, np.inf]]).astype(np.float32)
    for b in (1e-6, 0.4, 1e6):
        got = det("div_uniform", big, np.full_like(big, b))
        with np.errstate(over="ignore"):
            want = big / np.float32(b)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.isnan(det("div_uniform", [np.nan], [0.4])[0])


def test_branch_free_exp_equals_rt_exp_on_nonpositive_arguments():
    x = -np.concatenate([np.geomspace(1e-30, 200.0, 300001), np.linspace(0, 90, 200001), [0.0, np.inf]]).astype(np.float32)
    a, b = det("exp_nonpos", x), det("exp", x)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.isnan(det("exp_nonpos", [np.nan])[0])


def test_tan_is_sin_over_cos():
    x = np.linspace(-1.5, 1.5, 100001, dtype=np.float32)
    assert ulp_err(det("tan", x), np.tan(x.astype(np.float64))).max() <= 8.0


def test_float_to_int_conversions_follow_their_specification():
    """rt_ftoi / rt_ftou (include/rt_detmath.h) are written with clamps + a select (no branches on the GPU); they must be the
    function the contract states: NaN -> 0; saturation at 2147483520 / INT32_MIN resp. 0 / 4294967040; truncation otherwise.
    (The header's implementation was also compared with the early-return form over all 2^32 bit patterns when it was written.)"""
    rng = np.random.default_rng(9)
    bits = np.concatenate([rng.integers(0, 2 ** 32, 4_000_000, dtype=np.uint64).astype(np.uint32),
                           np.array([0, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 0xffc00000, 0x7f800001, 0xff800001, 0x4effffff, 0x4f000000, 0xcf000000, 0xcf000001,
                                     0x4f7fffff, 0x4f800000, 0x3f7fffff, 0xbf7fffff, 0x3f800000, 0xbf800000], dtype=np.uint32)])
    x = bits.view(np.float32)
    with np.errstate(invalid="ignore"):
        xi = np.where(np.isnan(x), 0.0, np.clip(np.trunc(x.astype(np.float64)), -2147483648.0, 2147483520.0))
        xu = np.where(np.isnan(x), 0.0, np.clip(np.trunc(x.astype(np.float64)), 0.0, 4294967040.0))
    assert np.array_equal(det("ftoi", x).astype(np.float64), xi)
    assert np.array_equal(det("ftou", x).astype(np.float64), xu)


def test_unorm8_shortcut_is_the_division_by_255():
    """csrc/dev_math.h unorm8ToFloat (texel and G-buffer channel decode): 3 fused steps instead of an IEEE division, equal to
    b / 255.0f for all 256 inputs."""
    b = np.arange(256, dtype=np.float32)
    assert np.array_equal(det("unorm8", b).view(np.uint32), (b / np.float32(255.0)).view(np.uint32))
