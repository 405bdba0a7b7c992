"""Known-answer vectors for the FLOAT helpers of the hot path, minted from the REFERENCE's own GLSL compiled on a small
vector shim (oracle/kat/kat_float.cpp + build_ref.sh -> tests/golden/kat_reference_float.json): common.glsl
toConcentricDisk / powerHeuristic / HDRToLDR / LDRToHDR, denoise_common.glsl luminance, all of reservoir.glsl and all of
pbr_metallicworkflow.glsl, all of sun_and_sky.glsl, and the display pass's helpers (random.glsl pcg3d, tonemapping.glsl
Uncharted-2 chain, post.frag dither / toneExposure).

Tolerances.  The reservoir arithmetic, powerHeuristic, luminance and the LDR maps use + - * / and compares only: BIT-EXACT.
Everything that goes through sqrt/sin/cos depends on whose libm evaluated it (the shim used glibc, the GLSL driver would
use its own, the oracle uses include/rt_detmath.h), so those vectors are compared with a tolerance:
  * toConcentricDisk, GetSphericalUv, CreateCoordinateSystem: 2 ulp of 1.0 absolute (2.4e-7)
  * sampled direction: 2e-6 absolute per component
  * pdf / f of the sampled direction: 2e-4 relative (GTR2 at low roughness amplifies the direction's last-ulp differences)
  * Eval's f and pdf for a given wi: 2e-4 relative (same amplification: cos^2*(a^2-1)+1 cancels near the lobe's peak)
  * toneMapUncharted (ends in pow): 2.4e-7 absolute on a [0,1] value; dither() lands on the same 1/255 step exactly
  * sun_and_sky(): 2e-4 relative (acos of a near-1 dot product at the sun disk's edge amplifies last-ulp differences)
The branch decisions (diffuse vs specular lobe, InvalidPdf) must agree exactly."""
import ctypes as C
import json
import os
import numpy as np
from oracle.binding import lib

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_reference_float.json")))
f32 = lambda bits: np.array(bits, dtype=np.uint32).view(np.float32)  # noqa: E731
bits = lambda v: int(np.float32(v).view(np.uint32))  # noqa: E731


def test_concentric_disk():
    for rx, ry, dx, dy in KAT["concentric_disk"]:
        r, out = f32([rx, ry]), np.zeros(2, np.float32)
        lib().orc_concentric_disk(float(r[0]), float(r[1]), out.ctypes.data)
        assert np.abs(out - f32([dx, dy])).max() <= 2.4e-7


def test_spherical_uv_and_coordinate_system():
    for row in KAT["spherical_uv_coord_system"]:                     # common.glsl:68-92
        v = f32(row)
        d, uv, tb = v[0:3].copy(), np.zeros(2, np.float32), np.zeros(6, np.float32)
        lib().orc_spherical_uv(d.ctypes.data, uv.ctypes.data)
        lib().orc_coordinate_system(d.ctypes.data, tb.ctypes.data)
        assert np.abs(uv - v[3:5]).max() <= 2.4e-7 and np.abs(tb - v[5:11]).max() <= 2.4e-7


def test_power_heuristic_luminance_ldr_bit_exact():
    for f, g, want in KAT["power_heuristic"]:
        a = f32([f, g])
        assert bits(lib().orc_power_heuristic(float(a[0]), float(a[1]))) == want
    for row in KAT["luminance_ldr"]:
        c = f32(row[0:3]).copy()
        assert bits(lib().orc_luminance(c.ctypes.data)) == row[3]
        ldr, hdr = np.zeros(3, np.float32), np.zeros(3, np.float32)
        lib().orc_hdr_to_ldr(c.ctypes.data, ldr.ctypes.data)
        assert list(ldr.view(np.uint32)) == row[4:7]
        lib().orc_ldr_to_hdr(ldr.ctypes.data, hdr.ctypes.data)
        assert list(hdr.view(np.uint32)) == row[7:10]


def _call(fn, m, n, wo, x):
    m, n, wo, x = (np.ascontiguousarray(v, dtype=np.float32) for v in (m, n, wo, x))
    out = np.zeros(8, dtype=np.float32)
    fn(m.ctypes.data, n.ctypes.data, wo.ctypes.data, x.ctypes.data, out.ctypes.data)
    return out


def _rel(a, b):
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-6)).max())


def test_metallic_workflow_bsdf_matches_reference_glsl():
    rows = KAT["bsdf"]
    assert len(rows) == 400
    invalid = valid = 0
    for row in rows:
        v = f32(row)
        m = np.array([*v[0:3], v[3], v[4]], np.float32)            # albedo, metallic, roughness
        n, wo, r, rdir, rpdf, rf, wi, ref_f, ref_pdf = v[5:8], v[8:11], v[11:14], v[14:17], v[17], v[18:21], v[21:24], v[24:27], v[27]
        s = _call(lib().orc_bsdf_sample, m, n, wo, r)
        if rpdf < 0:                                               # InvalidPdf: sampled direction under the surface
            assert s[3] == rpdf == -1.0
            invalid += 1
        else:
            valid += 1
            assert np.abs(s[0:3] - rdir).max() <= 2e-6
            assert _rel(s[3:4], np.array([rpdf])) <= 2e-4
            assert _rel(s[4:7], rf) <= 2e-4
        e = _call(lib().orc_bsdf_eval, m, n, wo, wi)
        assert _rel(e[0:3], ref_f) <= 2e-4 and _rel(e[3:4], np.array([ref_pdf])) <= 2e-4
        if not ref_f.any():                                        # wi under the surface: exactly zero on both sides
            assert not e[0:3].any()
    assert valid > 300 and invalid > 10                            # both outcomes are covered


class _LS(C.Structure):
    _fields_ = [("Li", C.c_float * 3), ("wi", C.c_float * 3), ("dist", C.c_float)]


class _DR(C.Structure):
    _fields_ = [("ls", _LS), ("num", C.c_uint32), ("weight", C.c_float)]


class _GS(C.Structure):
    _fields_ = [("L", C.c_float * 3), ("xv", C.c_float * 3), ("nv", C.c_float * 3), ("xs", C.c_float * 3), ("ns", C.c_float * 3), ("pHat", C.c_float)]


class _IR(C.Structure):
    _fields_ = [("gs", _GS), ("num", C.c_uint32), ("weight", C.c_float), ("bigW", C.c_float)]


def test_reservoir_op_stream_bit_exact():
    """reservoir.glsl: update / merge / clamp / checkValidity / reset over 600 random ops (with negative and NaN weights):
    num, weight, the selected sample and resvInvalid() after every op equal the reference's."""
    assert C.sizeof(_DR) == 36 and C.sizeof(_IR) == 76
    d, g, inv = _DR(), _IR(), (C.c_int * 2)()
    seen = set()
    for op, w, r, tag, rn, c, dnum, dw, dsel, dinv, gnum, gw, gsel, ginv in KAT["reservoir_ops"]:
        w, r, tag = (float(x) for x in f32([w, r, tag]))
        lib().orc_resv_op(C.byref(d), C.byref(g), op, w, r, tag, rn, c, inv)
        seen.add(op)
        got = (d.num, bits(d.weight), bits(d.ls.dist), inv[0], g.num, bits(g.weight), bits(g.gs.pHat), inv[1])
        want = (dnum, dw, dsel, dinv, gnum, gw, gsel, ginv)
        if np.isnan(f32([dw])[0]):                                  # NaN payloads are not part of the contract
            assert np.isnan(d.weight) and got[0] == want[0] and got[2:5] == want[2:5]
            assert (np.isnan(g.weight) if np.isnan(f32([gw])[0]) else got[5] == want[5]) and got[6:] == want[6:]
        else:
            assert got == want, (op, got, want)
    assert seen == {0, 1, 2, 3, 4}


def test_sun_and_sky_matches_reference_glsl():
    """sun_and_sky.glsl:453-601 over 6 parameter sets (default, hazy, physically scaled sun, sun below the horizon, z-up,
    extreme haze/saturation) x 48 directions each, including the sun disk, its glow, zenith, nadir and the horizon."""
    from restir_amd import abi
    assert len(KAT["sun_and_sky"]) == 6
    for case in KAT["sun_and_sky"]:
        v = case["ss"]
        raw = np.zeros(24, np.uint32)
        raw[:21] = v[:21]
        raw[21:] = np.array(v[21:], dtype=np.int32).view(np.uint32)
        ss = abi.SunAndSky()
        assert C.sizeof(ss) == 96
        C.memmove(C.byref(ss), raw.ctypes.data, 96)
        sm = f32(case["samples"])
        d, want = np.ascontiguousarray(sm[:, 0:3]), sm[:, 3:6]
        out = np.zeros_like(d)
        lib().orc_sun_and_sky_eval(C.byref(ss), len(d), d.ctypes.data, out.ctypes.data)
        assert np.isfinite(out).all() and (want >= 0).all()
        assert (np.abs(out - want) <= 2e-4 * np.abs(want) + 1e-9).all()


def _post(op, values):
    i, o = np.ascontiguousarray(values, np.float32), np.zeros(3, np.float32)
    lib().orc_post_fn(op, i.ctypes.data, o.ctypes.data)
    return o


def test_display_pass_helpers_match_reference_glsl():
    for row in KAT["pcg3d"]:                                          # random.glsl:81-92, integer: bit-exact
        v = np.array(row[:3], np.uint32)
        lib().orc_pcg3d(v.ctypes.data)
        assert list(v) == row[3:]
    for row in KAT["uncharted"]:                                      # tonemapping.glsl:48-65
        assert np.abs(_post(0, f32(row[:3])) - f32(row[3:6])).max() <= 2.4e-7
    for row in KAT["dither"]:                                         # post.frag:50-55: the chosen quantisation step is the same
        assert list(_post(1, f32(row[:6])).view(np.uint32)) == row[6:9]
    for row in KAT["tone_exposure"]:                                  # post.frag:63-68: + * / only, bit-exact
        assert list(_post(2, f32(row[:6])).view(np.uint32)) == row[6:9]
