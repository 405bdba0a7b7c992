"""glTF ingest (Scene::load, src/scene.cpp:57-173; tinygltf + nvh::GltfScene are un-vendored third-party code, so
host/gltf_loader.cpp is pinned by (1) a hand-authored fixture whose expectations are computed independently in
tests/golden/make_mini_gltf.py and (2) a round trip of the procedural scenes through the C++ writer)."""
import base64, ctypes as C, json, os, struct
import numpy as np
import pytest
from helpers import abi, host

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

VERTEX = np.dtype([("position", "<f4", 3), ("normal", "<u4"), ("texcoord", "<f4", 2), ("tangent", "<u4"), ("color", "<u4")])
PRIM = np.dtype([("vertexOffset", "<u4"), ("vertexCount", "<u4"), ("firstIndex", "<u4"), ("indexCount", "<u4"), ("materialIndex", "<i4")])
INST = np.dtype([("objectToWorld", "<f4", 12), ("primMesh", "<u4"), ("flags", "<u4")])
MAT = np.dtype([("baseColor", "<f4", 4), ("baseColorTexture", "<i4"), ("metallic", "<f4"), ("roughness", "<f4"), ("mrTexture", "<i4"), ("emissiveTexture", "<i4"),
                ("emissive", "<f4", 3), ("normalTexture", "<i4"), ("normalScale", "<f4"), ("transmission", "<f4"), ("transmissionTexture", "<i4"), ("ior", "<f4"),
                ("alphaMode", "<i4"), ("alphaCutoff", "<f4"), ("pad", "<i4")])
TEX = np.dtype([("ptr", "<u8"), ("width", "<i4"), ("height", "<i4"), ("wrapS", "<i4"), ("wrapT", "<i4"), ("magFilter", "<i4"), ("pad", "<i4")])
PUNC = np.dtype([("type", "<i4"), ("direction", "<f4", 3), ("intensity", "<f4"), ("color", "<f4", 3), ("position", "<f4", 3), ("range", "<f4"),
                 ("outerConeCos", "<f4"), ("innerConeCos", "<f4"), ("padding", "<f4", 2), ("impSamp", "<u4", 4)])
assert VERTEX.itemsize == 32 and MAT.itemsize == 80 and PUNC.itemsize == 80 and INST.itemsize == 56 and TEX.itemsize == 32


def _arr(ptr, n, dt):
    if not ptr or n == 0:
        return np.zeros(0, dt)
    return np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(ptr), dtype=dt).copy()


def dump(scene):
    d = scene.desc()
    out = {"prims": _arr(d.primMeshes, d.numPrimMeshes, PRIM), "vertices": _arr(d.vertices, d.numVertices, VERTEX),
           "indices": _arr(d.indices, d.numIndices, np.dtype("<u4")), "instances": _arr(d.instances, d.numInstances, INST),
           "materials": _arr(d.materials, d.numMaterials, MAT), "textures": _arr(d.textures, d.numTextures, TEX),
           "punc": _arr(d.puncLights, d.lightInfo.puncLightSize, PUNC),
           "trig": _arr(d.trigLights, d.lightInfo.trigLightSize, np.dtype(("u1", 96))),
           "lightInfo": (d.lightInfo.puncLightSize, d.lightInfo.trigLightSize, d.lightInfo.trigSampProb)}
    out["texels"] = [_arr(int(t["ptr"]), int(t["width"]) * int(t["height"]), np.dtype(("u1", 4))) for t in out["textures"]]
    return out


def _load(path):
    sc = host.Scene()
    assert sc.load(str(path)), path
    return sc


def _check_mini(sc):
    exp = json.load(open(os.path.join(GOLD, "mini_scene_expected.json")))
    st = sc.getStat()
    for k, v in exp["stats"].items():
        assert st[k] == v, (k, st)
    d = dump(sc)
    # instances: DFS order, primitives of a mesh in file order; orphan node (not in scenes[scene]) is not instanced
    assert list(d["instances"]["primMesh"]) == exp["instance_prim"]
    for i, inst in enumerate(d["instances"]):
        pm = d["prims"][inst["primMesh"]]
        M = inst["objectToWorld"].reshape(3, 4).astype(np.float64)
        P = d["vertices"]["position"][pm["vertexOffset"]:pm["vertexOffset"] + pm["vertexCount"]].astype(np.float64)
        W = P @ M[:, :3].T + M[:, 3]
        assert np.allclose(W, np.array(exp["instances_world_positions"][i]), atol=1e-5), i
    for k, pm in enumerate(d["prims"]):
        assert list(d["indices"][pm["firstIndex"]:pm["firstIndex"] + pm["indexCount"]]) == exp["indices"][k]
    assert list(d["prims"]["materialIndex"]) == [0, 1]
    vA = d["vertices"][:4]
    uv = vA["texcoord"].copy().view(np.uint32) & 0xfffffffe      # LSB of .y carries the tangent handedness (scene.cpp:254)
    assert np.array_equal(uv.view(np.float32) if False else uv, np.array(exp["uvA"], np.float32).view(np.uint32) & 0xfffffffe)
    col = np.stack([(vA["color"] >> s) & 255 for s in (0, 8, 16, 24)], axis=1)
    assert np.array_equal(col, np.array(exp["colA"]))             # normalized ubyte -> float -> packUnorm4x8 is lossless
    assert np.all(d["vertices"]["color"][4:] == 0xffffffff)       # no COLOR_0 => white
    # flags: material 0 is MASK + textured => not force-opaque, double sided => cull disable; material 1 opaque, single sided
    assert list(d["instances"]["flags"]) == [2, 1, 2, 1]
    # textures
    for t, (w, h), smp, px in zip(range(4), exp["texture_size"], exp["sampler"], exp["textures_bgra"]):
        T = d["textures"][t]
        assert (T["width"], T["height"]) == (w, h) and [T["wrapS"], T["wrapT"], T["magFilter"]] == smp, t
        assert np.array_equal(d["texels"][t], np.array(px, np.uint8)), t
    m0, m1 = d["materials"][0], d["materials"][1]
    e0, e1 = exp["material0"], exp["material1"]
    assert np.allclose(m0["baseColor"], e0["albedo"]) and np.isclose(m0["metallic"], e0["metallic"]) and np.isclose(m0["roughness"], e0["roughness"])
    assert [m0["baseColorTexture"], m0["mrTexture"], m0["normalTexture"]] == e0["textures"] and np.isclose(m0["normalScale"], e0["normalScale"])
    assert m0["alphaMode"] == e0["alphaMode"] and np.isclose(m0["alphaCutoff"], e0["alphaCutoff"])
    assert np.allclose(m1["emissive"], e1["emissive"]) and np.isclose(m1["transmission"], e1["transmission"]) and np.isclose(m1["ior"], e1["ior"])
    assert m1["emissiveTexture"] == e1["emissiveTexture"] and m1["baseColorTexture"] == -1 and m1["alphaMode"] == 0
    assert np.allclose(m1["baseColor"], 1) and np.isclose(m1["metallic"], 1) and np.isclose(m1["roughness"], 1)     # glTF defaults
    # light (scene.cpp:700-735: position/direction from the node's world matrix, cone cosines)
    L = d["punc"][0]; el = exp["light"]
    assert L["type"] == el["type"] and np.allclose(L["position"], el["position"], atol=1e-5) and np.allclose(L["direction"], el["direction"], atol=1e-5)
    assert np.allclose(L["color"], el["color"]) and np.isclose(L["intensity"], el["intensity"]) and np.isclose(L["range"], el["range"])
    assert np.isclose(L["outerConeCos"], np.cos(el["outer"]), atol=1e-6) and np.isclose(L["innerConeCos"], np.cos(el["inner"]), atol=1e-6)
    # emissive triangles of material 1 become triangle lights: 3 triangles x 2 instances
    assert d["lightInfo"][1] == 6
    eye, center, up, fov = sc.cameraPose()
    assert np.allclose(eye, exp["camera"]["eye"]) and np.isclose(fov, exp["camera"]["fov_deg"], atol=1e-4)
    f = (center - eye) / np.linalg.norm(center - eye)
    assert np.allclose(f, exp["camera"]["forward"], atol=1e-6)
    return d


def test_mini_fixture_gltf():
    d = _check_mini(_load(os.path.join(GOLD, "mini_scene.gltf")))
    # synthesised normals of the flat quad (no NORMAL attribute): every vertex of primitive A decodes to the same direction
    assert len(set(d["vertices"]["normal"][:4])) == 1


def test_mini_fixture_glb_and_external_bin(tmp_path):
    g = json.load(open(os.path.join(GOLD, "mini_scene.gltf")))
    blob = base64.b64decode(g["buffers"][0]["uri"].split(",", 1)[1])
    ref = dump(_load(os.path.join(GOLD, "mini_scene.gltf")))
    # (a) external .bin next to the .gltf, with a percent-escaped name
    ext = json.loads(json.dumps(g)); ext["buffers"][0]["uri"] = "mini%20data.bin"
    (tmp_path / "mini data.bin").write_bytes(blob)
    (tmp_path / "ext.gltf").write_text(json.dumps(ext))
    # (b) .glb container: buffer 0 without uri = BIN chunk
    glb = json.loads(json.dumps(g)); del glb["buffers"][0]["uri"]
    js = json.dumps(glb).encode(); js += b" " * (-len(js) % 4)
    bn = blob + b"\0" * (-len(blob) % 4)
    body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(bn), 0x004E4942) + bn
    (tmp_path / "mini.glb").write_bytes(b"glTF" + struct.pack("<II", 2, 12 + len(body)) + body)
    for name in ("ext.gltf", "mini.glb"):
        sc = _load(tmp_path / name)
        _check_mini(sc)
        got = dump(sc)
        for k in ("prims", "vertices", "indices", "instances", "materials", "punc", "trig"):
            assert got[k].tobytes() == ref[k].tobytes(), (name, k)


@pytest.mark.parametrize("kind,scale", [(abi.PROC_CORNELL, 1.0), (abi.PROC_HELMET, 0.05), (abi.PROC_SPONZA, 0.02)])
def test_procedural_round_trip_is_bit_exact(tmp_path, kind, scale):
    a = host.Scene().makeProcedural(kind, scale, 3)
    path = str(tmp_path / "rt.gltf")
    assert a.saveGltf(path)
    b = _load(path)
    da, db = dump(a), dump(b)
    assert a.getStat() == b.getStat()
    for k in ("prims", "vertices", "indices", "instances", "materials", "punc", "trig"):
        assert da[k].tobytes() == db[k].tobytes(), k                       # %.9g floats + raw binary attributes => identical upload
    assert da["lightInfo"] == db["lightInfo"]
    for ta, tb, xa, xb in zip(da["textures"], db["textures"], da["texels"], db["texels"]):
        assert [ta[f] for f in ("width", "height", "wrapS", "wrapT", "magFilter")] == [tb[f] for f in ("width", "height", "wrapS", "wrapT", "magFilter")]
        assert np.array_equal(xa, xb)                                      # PNG encode -> decode
    ea, ca, ua, fa = a.cameraPose(); eb, cb, ub, fb = b.cameraPose()
    assert np.allclose(ea, eb) and np.isclose(fa, fb, atol=1e-4)
    assert np.allclose((ca - ea) / np.linalg.norm(ca - ea), (cb - eb) / np.linalg.norm(cb - eb), atol=1e-6)


def test_load_errors(tmp_path):
    sc = host.Scene()
    assert not sc.load(str(tmp_path / "missing.gltf"))
    (tmp_path / "bad.gltf").write_text("{ \"asset\": ")
    assert not sc.load(str(tmp_path / "bad.gltf"))
    (tmp_path / "empty.gltf").write_text(json.dumps({"asset": {"version": "2.0"}, "scenes": [{"nodes": []}], "nodes": []}))
    assert not sc.load(str(tmp_path / "empty.gltf"))
    g = json.load(open(os.path.join(GOLD, "mini_scene.gltf")))
    g["accessors"][3]["count"] = 6000                                       # indices run past the buffer view
    (tmp_path / "oob.gltf").write_text(json.dumps(g))
    sc2 = host.Scene()
    ok = sc2.load(str(tmp_path / "oob.gltf"))
    assert (not ok) or sc2.getStat()["triangles"] <= 5                      # rejected outright or the primitive is dropped; never read out of bounds


# ---- JPEG textures (host/jpeg_decoder.cpp) ------------------------------------------------------------------------------
def _smooth(w, h):
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 17.0) * np.cos(y / 23.0), 128 + 90 * np.cos(x / 29.0 + 1), 128 + 80 * np.sin((x + y) / 31.0)], -1)
    return np.clip(img, 0, 255).astype(np.uint8)


JPEG_VARIANTS = {
    "baseline-444": dict(quality=95, subsampling=0),
    "baseline-422": dict(quality=85, subsampling=1),
    "baseline-420-optimized": dict(quality=90, subsampling=2, optimize=True),
    "progressive-420": dict(quality=92, subsampling=2, progressive=True),
    "progressive-444": dict(quality=95, subsampling=0, progressive=True),
    "restart-markers": dict(quality=90, subsampling=2, restart_marker_blocks=3),
}


@pytest.mark.parametrize("variant", list(JPEG_VARIANTS))
@pytest.mark.parametrize("size", [(64, 48), (77, 53), (200, 9), (1, 1)])
def test_jpeg_decoder_matches_libjpeg(variant, size):
    """Against libjpeg (through PIL) on smooth images: <= 1 code value where no chroma upsampling is involved (inverse DCT
    rounding), <= 8 where libjpeg's triangle-filter upsampling meets this decoder's pixel replication."""
    import io
    Image = pytest.importorskip("PIL.Image")
    w, h = size
    src = _smooth(w, h)
    for mode in ("RGB", "L"):
        im = Image.fromarray(src if mode == "RGB" else src[..., 0], mode)
        bio = io.BytesIO()
        try:
            im.save(bio, "JPEG", **JPEG_VARIANTS[variant])
        except TypeError:
            pytest.skip("this Pillow cannot write restart markers")
        data = bio.getvalue()
        got = host.decode_jpeg(data)
        assert got is not None, (variant, mode)
        ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(int)
        assert got.shape == (h, w, 4) and (got[..., 3] == 255).all()
        d = np.abs(got[..., [2, 1, 0]].astype(int) - ref)
        exact = mode == "L" or JPEG_VARIANTS[variant]["subsampling"] == 0
        assert d.max() <= (3 if exact else 8) and d.mean() < (0.1 if exact else 2.0), (variant, mode, int(d.max()), float(d.mean()))


def test_jpeg_decoder_rejects_what_it_cannot_decode():
    assert host.decode_jpeg(b"\xff\xd8\xff\xe0 not really a jpeg") is None
    assert host.decode_jpeg(b"") is None
    assert host.decode_jpeg(b"\x89PNG\r\n\x1a\n") is None


def test_gltf_with_jpeg_texture(tmp_path):
    import io
    Image = pytest.importorskip("PIL.Image")
    g = json.load(open(os.path.join(GOLD, "mini_scene.gltf")))
    bio = io.BytesIO(); Image.fromarray(_smooth(32, 16), "RGB").save(bio, "JPEG", quality=95, subsampling=0)
    g["images"][3] = {"uri": "data:image/jpeg;base64," + base64.b64encode(bio.getvalue()).decode()}
    (tmp_path / "jpeg.gltf").write_text(json.dumps(g))
    d = dump(_load(tmp_path / "jpeg.gltf"))
    T = d["textures"][3]
    assert (T["width"], T["height"]) == (32, 16)
    ref = np.asarray(Image.open(io.BytesIO(bio.getvalue())).convert("RGB")).astype(int).reshape(-1, 3)
    assert np.abs(d["texels"][3][:, [2, 1, 0]].astype(int) - ref).max() <= 3


def test_gltf_with_16bit_png_texture(tmp_path):
    """16-bit PNG samples keep their high byte."""
    import struct, zlib
    w, h = 5, 3
    rng = np.random.default_rng(2)
    v = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
    raw = b"".join(b"\x00" + v[y].astype(">u2").tobytes() for y in range(h))
    def chunk(tag, body): return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    g = json.load(open(os.path.join(GOLD, "mini_scene.gltf")))
    g["images"][3] = {"uri": "data:image/png;base64," + base64.b64encode(png).decode()}
    (tmp_path / "p16.gltf").write_text(json.dumps(g))
    d = dump(_load(tmp_path / "p16.gltf"))
    assert (d["textures"][3]["width"], d["textures"][3]["height"]) == (w, h)
    assert np.array_equal(d["texels"][3][:, [2, 1, 0]], (v >> 8).astype(np.uint8).reshape(-1, 3)) and (d["texels"][3][:, 3] == 255).all()


def test_png_writer_round_trip(tmp_path):
    """host/png_writer.cpp (the displayed frame) read back by the loader's own PNG reader and by an independent zlib parse"""
    import struct, zlib
    rng = np.random.default_rng(5)
    for (h, w, alpha) in [(1, 1, False), (7, 13, True), (64, 48, False)]:
        img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        p = str(tmp_path / f"t{h}x{w}.png")
        host.write_png(p, img, keep_alpha=alpha)
        data = open(p, "rb").read()
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        # independent parse: chunks + CRCs + inflate + filter 0
        pos, idat, ihdr = 8, b"", None
        while pos < len(data):
            n, typ = struct.unpack(">I4s", data[pos:pos + 8])
            body = data[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
            if typ == b"IHDR": ihdr = struct.unpack(">IIBBBBB", body)
            if typ == b"IDAT": idat += body
            pos += 12 + n
        ch = 4 if alpha else 3
        assert ihdr == (w, h, 8, 6 if alpha else 2, 0, 0, 0)
        raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * ch)
        assert (raw[:, 0] == 0).all() and np.array_equal(raw[:, 1:].reshape(h, w, ch), img[..., :ch])
        back = host.decode_png(data)                       # BGRA
        assert back is not None and np.array_equal(back[..., [2, 1, 0]], img[..., :3])
        if alpha: assert np.array_equal(back[..., 3], img[..., 3])
