#!/usr/bin/env python3
"""Generates tests/golden/mini_scene.gltf (+ mini_scene_expected.json): a hand-authored glTF 2.0 file that exercises what
host/gltf_loader.cpp has to understand, written with nothing but json/struct/zlib so that it is independent of the C++ writer:

  * node hierarchy with TRS on the parent and a matrix on the child, a second instance of the same mesh
  * an interleaved vertex buffer (byteStride), uint16 indices, normalized-ubyte COLOR_0, no NORMAL/TANGENT (synthesised on load)
  * a second primitive with float attributes, uint8 indices and NORMAL
  * PNG textures: 4x4 RGB using all five scanline filters, 2x2 palette+tRNS, 2x2 gray+alpha; one "JPEG" (undecodable) image
  * materials with transmission / ior / emissive_strength extensions, MASK alpha mode
  * KHR_lights_punctual spot light under a transformed node, a perspective camera
  * a mesh-less scene root listed in `scenes`, plus an orphan node that must NOT be instanced

Usage: python tests/golden/make_mini_gltf.py   (rewrites the two files next to it)"""
import base64, json, os, struct, zlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

def png(w, h, ctype, rows, filters, plte=None, trns=None):
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = bytearray()
    prev = bytes(w * ch)
    for y in range(h):
        cur = bytes(rows[y]); ft = filters[y % len(filters)]
        out = bytearray()
        for x in range(w * ch):
            a = cur[x - ch] if x >= ch else 0; b = prev[x]; c = prev[x - ch] if x >= ch else 0
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = b
            elif ft == 3: p = (a + b) >> 1
            else:
                q = a + b - c; pa, pb, pc = abs(q - a), abs(q - b), abs(q - c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out.append((cur[x] - p) & 255)
        raw.append(ft); raw += out; prev = cur
    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)
    z = zlib.compress(bytes(raw), 9)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
    if plte: data += chunk(b"PLTE", bytes(plte))
    if trns: data += chunk(b"tRNS", bytes(trns))
    half = len(z) // 2  # two IDAT chunks: the decoder has to concatenate them
    return data + chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")

def uri(b, mime="application/octet-stream"):
    return f"data:{mime};base64," + base64.b64encode(b).decode()

def quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

def trs(t, q, s):
    m = np.eye(4); m[:3, :3] = quat_to_mat(q) @ np.diag(s); m[:3, 3] = t; return m

def main():
    rng = np.random.RandomState(5)
    # ---- primitive A: quad, interleaved pos(12) + uv(8) + color ubyte4 normalized (4) = stride 24, uint16 indices
    posA = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    uvA = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    colA = np.array([[255, 0, 0, 255], [0, 255, 0, 255], [0, 0, 255, 255], [255, 255, 255, 128]], np.uint8)
    inter = b"".join(posA[i].tobytes() + uvA[i].tobytes() + colA[i].tobytes() for i in range(4))
    idxA = np.array([0, 2, 1, 0, 3, 2], np.uint16)
    # ---- primitive B: tetra-ish fan with normals, uint8 indices
    posB = np.array([[0, 0.2, 0], [0.5, 0.2, 0], [0, 0.7, 0], [0, 0.2, 0.5]], np.float32)
    nrmB = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    idxB = np.array([0, 1, 2, 0, 2, 3, 0, 3, 1], np.uint8)
    blob = bytearray()
    views = []
    def add(b, stride=None):
        while len(blob) % 4: blob.append(0)
        v = {"buffer": 0, "byteOffset": len(blob), "byteLength": len(b)}
        if stride: v["byteStride"] = stride
        blob.extend(b); views.append(v); return len(views) - 1
    vInter = add(inter, 24); vIdxA = add(idxA.tobytes()); vPosB = add(posB.tobytes()); vNrmB = add(nrmB.tobytes()); vIdxB = add(idxB.tobytes())
    # ---- images
    rgb = rng.randint(0, 256, size=(4, 12)).astype(np.uint8)
    img0 = png(4, 4, 2, rgb, [0, 1, 2, 3, 4][1:] + [0])          # filters 1,2,3,4 on the four rows
    pal = [10, 20, 30, 200, 100, 50, 0, 255, 0]; trn = [255, 64]
    img1 = png(2, 2, 3, np.array([[0, 1], [2, 1]], np.uint8), [0], plte=pal, trns=trn)
    ga = np.array([[40, 255, 90, 10], [200, 128, 7, 0]], np.uint8)
    img2 = png(2, 2, 4, ga, [4, 2])
    vImg1 = add(img1)
    accessors = [
        {"bufferView": vInter, "byteOffset": 0, "componentType": 5126, "count": 4, "type": "VEC3", "min": [-1, 0, -1], "max": [1, 0, 1]},
        {"bufferView": vInter, "byteOffset": 12, "componentType": 5126, "count": 4, "type": "VEC2"},
        {"bufferView": vInter, "byteOffset": 20, "componentType": 5121, "normalized": True, "count": 4, "type": "VEC4"},
        {"bufferView": vIdxA, "componentType": 5123, "count": 6, "type": "SCALAR"},
        {"bufferView": vPosB, "componentType": 5126, "count": 4, "type": "VEC3", "min": [0, 0.2, 0], "max": [0.5, 0.7, 0.5]},
        {"bufferView": vNrmB, "componentType": 5126, "count": 4, "type": "VEC3"},
        {"bufferView": vIdxB, "componentType": 5121, "count": 9, "type": "SCALAR"},
    ]
    parent_t, parent_q, parent_s = [1.0, 2.0, -3.0], [0.0, 0.38268343, 0.0, 0.92387953], [2.0, 1.0, 0.5]
    child_m = trs([0.25, 0, 0], [0, 0, 0, 1], [1, 1, 1])
    light_t, light_q = [0.0, 3.0, 0.0], [-0.70710678, 0.0, 0.0, 0.70710678]
    cam_t = [0.0, 1.0, 6.0]
    gltf = {
        "asset": {"version": "2.0", "generator": "tests/golden/make_mini_gltf.py"},
        "scene": 0,
        "scenes": [{"nodes": [0, 4]}],
        "nodes": [
            {"name": "root", "children": [1, 3, 5]},
            {"name": "parent", "translation": parent_t, "rotation": parent_q, "scale": parent_s, "mesh": 0, "children": [2]},
            {"name": "child", "matrix": [float(v) for v in child_m.T.reshape(-1)], "mesh": 0},
            {"name": "lightNode", "translation": light_t, "rotation": light_q, "extensions": {"KHR_lights_punctual": {"light": 0}}},
            {"name": "camNode", "translation": cam_t, "camera": 0},
            {"name": "emptyLeaf"},
            {"name": "orphan", "mesh": 0, "translation": [100, 100, 100]},
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.6, "znear": 0.1, "aspectRatio": 1.5}}],
        "meshes": [{"primitives": [
            {"attributes": {"POSITION": 0, "TEXCOORD_0": 1, "COLOR_0": 2}, "indices": 3, "material": 0},
            {"attributes": {"POSITION": 4, "NORMAL": 5}, "indices": 6, "material": 1, "mode": 4},
            {"attributes": {"POSITION": 4}, "mode": 1, "material": 1},
        ]}],
        "materials": [
            {"name": "textured", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.7, 0.6, 1.0], "baseColorTexture": {"index": 0}, "metallicFactor": 0.25, "roughnessFactor": 0.75,
                                                         "metallicRoughnessTexture": {"index": 1}},
             "normalTexture": {"index": 2, "scale": 0.5}, "alphaMode": "MASK", "alphaCutoff": 0.3, "doubleSided": True},
            {"name": "glass", "emissiveFactor": [1.0, 0.5, 0.25], "emissiveTexture": {"index": 3},
             "extensions": {"KHR_materials_transmission": {"transmissionFactor": 0.9}, "KHR_materials_ior": {"ior": 1.33}, "KHR_materials_emissive_strength": {"emissiveStrength": 4.0}}},
        ],
        "samplers": [{"magFilter": 9728, "wrapS": 33071, "wrapT": 33648}, {}],
        "textures": [{"source": 0, "sampler": 0}, {"source": 1, "sampler": 1}, {"source": 2}, {"source": 3}],
        "images": [{"uri": uri(img0, "image/png")}, {"bufferView": vImg1, "mimeType": "image/png"}, {"uri": uri(img2, "image/png")},
                   {"uri": uri(b"\xff\xd8\xff\xe0 not really a jpeg", "image/jpeg")}],
        "extensionsUsed": ["KHR_lights_punctual", "KHR_materials_transmission", "KHR_materials_ior", "KHR_materials_emissive_strength"],
        "extensions": {"KHR_lights_punctual": {"lights": [{"type": "spot", "color": [1.0, 0.9, 0.8], "intensity": 50.0, "range": 20.0,
                                                            "spot": {"innerConeAngle": 0.2, "outerConeAngle": 0.6}}]}},
        "bufferViews": views,
        "accessors": accessors,
    }
    gltf["buffers"] = [{"byteLength": len(blob), "uri": uri(bytes(blob))}]
    with open(os.path.join(HERE, "mini_scene.gltf"), "w") as f:
        json.dump(gltf, f, indent=1)
    # .glb with the same content (binary chunk instead of the data URI) is produced by the test from this JSON.

    # ---- expectations computed here, independently of the loader ---------------------------------------------------
    Mp = trs(parent_t, parent_q, parent_s); Mc = Mp @ child_m
    def world(M, P): return (np.c_[P.astype(np.float64), np.ones(len(P))] @ M.T)[:, :3]
    def bgra(img_rgba): return [[int(p[2]), int(p[1]), int(p[0]), int(p[3])] for p in img_rgba]
    rgb_px = rgb.reshape(16, 3); img0_px = [[int(p[0]), int(p[1]), int(p[2]), 255] for p in rgb_px]
    pal_px = [[pal[3 * k], pal[3 * k + 1], pal[3 * k + 2], trn[k] if k < len(trn) else 255] for k in [0, 1, 2, 1]]
    ga_px = [[int(g), int(g), int(g), int(a)] for g, a in ga.reshape(4, 2)]
    Ml = trs(light_t, light_q, [1, 1, 1])
    expected = {
        "stats": {"primMeshes": 2, "nodes": 4, "materials": 2, "textures": 4, "triangles": 5, "instancedTriangles": 10, "vertices": 8, "puncLights": 1},
        "instances_world_positions": [world(Mp, posA).tolist(), world(Mp, posB).tolist(), world(Mc, posA).tolist(), world(Mc, posB).tolist()],
        "instance_prim": [0, 1, 0, 1],
        "indices": [[0, 2, 1, 0, 3, 2], [0, 1, 2, 0, 2, 3, 0, 3, 1]],
        "uvA": uvA.tolist(), "colA": colA.tolist(), "nrmB": nrmB.tolist(),
        "textures_bgra": [bgra(img0_px), bgra(pal_px), bgra(ga_px), [[255, 255, 255, 255]]],
        "texture_size": [[4, 4], [2, 2], [2, 2], [1, 1]],
        "sampler": [[33071, 33648, 9728], [10497, 10497, 9729], [10497, 10497, 9729], [10497, 10497, 9729]],
        "camera": {"eye": cam_t, "forward": [0, 0, -1], "fov_deg": 0.6 * 180 / np.pi},
        "light": {"position": light_t, "direction": (Ml[:3, :3] @ np.array([0, 0, -1.0])).tolist(), "color": [1.0, 0.9, 0.8], "intensity": 50.0, "range": 20.0, "inner": 0.2, "outer": 0.6, "type": 2},
        "material0": {"albedo": [0.8, 0.7, 0.6, 1.0], "metallic": 0.25, "roughness": 0.75, "alphaMode": 1, "alphaCutoff": 0.3, "doubleSided": 1, "textures": [0, 1, 2], "normalScale": 0.5},
        "material1": {"emissive": [4.0, 2.0, 1.0], "transmission": 0.9, "ior": 1.33, "emissiveTexture": 3},
    }
    with open(os.path.join(HERE, "mini_scene_expected.json"), "w") as f:
        json.dump(expected, f, indent=1)

if __name__ == "__main__":
    main()
