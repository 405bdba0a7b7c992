"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on identical seeded inputs — bit-exact on every
screen-space buffer (G-buffer words, motion, reservoirs, light ids, all float images), every frame.

Bit-exactness of the float outputs is possible because both sides implement the numerics contract of
include/rt_detmath.h and build with -ffp-contract=off; the tolerance BASELINE.json allows ("stated per-pixel L2
tolerance, reservoir sample indices bit-exact") is therefore stated as 0."""
import numpy as np
import pytest
from helpers import abi, host, make_scene, frame_buffers, compare_buffers, RendererBackend

pytestmark = pytest.mark.gpu


def _pair(sc, env, W, H, latency=True):
    from restir_amd.renderer import Renderer
    from oracle.binding import Oracle
    desc = sc.desc(env)
    o = Oracle(0); o.upload_scene(desc); o.resize(W, H)
    r = Renderer().setup(0); r.load_scene(desc); r.update(W, H)
    # both builds of the traced kernels must reproduce the oracle bit for bit (AUTO would pick the latency build for every image this small)
    r.set_traversal(abi.TRAVERSAL_LATENCY if latency else abi.TRAVERSAL_THROUGHPUT)
    return o, r


def _run(sc, st, o, r, W, H, nframes, moving=False, stages=None, buffers=None):
    gpu = RendererBackend(r)
    eye, center, up, fov = sc.cameraPose()
    sc.updateCamera(W, H)
    for f in range(nframes):
        st.time = 1000 + f
        if moving:
            sc.setCamera(eye + np.array([0.04 * f, 0.01 * f, -0.03 * f], dtype=np.float32), center, up, fov)
        sc.updateCamera(W, H)
        cam = sc.getCamera()
        o.set_camera(cam); gpu.set_camera(cam)
        if stages is None:
            o.render_frame(st, f); gpu.render_frame(st, f)
        else:
            for stage, level in stages:
                o.run_stage(st, f, stage, level); gpu.run_stage(st, f, stage, level)
        cmp = compare_buffers(o, gpu, buffers(f) if buffers else frame_buffers(f))
        bad = {k: v for k, v in cmp.items() if v[0]}
        assert not bad, f"frame {f}: {bad}"


CASES = [
    # name, kind, scale, W, H, frames, env, moving
    ("cornell", abi.PROC_CORNELL, 1.0, 256, 256, 3, None, False),            # BASELINE config 2 class (DI+GI here)
    ("cornell-odd-size", abi.PROC_CORNELL, 1.0, 101, 51, 3, None, True),     # ragged tiles, odd half-res
    ("helmet-env", abi.PROC_HELMET, 0.05, 128, 128, 3, (256, 128), False),   # textures: base colour, metal-rough, normal map
    ("sponza-moving", abi.PROC_SPONZA, 0.02, 320, 180, 4, (512, 256), True), # instancing, emissive mesh, env, reprojection
    ("bistro-ext", abi.PROC_BISTRO_EXT, 0.01, 320, 180, 3, (512, 256), True),  # alpha-masked foliage, mirrored instances
    ("bistro-int", abi.PROC_BISTRO_INT, 0.01, 256, 144, 3, (128, 64), False),
    ("bistro-ext-real", abi.PROC_BISTRO_EXT_REAL, 0.01, 320, 180, 3, (512, 256), True),  # 16 cut-out cards, 128 texture sets, beams / strips (round 5)
]


@pytest.mark.parametrize("latency", [True, False], ids=["latency", "throughput"])
@pytest.mark.parametrize("name,kind,scale,W,H,frames,env_size,moving", CASES, ids=[c[0] for c in CASES])
def test_full_frame_bit_exact(name, kind, scale, W, H, frames, env_size, moving, latency):
    sc, env = make_scene(kind, scale, 1, env_size)
    st = host.default_state(W, H, sc, env)
    if env is None:
        st.environmentProb = 0.0; st.fireflyClampThreshold = 100.0
    o, r = _pair(sc, env, W, H, latency)
    _run(sc, st, o, r, W, H, frames, moving)
    img = r.readback(abi.BUF_DIRECT_RESULT0 + ((frames - 1) & 1)).view(np.float32)
    assert np.isfinite(img).all() and img.max() > 0.01       # not comparing two empty frames


@pytest.mark.parametrize("latency", [True, False], ids=["latency", "throughput"])
def test_gltf_file_scene_bit_exact(latency):
    """Scene::load path: the hand-authored tests/golden/mini_scene.gltf (TRS hierarchy, PNG textures with nearest / clamp /
    mirror samplers, MASK material, transmission + ior, emissive strength => triangle lights, spot light, file camera)."""
    import os
    W, H = 160, 96
    sc = host.Scene()
    assert sc.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mini_scene.gltf"))
    env = host.HdrSampling(); env.makeSyntheticSky(64, 32, 5e3, 7)
    st = host.default_state(W, H, sc, env)
    o, r = _pair(sc, env, W, H, latency)
    _run(sc, st, o, r, W, H, 3, moving=True)
    g = r.readback(abi.BUF_GBUFFER0 + 0).view(np.uint32)
    assert (g != g.flat[0]).any()                            # the camera from the file sees geometry


def test_back_to_back_frames_vs_oracle():
    """Frames submitted without any synchronisation in between (the pipelined path of rt_render_frame) against the oracle."""
    W, H = 320, 180
    sc, env = make_scene(abi.PROC_SPONZA, 0.02, 1, (512, 256))
    st = host.default_state(W, H, sc, env)
    o, r = _pair(sc, env, W, H, latency=False)
    gpu = RendererBackend(r)
    eye, center, up, fov = sc.cameraPose()
    sc.updateCamera(W, H)
    N = 6
    for f in range(N):
        st.time = 1000 + f
        sc.setCamera(eye + np.array([0.04 * f, 0.01 * f, -0.03 * f], dtype=np.float32), center, up, fov)
        sc.updateCamera(W, H)
        cam = sc.getCamera()
        o.set_camera(cam); gpu.set_camera(cam)
        o.render_frame(st, f); gpu.render_frame(st, f)
    cmp = compare_buffers(o, gpu, frame_buffers(N - 1) + [abi.BUF_GBUFFER0 + (N & 1), abi.BUF_DIRECT_RESV0 + (N & 1), abi.BUF_INDIRECT_RESV0 + (N & 1)])
    bad = {k: v for k, v in cmp.items() if v[0]}
    assert not bad, bad


def test_cornell_config2_di_only_512():
    """BASELINE config 2: Cornell box 512x512, ReSTIR DI only (temporal, M=4, clamp 80), time = 1000+frame, 8 frames."""
    W = H = 512
    sc, env = make_scene(abi.PROC_CORNELL)
    st = host.default_state(W, H, sc, None); st.environmentProb = 0.0; st.fireflyClampThreshold = 100.0
    o, r = _pair(sc, None, W, H)
    bufs = lambda f: frame_buffers(f, indirect=False)  # noqa: E731
    _run(sc, st, o, r, W, H, 8, stages=[(abi.STAGE_DIRECT, 0)], buffers=bufs)


@pytest.mark.parametrize("latency", [True, False], ids=["latency", "throughput"])
@pytest.mark.parametrize("variant", ["restir_none", "ris_only", "no_denoise", "no_modulate", "no_mis_depth2", "debug_normal", "gen_reuse_split", "m16_clamp4", "spatial", "spatiotemporal"])
def test_state_variants(variant, latency):
    W, H = 160, 96
    sc, env = make_scene(abi.PROC_SPONZA, 0.01, 1, (128, 64))
    st = host.default_state(W, H, sc, env)
    stages = None
    if variant == "restir_none": st.ReSTIRState = abi.RESTIR_NONE
    if variant == "ris_only": st.ReSTIRState = abi.RESTIR_RIS
    if variant == "spatial": st.ReSTIRState = abi.RESTIR_SPATIAL
    if variant == "spatiotemporal": st.ReSTIRState = abi.RESTIR_SPATIOTEMPORAL
    if variant == "no_denoise": st.denoise = 0
    if variant == "no_modulate": st.modulate = 0
    if variant == "no_mis_depth2": st.MIS = 0; st.maxDepth = 2
    if variant == "debug_normal": st.debugging_mode = 4
    if variant == "m16_clamp4": st.RISSampleNum = 16; st.reservoirClamp = 4
    if variant == "gen_reuse_split":
        stages = [(abi.STAGE_DIRECT_GEN, 0), (abi.STAGE_DIRECT_REUSE, 0), (abi.STAGE_INDIRECT, 0)] + \
                 [(abi.STAGE_DENOISE_DIRECT, l) for l in range(4)] + [(abi.STAGE_DENOISE_INDIRECT, l) for l in range(5)] + [(abi.STAGE_COMPOSE, 0)]
    o, r = _pair(sc, env, W, H, latency)
    bufs = (lambda f: frame_buffers(f) + [abi.BUF_DIRECT_RESV_TEMP]) if variant.startswith("spatial") or variant == "spatiotemporal" else None
    _run(sc, st, o, r, W, H, 3, moving=True, stages=stages, buffers=bufs)


def test_history_upload_roundtrip_and_determinism():
    """rt_upload_history / rt_readback: restoring a history snapshot reproduces the next frame exactly."""
    from restir_amd.renderer import Renderer
    W, H = 128, 72
    sc, env = make_scene(abi.PROC_HELMET, 0.03, 1, (64, 32))
    st = host.default_state(W, H, sc, env)
    r = Renderer().setup(0); r.load_scene(sc.desc(env)); r.update(W, H)
    sc.updateCamera(W, H)
    cams = []
    for f in range(2):
        st.time = 10 + f; sc.updateCamera(W, H); cams.append(sc.getCamera()); r.set_camera(cams[-1]); r.run(st, f)
    snap = {b: r.readback(b) for b in range(abi.BUF_COUNT)}
    st.time = 12; sc.updateCamera(W, H); cam2 = sc.getCamera(); r.set_camera(cam2); r.run(st, 2)
    want = {b: r.readback(b) for b in frame_buffers(2)}
    r2 = Renderer().setup(0); r2.load_scene(sc.desc(env)); r2.update(W, H)
    for b, data in snap.items():
        r2.upload_history(b, data)
    r2.set_camera(cam2); r2.run(st, 2)
    for b, data in want.items():
        assert np.array_equal(r2.readback(b), data), abi.BUFFER_NAMES[b]


def test_error_paths():
    from restir_amd.renderer import Renderer, RtError
    r = Renderer().setup(0)
    st = host.default_state(64, 64)
    with pytest.raises(RtError, match="no scene"):
        r.run(st, 0)
    sc, _ = make_scene(abi.PROC_CORNELL)
    r.load_scene(sc.desc())
    with pytest.raises(RtError, match="size"):
        r.run(st, 0)
    r.update(64, 64)
    with pytest.raises(RtError, match="multiple of 8"):
        r.run_stage(st, 0, abi.STAGE_DIRECT, 0, 3, 20)
    with pytest.raises(RtError):
        r.update(0, 10)
    r.set_camera(sc.getCamera()); r.run(st, 0); r.sync()


def test_malformed_scene_is_rejected_not_faulted():
    """Everything the kernels use as an array index is validated by rt_upload_scene: a bad id is an error code, never a GPU fault."""
    import ctypes as C
    from restir_amd.renderer import Renderer, RtError
    r = Renderer().setup(0)
    sc, env = make_scene(abi.PROC_SPONZA, 0.01, 1, (64, 32))
    good = sc.desc(env)

    def corrupted(patch):
        d = abi.SceneDesc.from_buffer_copy(good)
        keep = patch(d)                       # returns the patched host array (kept alive during the call)
        with pytest.raises(RtError):
            r.load_scene(d)
        return keep

    def bad_index(d):
        a = (C.c_uint32 * d.numIndices).from_buffer_copy(C.string_at(d.indices, d.numIndices * 4)); a[5] = 0x7fffffff
        d.indices = C.cast(a, C.c_void_p).value; return a
    def bad_texture(d):
        a = bytearray(C.string_at(d.materials, d.numMaterials * 80)); a[16:20] = (123456).to_bytes(4, "little")   # pbrBaseColorTexture
        b = (C.c_uint8 * len(a)).from_buffer(a); d.materials = C.cast(b, C.c_void_p).value; return (a, b)
    def bad_light_material(d):
        a = bytearray(C.string_at(d.trigLights, d.lightInfo.trigLightSize * 96)); a[0:4] = (99999).to_bytes(4, "little")
        b = (C.c_uint8 * len(a)).from_buffer(a); d.trigLights = C.cast(b, C.c_void_p).value; return (a, b)
    def bad_env_alias(d):
        n = d.envWidth * d.envHeight
        a = bytearray(C.string_at(d.envAccel, n * 16)); a[16:20] = (n + 7).to_bytes(4, "little")
        b = (C.c_uint8 * len(a)).from_buffer(a); d.envAccel = C.cast(b, C.c_void_p).value; return (a, b)
    assert good.lightInfo.trigLightSize > 0 and good.envAccel
    for patch in (bad_index, bad_texture, bad_light_material, bad_env_alias):
        corrupted(patch)
    r.load_scene(good); r.update(64, 48)                             # the context is still usable afterwards
    st = host.default_state(64, 48, sc, env)
    sc.updateCamera(64, 48); r.set_camera(sc.getCamera()); r.run(st, 0); r.sync()


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


def test_pick_matches_oracle():
    """rt_pick (SampleExample::screenPicking / nvvk::RayPickerKHR stand-in): same hit record as the oracle for a grid of window positions."""
    W, H = 64, 64
    sc, env = make_scene(abi.PROC_SPONZA, 0.02, 1)
    o, r = _pair(sc, None, W, H, latency=False)
    sc.updateCamera(W, H)
    cam = sc.getCamera()
    hits = 0
    for y in np.linspace(0.05, 0.95, 7):
        for x in np.linspace(0.05, 0.95, 7):
            a = r.pick(cam.viewInverse, cam.projInverse, float(x), float(y)); b = o.pick(cam.viewInverse, cam.projInverse, float(x), float(y))
            assert bytes(a) == bytes(b), (x, y)
            hits += a.instanceID >= 0
            if a.instanceID >= 0:
                assert abs(sum(a.baryCoord) - 1.0) < 1e-5 and a.hitT > 0
    assert hits > 20


def test_cpp_demo_application_matches_python_host(tmp_path):
    """host/restir_demo (C++ Scene / AccelStructure / Renderer / RenderOutput classes of host/renderer.hpp in main.cpp's call
    order) against the same frames driven through the ctypes mirror: the HDR sum (.pfm) and the tonemapped image (.ppm)."""
    import os, subprocess
    from helpers import ROOT
    from restir_amd.renderer import Renderer
    demo = os.path.join(ROOT, "cis-565-final-vr-raytracer_amd", "host", "restir_demo")
    if not os.path.exists(demo):
        pytest.skip("restir_demo not built (run __graft_entry__.build())")
    gltf = os.path.join(ROOT, "tests", "golden", "mini_scene.gltf")
    W, H, N = 96, 64, 3
    out = str(tmp_path / "demo")
    subprocess.check_call([demo, "-f", gltf, "-w", str(W), "-h", str(H), "-n", str(N), "-o", out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(out + ".pfm", "rb") as f:
        assert f.readline().strip() == b"PF" and f.readline().split() == [str(W).encode(), str(H).encode()] and float(f.readline()) < 0
        hdr = np.frombuffer(f.read(), dtype="<f4").reshape(H, W, 3)[::-1]                  # PFM scanlines are bottom-up
    with open(out + ".ppm", "rb") as f:
        assert f.readline().strip() == b"P6"; f.readline(); f.readline()
        ldr = np.frombuffer(f.read(), dtype=np.uint8).reshape(H, W, 3)
    # the same through Python: RtxState exactly as restir_demo.cpp fills it (no environment for a -f scene without -e)
    sc = host.Scene(); assert sc.load(gltf)
    st = host.default_state(W, H, sc, None)
    st.fireflyClampThreshold = 100.0; st.environmentProb = 0.0; st.envMapLuminIntegInv = 0.0
    r = Renderer().setup(0); r.load_scene(sc.desc(None)); r.update(W, H)
    for f in range(N):
        sc.updateCamera(W, H); r.set_camera(sc.getCamera())
        st.frame = f; st.time = 1000 + f
        r.run(st, f)
    cur = (N - 1) & 1
    d = r.readback(abi.BUF_DIRECT_RESULT0 + cur).view(np.float32).reshape(H, W, 4)
    i = r.readback(abi.BUF_INDIRECT_RESULT0 + cur).view(np.float32).reshape(H, W, 4)
    assert np.array_equal((d[..., :3] + i[..., :3]).view(np.uint32), hdr.view(np.uint32))
    r.tonemap(abi.Tonemapper(), 0, N - 1)
    assert np.array_equal(r.readback(abi.BUF_LDR).reshape(H, W, 4)[..., :3], ldr)


@pytest.mark.parametrize("persist", [1, 2, 3, 5])
def test_persistent_multibounce_waves_bit_exact(persist, monkeypatch):
    """Round 6: the multi-bounce tiles of LARGE indirect launches (> 3072 half-res tiles: the three-tiles-per-wave single-bounce body + persistent waves that regenerate a
    path per lane from their XCD's tile list, csrc/stages.hip indirectMultiBouncePersistent) against the oracle: full frames of 1040 x 784 (65 x 49 half-res tiles), maxDepth 4,
    temporal reuse under a moving camera, a scene with sky pixels (lanes that pull a pixel without a surface), RESTIR_IND_PERSIST tiles' worth of pixels per wave."""
    monkeypatch.setenv("RESTIR_IND_PERSIST", str(persist))
    W, H = 1040, 784
    sc, env = make_scene(abi.PROC_BISTRO_EXT_REAL, 0.01, 1, (256, 128))
    st = host.default_state(W, H, sc, env)
    st.maxDepth = 4
    o, r = _pair(sc, env, W, H, latency=False)
    stages = [(abi.STAGE_DIRECT, 0), (abi.STAGE_INDIRECT, 0)]      # (the filters do not depend on how the indirect stage is organised: the traced stages' buffers are compared)
    bufs = lambda f: [abi.BUF_GBUFFER0 + (f & 1), abi.BUF_MOTION, abi.BUF_DIRECT_RESV0 + (f & 1), abi.BUF_INDIRECT_RESV0 + (f & 1), abi.BUF_DENOISE_IND_A]   # noqa: E731
    _run(sc, st, o, r, W, H, 3, moving=True, stages=stages, buffers=bufs)
    img = r.readback(abi.BUF_DENOISE_IND_A).view(np.float32)
    assert np.isfinite(img).all() and img.max() > 0.0
