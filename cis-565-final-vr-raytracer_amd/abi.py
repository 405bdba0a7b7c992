"""ctypes mirrors of include/rt_abi.h (the data contract of shaders/host_device.h:153-333)."""
import ctypes as C

class Vec2(C.Structure): _fields_ = [("x", C.c_float), ("y", C.c_float)]
class Vec3(C.Structure): _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]
class Vec4(C.Structure): _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]
class IVec2(C.Structure): _fields_ = [("x", C.c_int32), ("y", C.c_int32)]
class Mat4(C.Structure): _fields_ = [("m", C.c_float * 16)]

class SceneCamera(C.Structure):  # host_device.h:153-165
    _fields_ = [("viewInverse", Mat4), ("projInverse", Mat4), ("projView", Mat4), ("lastView", Mat4),
                ("lastProjView", Mat4), ("lastPosition", Vec3), ("nbLights", C.c_int32)]

class RtxState(C.Structure):  # host_device.h:207-238
    _fields_ = [("frame", C.c_int32), ("maxDepth", C.c_int32), ("modulate", C.c_int32), ("fireflyClampThreshold", C.c_float),
                ("hdrMultiplier", C.c_float), ("debugging_mode", C.c_int32), ("environmentProb", C.c_float), ("time", C.c_uint32),
                ("ReSTIRState", C.c_int32), ("RISSampleNum", C.c_int32), ("reservoirClamp", C.c_int32), ("accumulate", C.c_int32),
                ("size", IVec2), ("envMapLuminIntegInv", C.c_float), ("lightLuminIntegInv", C.c_float), ("MIS", C.c_int32),
                ("sigLuminDirect", C.c_float), ("sigNormalDirect", C.c_float), ("sigDepthDirect", C.c_float), ("denoise", C.c_int32),
                ("sigLuminIndirect", C.c_float), ("sigNormalIndirect", C.c_float), ("sigDepthIndirect", C.c_float), ("denoiseLevel", C.c_int32)]

class ImptSamp(C.Structure): _fields_ = [("alias", C.c_int32), ("q", C.c_float), ("pdf", C.c_float), ("aliasPdf", C.c_float)]
class LightBufInfo(C.Structure): _fields_ = [("puncLightSize", C.c_uint32), ("trigLightSize", C.c_uint32), ("trigSampProb", C.c_float), ("pad", C.c_int32)]

class SceneDesc(C.Structure):  # rt_scene_desc
    _fields_ = [("numPrimMeshes", C.c_uint32), ("primMeshes", C.c_void_p),
                ("numVertices", C.c_uint64), ("vertices", C.c_void_p),
                ("numIndices", C.c_uint64), ("indices", C.c_void_p),
                ("numInstances", C.c_uint32), ("instances", C.c_void_p),
                ("numMaterials", C.c_uint32), ("materials", C.c_void_p),
                ("numTextures", C.c_uint32), ("textures", C.c_void_p),
                ("puncLights", C.c_void_p), ("trigLights", C.c_void_p), ("lightInfo", LightBufInfo),
                ("envWidth", C.c_int32), ("envHeight", C.c_int32), ("envRgba32f", C.c_void_p), ("envAccel", C.c_void_p)]

class Tonemapper(C.Structure):  # rt_tonemapper (host_device.h:336-351); defaults = RenderOutput::m_tm (render_output.hpp:44-55)
    _fields_ = [("brightness", C.c_float), ("contrast", C.c_float), ("saturation", C.c_float), ("vignette", C.c_float), ("avgLum", C.c_float), ("zoom", C.c_float),
                ("renderingRatio", C.c_float * 2), ("autoExposure", C.c_int32), ("Ywhite", C.c_float), ("key", C.c_float), ("pad", C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        self.brightness = self.contrast = self.saturation = 1.0; self.vignette = 0.0; self.avgLum = 1.0; self.zoom = 1.0
        self.renderingRatio[0] = self.renderingRatio[1] = 1.0; self.autoExposure = 0; self.Ywhite = 0.5; self.key = 0.5
        for k, v in kw.items():
            setattr(self, k, v)

assert C.sizeof(Tonemapper) == 48

class SunAndSky(C.Structure):  # rt_sun_and_sky (host_device.h:353-377); defaults = SampleExample::m_sunAndSky (sample_example.hpp:186-203)
    _fields_ = [("rgb_unit_conversion", C.c_float * 3), ("multiplier", C.c_float), ("haze", C.c_float), ("redblueshift", C.c_float), ("saturation", C.c_float),
                ("horizon_height", C.c_float), ("ground_color", C.c_float * 3), ("horizon_blur", C.c_float), ("night_color", C.c_float * 3),
                ("sun_disk_intensity", C.c_float), ("sun_direction", C.c_float * 3), ("sun_disk_scale", C.c_float), ("sun_glow_intensity", C.c_float),
                ("y_is_up", C.c_int32), ("physically_scaled_sun", C.c_int32), ("in_use", C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        self.rgb_unit_conversion[:] = [1, 1, 1]; self.multiplier = 0.0000101320; self.haze = 0.0; self.redblueshift = 0.0; self.saturation = 1.0
        self.horizon_height = 0.0; self.ground_color[:] = [0.4, 0.4, 0.4]; self.horizon_blur = 0.1; self.night_color[:] = [0.0, 0.0, 0.01]
        self.sun_disk_intensity = 0.8; self.sun_direction[:] = [0.0, 0.78, 0.62]; self.sun_disk_scale = 5.0; self.sun_glow_intensity = 1.0
        self.y_is_up = 1; self.physically_scaled_sun = 1; self.in_use = 0
        for k, v in kw.items():
            if isinstance(v, (list, tuple)):
                getattr(self, k)[:] = v
            else:
                setattr(self, k, v)

assert C.sizeof(SunAndSky) == 96

class PickResult(C.Structure):  # rt_pick_result == nvvk::RayPickerKHR::PickResult
    _fields_ = [("worldRayOrigin", C.c_float * 4), ("worldRayDirection", C.c_float * 4), ("hitT", C.c_float), ("primitiveID", C.c_int32), ("instanceID", C.c_int32),
                ("instanceCustomIndex", C.c_int32), ("baryCoord", C.c_float * 3)]
RT_STAGE_COUNT = 7
TRAVERSAL_AUTO, TRAVERSAL_THROUGHPUT, TRAVERSAL_LATENCY = 0, 1, 2   # rt_set_traversal
class Counters(C.Structure):  # rt_counters
    _fields_ = [("closestHitRays", C.c_uint64), ("anyHitRays", C.c_uint64), ("nodesVisited", C.c_uint64), ("trisTested", C.c_uint64),
                ("hitsShaded", C.c_uint64), ("risCandidates", C.c_uint64), ("stageMs", C.c_float * RT_STAGE_COUNT), ("frameMs", C.c_float), ("framesTimed", C.c_uint32),
                ("laneRounds", C.c_uint64), ("laneLiveRounds", C.c_uint64)]

assert C.sizeof(SceneCamera) == 336 and C.sizeof(RtxState) == 100

# rt_buffer_id
(BUF_GBUFFER0, BUF_GBUFFER1, BUF_MOTION, BUF_DIRECT_RESV0, BUF_DIRECT_RESV1, BUF_DIRECT_RESV_TEMP, BUF_INDIRECT_RESV0,
 BUF_INDIRECT_RESV1, BUF_INDIRECT_RESV_TEMP, BUF_DENOISE_DIR_A, BUF_DENOISE_DIR_B, BUF_DENOISE_IND_A, BUF_DENOISE_IND_B,
 BUF_DIRECT_RESULT0, BUF_DIRECT_RESULT1, BUF_INDIRECT_RESULT0, BUF_INDIRECT_RESULT1, BUF_LIGHT_ID0, BUF_LIGHT_ID1, BUF_LDR, BUF_COUNT) = range(21)
BUFFER_NAMES = ["gbuffer0", "gbuffer1", "motion", "direct_resv0", "direct_resv1", "direct_resv_temp", "indirect_resv0", "indirect_resv1",
                "indirect_resv_temp", "denoise_dir_a", "denoise_dir_b", "denoise_ind_a", "denoise_ind_b", "direct_result0", "direct_result1",
                "indirect_result0", "indirect_result1", "light_id0", "light_id1", "ldr"]
# rt_stage_id
(STAGE_DIRECT, STAGE_INDIRECT, STAGE_DENOISE_DIRECT, STAGE_DENOISE_INDIRECT, STAGE_COMPOSE, STAGE_DIRECT_GEN, STAGE_DIRECT_REUSE) = range(7)
# rt_restir_state
RESTIR_NONE, RESTIR_RIS, RESTIR_SPATIAL, RESTIR_TEMPORAL, RESTIR_SPATIOTEMPORAL = range(5)
# ProcScene (host/scene.hpp)
PROC_CORNELL, PROC_HELMET, PROC_SPONZA, PROC_BISTRO_EXT, PROC_BISTRO_INT, PROC_BISTRO_EXT_REAL, PROC_SPONZA_1K = range(7)
