/*
 * rt_abi.h — data contract + C-ABI of librestir_hip.so (MI355X / gfx950 ReSTIR DI+GI frame path).
 *
 * This header is the drop-in boundary for the per-frame hot path of
 * IwakuraRein/CIS-565-Final-VR-Raytracer.  It replaces two reference interfaces:
 *
 *   (1) the host<->shader data contract   shaders/host_device.h:153-333
 *       (RtxState push constant, SceneCamera UBO, VertexAttributes, GltfShadeMaterial,
 *        reservoirs, light records, ImptSampData).  Every POD below has the same field
 *        order and the same byte size as the reference struct it cites (GLSL `scalar`
 *        block layout == tightly packed C), checked by static_assert at the bottom.
 *
 *   (2) the Vulkan plumbing inside Renderer / Scene / AccelStructure / HdrSampling
 *       (src/renderer.cpp:97-237, src/scene.cpp:57-125, src/accelstruct.cpp:55-162,
 *        src/hdr_sampling.cpp:56-99): buffer creation + vkCmdDispatch sequences become
 *       the extern "C" calls at the bottom.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  All calls return 0 on success or a
 * negative rt_status; the message is available from rt_last_error().  Nothing throws.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Small POD vector types (nvmath::vecNf stand-ins; column-major mat4 like nvmath::mat4f)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float x, y; } rt_vec2;
typedef struct { float x, y, z; } rt_vec3;
typedef struct { float x, y, z, w; } rt_vec4;
typedef struct { int32_t x, y; } rt_ivec2;
typedef struct { float m[16]; } rt_mat4; /* column-major: m[c*4+r] */

/* ------------------------------------------------------------------------------------------------
 * host_device.h mirrors
 * ---------------------------------------------------------------------------------------------- */

/* host_device.h:128-139 */
enum rt_debug_mode {
  RT_DBG_NONE = 0, RT_DBG_DIRECT_STAGE = 1, RT_DBG_INDIRECT_STAGE = 2, RT_DBG_BASECOLOR = 3, RT_DBG_NORMAL = 4,
  RT_DBG_DEPTH = 5, RT_DBG_METALLIC = 6, RT_DBG_EMISSIVE = 7, RT_DBG_ROUGHNESS = 8, RT_DBG_TEXCOORD = 9
};

/* host_device.h:142-148 */
enum rt_restir_state { RT_RESTIR_NONE = 0, RT_RESTIR_RIS = 1, RT_RESTIR_SPATIAL = 2, RT_RESTIR_TEMPORAL = 3, RT_RESTIR_SPATIOTEMPORAL = 4 };

#define RT_ALPHA_OPAQUE 0 /* host_device.h:179-181 */
#define RT_ALPHA_MASK 1
#define RT_ALPHA_BLEND 2
#define RT_MAX_IOR_MINUS_ONE 3.0f /* host_device.h:182 */
#define RT_INFINITY 1e28f         /* globals.glsl:35 */
#define RT_INVALID_MAT_ID 0xff000000u /* globals.glsl:106 */

/* host_device.h:153-165 — 336 B */
typedef struct {
  rt_mat4 viewInverse;
  rt_mat4 projInverse;
  rt_mat4 projView;
  rt_mat4 lastView;
  rt_mat4 lastProjView;
  rt_vec3 lastPosition;
  int32_t nbLights;
} rt_scene_camera;

/* host_device.h:167-174 — 32 B; packed by scene.cpp:236-258 */
typedef struct {
  rt_vec3 position;
  uint32_t normal;   /* oct-compressed (compress.glsl:111-139) */
  rt_vec2 texcoord;  /* LSB of .y = tangent handedness */
  uint32_t tangent;  /* oct-compressed */
  uint32_t color;    /* unorm8 x4 */
} rt_vertex;

/* host_device.h:183-204 — 80 B */
typedef struct {
  rt_vec4 pbrBaseColorFactor;
  int32_t pbrBaseColorTexture;
  float pbrMetallicFactor;
  float pbrRoughnessFactor;
  int32_t pbrMetallicRoughnessTexture;
  int32_t emissiveTexture;
  rt_vec3 emissiveFactor;
  int32_t normalTexture;
  float normalTextureScale;
  float transmissionFactor;
  int32_t transmissionTexture;
  float ior;
  int32_t alphaMode;
  float alphaCutoff;
  int32_t pad;
} rt_material;

/* host_device.h:207-238 — 100 B push constant, passed by value to every stage */
typedef struct {
  int32_t frame;
  int32_t maxDepth;
  int32_t modulate;
  float fireflyClampThreshold;
  float hdrMultiplier;
  int32_t debugging_mode;
  float environmentProb;
  uint32_t time; /* RNG seed input; wall-clock ms in the reference (sample_example.cpp:402), explicit here */
  int32_t ReSTIRState;
  int32_t RISSampleNum;
  int32_t reservoirClamp;
  int32_t accumulate;
  rt_ivec2 size;
  float envMapLuminIntegInv;
  float lightLuminIntegInv;
  int32_t MIS;
  float sigLuminDirect;
  float sigNormalDirect;
  float sigDepthDirect;
  int32_t denoise;
  float sigLuminIndirect;
  float sigNormalIndirect;
  float sigDepthIndirect;
  int32_t denoiseLevel;
} rt_state;

/* host_device.h:260-264 — 28 B */
typedef struct { rt_vec3 Li; rt_vec3 wi; float dist; } rt_light_sample;
/* host_device.h:266-271 — 64 B */
typedef struct { rt_vec3 L; rt_vec3 xv, nv; rt_vec3 xs, ns; float pHat; } rt_gi_sample;
/* host_device.h:273-277 — 36 B */
typedef struct { rt_light_sample lightSample; uint32_t num; float weight; } rt_direct_reservoir;
/* host_device.h:279-284 — 76 B */
typedef struct { rt_gi_sample giSample; uint32_t num; float weight; float bigW; } rt_indirect_reservoir;

/* host_device.h:287-293 — 16 B */
typedef struct { int32_t alias; float q; float pdf; float aliasPdf; } rt_impt_samp;

/* host_device.h:295-312 — 80 B */
typedef struct {
  int32_t type; rt_vec3 direction;
  float intensity; rt_vec3 color;
  rt_vec3 position; float range;
  float outerConeCos; float innerConeCos; rt_vec2 padding;
  rt_impt_samp impSamp;
} rt_punc_light;

/* host_device.h:314-325 — 96 B */
typedef struct {
  uint32_t matIndex; uint32_t transformIndex;
  rt_vec3 v0, v1, v2;
  rt_vec2 uv0, uv1, uv2;
  rt_impt_samp impSamp;
  rt_vec3 pad;
} rt_trig_light;

/* host_device.h:336-351 — 48 B; push constant of post.frag together with debugging_mode (render_output.hpp:40-43) */
typedef struct {
  float brightness, contrast, saturation, vignette;
  float avgLum, zoom; rt_vec2 renderingRatio;
  int32_t autoExposure; float Ywhite, key; int32_t pad;
} rt_tonemapper;

/* host_device.h:353-377 — 96 B; the uniform block `_sunAndSky` (layouts.glsl:53), defaults sample_example.hpp:186-203 */
typedef struct {
  rt_vec3 rgb_unit_conversion; float multiplier;
  float haze, redblueshift, saturation, horizon_height;
  rt_vec3 ground_color; float horizon_blur;
  rt_vec3 night_color; float sun_disk_intensity;
  rt_vec3 sun_direction; float sun_disk_scale;
  float sun_glow_intensity; int32_t y_is_up; int32_t physically_scaled_sun; int32_t in_use;
} rt_sun_and_sky;

/* host_device.h:327-333 — 16 B */
typedef struct { uint32_t puncLightSize; uint32_t trigLightSize; float trigSampProb; int32_t pad; } rt_light_buf_info;

/* ------------------------------------------------------------------------------------------------
 * Scene description handed to rt_upload_scene.  The reference hands the same content to Vulkan as
 * one vertex + one index VkBuffer per glTF primitive mesh (scene.cpp:209-289), an InstanceData table
 * with buffer device addresses (host_device.h:242-247, scene.cpp:179-195), one TLAS instance per node
 * (accelstruct.cpp:132-162), a material SSBO, a texture array, light SSBOs and the env texture + alias
 * table (hdr_sampling.cpp:56-99).  Device addresses become offsets into two shared arrays.
 * All arrays are copied by rt_upload_scene; the caller keeps ownership.
 * ---------------------------------------------------------------------------------------------- */

/* one per glTF primitive mesh == one reference BLAS + one InstanceData row (index = instanceCustomIndex) */
typedef struct {
  uint32_t vertexOffset; /* into rt_scene_desc.vertices */
  uint32_t vertexCount;
  uint32_t firstIndex;   /* into rt_scene_desc.indices; indices are relative to vertexOffset */
  uint32_t indexCount;   /* multiple of 3 */
  int32_t materialIndex;
} rt_prim_mesh;

/* accelstruct.cpp:145-149 */
#define RT_INST_FORCE_OPAQUE 1u /* VK_GEOMETRY_INSTANCE_FORCE_OPAQUE_BIT_KHR */
#define RT_INST_CULL_DISABLE 2u /* VK_GEOMETRY_INSTANCE_TRIANGLE_FACING_CULL_DISABLE_BIT_KHR */

/* one per drawable glTF node == one reference TLAS instance */
typedef struct {
  float objectToWorld[12]; /* 3 rows x 4 cols, row-major (VkTransformMatrixKHR, accelstruct.cpp:152) */
  uint32_t primMesh;       /* instanceCustomIndex */
  uint32_t flags;          /* RT_INST_* */
} rt_instance;

/* glTF sampler enums kept as glTF integers (scene.cpp:513-548) */
#define RT_WRAP_REPEAT 10497
#define RT_WRAP_CLAMP 33071
#define RT_WRAP_MIRROR 33648
#define RT_FILTER_NEAREST 9728
#define RT_FILTER_LINEAR 9729

/* VK_FORMAT_B8G8R8A8_UNORM texels, LOD 0 only (scene.cpp:559, 598 — mip generation is commented out) */
typedef struct {
  const uint8_t* bgra8;
  int32_t width, height;
  int32_t wrapS, wrapT; /* RT_WRAP_* */
  int32_t magFilter;    /* RT_FILTER_*; compute shaders sample LOD 0 => mag filter applies */
  int32_t pad;
} rt_texture;

typedef struct {
  uint32_t numPrimMeshes;  const rt_prim_mesh* primMeshes;
  uint64_t numVertices;    const rt_vertex* vertices;
  uint64_t numIndices;     const uint32_t* indices;
  uint32_t numInstances;   const rt_instance* instances;
  uint32_t numMaterials;   const rt_material* materials;
  uint32_t numTextures;    const rt_texture* textures;
  const rt_punc_light* puncLights;  /* lightInfo.puncLightSize entries (may be NULL when 0) */
  const rt_trig_light* trigLights;  /* lightInfo.trigLightSize entries (may be NULL when 0) */
  rt_light_buf_info lightInfo;
  int32_t envWidth, envHeight;      /* RGBA32F lat-long map, U repeat / V clamp, bilinear (hdr_sampling.cpp:69-77) */
  const float* envRgba32f;          /* envWidth*envHeight*4 floats; NULL => 1x1 black */
  const rt_impt_samp* envAccel;     /* envWidth*envHeight entries (hdr_sampling.cpp:181-242); NULL when no env */
} rt_scene_desc;

/* ------------------------------------------------------------------------------------------------
 * Screen-space buffers owned by the context (renderer.cpp:227-302, render_output.cpp:82-148).
 * Reference element layouts are kept at the boundary (readback/upload use them verbatim).
 * ---------------------------------------------------------------------------------------------- */
typedef enum {
  RT_BUF_GBUFFER0 = 0,        /* RGBA32UI 16 B/px (renderer.hpp:88)              */
  RT_BUF_GBUFFER1 = 1,
  RT_BUF_MOTION = 2,          /* RG16_SINT 4 B/px (renderer.hpp:94)               */
  RT_BUF_DIRECT_RESV0 = 3,    /* rt_direct_reservoir 36 B/px (renderer.cpp:229)   */
  RT_BUF_DIRECT_RESV1 = 4,
  RT_BUF_DIRECT_RESV_TEMP = 5,
  RT_BUF_INDIRECT_RESV0 = 6,  /* rt_indirect_reservoir 76 B/half-px               */
  RT_BUF_INDIRECT_RESV1 = 7,
  RT_BUF_INDIRECT_RESV_TEMP = 8,
  RT_BUF_DENOISE_DIR_A = 9,   /* RGBA32F 16 B/px (renderer.cpp:267-285)           */
  RT_BUF_DENOISE_DIR_B = 10,
  RT_BUF_DENOISE_IND_A = 11,  /* RGBA32F, allocated full-res, top-left quarter used */
  RT_BUF_DENOISE_IND_B = 12,
  RT_BUF_DIRECT_RESULT0 = 13, /* RGBA32F 16 B/px (render_output.cpp:104-133)      */
  RT_BUF_DIRECT_RESULT1 = 14,
  RT_BUF_INDIRECT_RESULT0 = 15,
  RT_BUF_INDIRECT_RESULT1 = 16,
  RT_BUF_LIGHT_ID0 = 17,      /* u32/px, ping-pong like the reservoirs: id of the light sample held by the direct
                                 reservoir (0x80000000|env texel, 0x40000000|triangle light, 0x20000000|punctual light,
                                 0xffffffff none).  Added for the "reservoir sample indices bit-exact" check of
                                 BASELINE.json; the reference stores the sample itself, not an index. */
  RT_BUF_LIGHT_ID1 = 18,
  RT_BUF_LDR = 19,            /* RGBA8 UNORM 4 B/px (R in the low byte): output of rt_tonemap = the swapchain image post.frag draws */
  RT_BUF_COUNT = 20
} rt_buffer_id;

/* per-frame stage selector for rt_run_stage — the dispatch list of renderer.cpp:163-205 */
typedef enum {
  RT_STAGE_DIRECT = 0,           /* direct_stage.comp  (live); rt_run_stage level 1 / 2: only the first / second half of the spatial-reuse modes (see rt_run_stage) */
  RT_STAGE_INDIRECT = 1,         /* indirect_stage.comp, half resolution         */
  RT_STAGE_DENOISE_DIRECT = 2,   /* denoise_direct.comp, level 0..3              */
  RT_STAGE_DENOISE_INDIRECT = 3, /* denoise_indirect.comp, level 0..4            */
  RT_STAGE_COMPOSE = 4,          /* compose.comp                                 */
  RT_STAGE_DIRECT_GEN = 5,       /* direct_gen.comp   (compiled, not dispatched: renderer.cpp:166-168) */
  RT_STAGE_DIRECT_REUSE = 6,     /* direct_reuse.comp (compiled, not dispatched: renderer.cpp:170-171) */
  RT_STAGE_COUNT = 7
} rt_stage_id;

typedef struct {
  uint64_t closestHitRays;   /* ClosestHit() invocations (traceray_rq.glsl:108)  */
  uint64_t anyHitRays;       /* AnyHit() invocations (traceray_rq.glsl:153)      */
  uint64_t nodesVisited;     /* BVH8 nodes fetched (80 B each)                   */
  uint64_t trisTested;       /* triangle records fetched (48 B + 8 B each)       */
  uint64_t hitsShaded;       /* GetState() invocations (108 B gather + 80 B material) */
  uint64_t risCandidates;    /* SampleDirectLightNoVisibility() invocations      */
  /* HIP-event timings on the ctx stream, ACCUMULATED over the rt_render_frame calls since the last rt_set_counting()
   * (which resets them); denoise entries sum their 4 / 5 levels. Divide by framesTimed for per-frame averages. */
  float stageMs[RT_STAGE_COUNT];
  float frameMs;
  uint32_t framesTimed;
  /* SIMD efficiency of the traversal loops (counting mode): summed over lanes, the scheduling rounds a lane sat through while
   * its wave was traversing (laneRounds) and the rounds in which it still had a ray to advance (laneLiveRounds).
   * (nodesVisited + trisTested) / laneLiveRounds = share of live lanes the majority vote lets step;
   * laneLiveRounds / laneRounds = share of lanes that are not just waiting for the slowest ray of their wave. */
  uint64_t laneRounds, laneLiveRounds;
} rt_counters;

typedef enum {
  RT_OK = 0,
  RT_ERR_INVALID_ARG = -1,
  RT_ERR_NO_DEVICE = -2,
  RT_ERR_HIP = -3,
  RT_ERR_NO_SCENE = -4,
  RT_ERR_NO_ACCEL = -5,
  RT_ERR_NO_TARGET = -6,
  RT_ERR_OOM = -7
} rt_status;

typedef struct rt_ctx rt_ctx;

/* Renderer::setup (renderer.cpp:62-73): bind to HIP device `device`, create the stream + events. */
int rt_create(rt_ctx** out, int device);
/* Renderer::destroy + Scene::destroy + AccelStructure::destroy (renderer.cpp:75-91). */
int rt_destroy(rt_ctx* ctx);
/* Run every later launch on a caller-owned hipStream_t (NULL = ctx-owned stream). Lets the caller
 * order RCCL halo exchanges with the stages without host syncs. */
int rt_set_stream(rt_ctx* ctx, void* hipStream);
/* Scene::load's buffer uploads (scene.cpp:94-112) + HdrSampling::loadEnvironment upload (hdr_sampling.cpp:79-95).
 * Every value the kernels use as an array index (vertex indices, material / texture ids, light material ids, alias-table
 * entries) is range-checked here: RT_ERR_INVALID_ARG instead of a device fault.  (The reference trusts nvh::GltfScene.) */
int rt_upload_scene(rt_ctx* ctx, const rt_scene_desc* scene);
/* AccelStructure::create (accelstruct.cpp:55-65): host-built flat BVH8 over world-space triangles. */
int rt_build_accel(rt_ctx* ctx);
/* Renderer::create / Renderer::update (renderer.cpp:97-148, 209-225): (re)allocate all screen-space buffers, zeroed. */
int rt_resize(rt_ctx* ctx, int width, int height);
/* Scene::updateCamera's vkCmdUpdateBuffer (scene.cpp:777-811). */
int rt_set_camera(rt_ctx* ctx, const rt_scene_camera* cam);
/* Renderer::run (renderer.cpp:154-206): the 12 dispatches of one frame. `frames` selects the ping-pong
 * side exactly like m_descSet[(frames+1)%2] (renderer.cpp:157, 346-356): this = [frames&1], last = [(frames+1)&1]. */
int rt_render_frame(rt_ctx* ctx, const rt_state* state, int frames);
/* One dispatch of the list above, restricted to pixel rows [rowBegin,rowEnd) of the stage's own grid
 * (full-res rows for direct/denoise_direct/compose, half-res rows for indirect/denoise_indirect).
 * rowEnd <= 0 means "all rows".  Used for row-tiled multi-GPU frames.
 * `level`: the a-trous level for the two denoise stages.  For RT_STAGE_DIRECT with ReSTIRState eSpatial / eSpatiotemporal it selects
 * the half of the stage: 0 = both, 1 = everything up to cacheTempReservoir (direct_stage.comp:224-235), 2 = the neighbour merges and
 * the shading (:236-262) — a row-tiled host runs 1, exchanges the neighbouring rows of RT_BUF_DIRECT_RESV_TEMP, then runs 2. */
int rt_run_stage(rt_ctx* ctx, const rt_state* state, int frames, int stage, int level, int rowBegin, int rowEnd);
/* Copy a screen-space buffer to / from host memory in the reference element layout. Synchronous. */
int rt_readback(rt_ctx* ctx, int buffer, void* dst, size_t bytes);
int rt_upload_history(rt_ctx* ctx, int buffer, const void* src, size_t bytes);
/* Size in bytes of a buffer's boundary layout at the current resolution. */
size_t rt_buffer_bytes(rt_ctx* ctx, int buffer);
/* Raw device pointer of a buffer's device-side storage, its ALLOCATED byte size and the bytes per row of the stage
 * grid, for RCCL exchanges by the caller.  Device-side layout == boundary layout; every allocation carries 128 rows
 * (64 for the half-resolution reservoirs) of slack behind the image so equal-height row bands of up to 8 ranks can be
 * all-gathered in place. */
int rt_device_ptr(rt_ctx* ctx, int buffer, void** ptr, size_t* bytes, size_t* rowPitch);
/* Enable (1) / disable (0) traversal + gather counting for subsequent frames (adds atomics to the kernels), and reset
 * every counter and accumulated timing. */
int rt_set_counting(rt_ctx* ctx, int enable);
int rt_get_counters(rt_ctx* ctx, rt_counters* out);
/* Row-tiled multi-GPU: declare which rows [row0,row1) of the LAST-frame buffers (G-buffer, reservoirs, light ids; full-res
 * rows, the half-res reservoirs use row/2) hold valid data on this GPU — its own band plus the halo rows received from its
 * neighbours.  Temporal reuse that reprojects inside the image but outside this range sets a flag instead of silently
 * reading stale rows; rt_history_miss() returns and clears the flag (synchronises the ctx stream) so the caller can
 * gather the full history and redo the frame.  Default: every row is valid. */
int rt_set_history_rows(rt_ctx* ctx, int row0, int row1);
int rt_history_miss(rt_ctx* ctx, int* missed);
/* The same flag per stage kind, for hosts that keep the direct stage of frame f+1 in flight beside the indirect stage of
 * frame f on different streams: RT_STAGE_DIRECT* launches raise flag 0, RT_STAGE_INDIRECT launches raise flag 1;
 * this call waits for the ctx stream only, then reads and clears the flag of `stage`.  (rt_history_miss = both flags.) */
int rt_history_miss_stage(rt_ctx* ctx, int stage, int* missed);
/* Stage-wise submission with frames in flight (what rt_render_frame does internally in overlap mode 2): before submitting the
 * stages of frame `frames`, swap RT_BUF_GBUFFER0 + (frames & 1) and RT_BUF_MOTION with their spare buffers so that this
 * frame's direct stage does not overwrite the G-buffer / motion vectors that the previous frame's indirect stage, possibly
 * still running on another stream, reads.  Calling it a second time with the same `frames` undoes the swap.
 * Invalidates rt_device_ptr results for those three buffers. */
int rt_rotate_buffers(rt_ctx* ctx, int frames);
/* RT_BUF_DENOISE_IND_A (the noisy indirect colour: indirect stage -> its five filter levels) exists once per frame parity, so that the indirect stage of frame
 * f+1 does not wait for the filters of frame f.  The id names the buffer of the frame most recently passed to rt_render_frame / rt_run_stage / this call:
 * a host that touches the buffer itself (rt_device_ptr for a halo exchange) of a frame other than the last one it launched selects that frame first. */
int rt_select_frame(rt_ctx* ctx, int frames);
/* RenderOutput::run (render_output.cpp:224-237) + post.frag:103-175 as a compute pass: reads the two result images of
 * frame `frames` (RT_BUF_DIRECT_RESULT0/INDIRECT_RESULT0 + (frames & 1)), applies auto-exposure (tonemapping by the image
 * mean, post.frag:133-153), the Uncharted-2 tone curve (tonemapping.glsl:48-66), dithering (post.frag:50-55), contrast /
 * brightness / saturation / vignette (post.frag:163-171) or the debug views (post.frag:106-118) and writes RT_BUF_LDR.
 * Differences from the fragment shader, all documented in DESIGN.md: the image mean replaces the driver-generated mip
 * pyramid's top level; tm.zoom samples the nearest texel.  The "local" auto-exposure bit (autoExposure & 2, toneLocalExposure,
 * post.frag:70-101) samples mip levels 0..7 of the two result images: the pyramid is built level by level with the linear-blit
 * formula of RenderOutput::genMipmap (render_output.cpp:243-254; 2x2 box for even sizes) and sampled bilinearly at the fragment;
 * the variable the reference leaves uninitialised in the default view (`v2 ==` at post.frag:91) is 0. */
int rt_tonemap(rt_ctx* ctx, const rt_tonemapper* tm, int debugging_mode, int frames);
/* The ray query on its own, for tests and tools: n rays (8 floats each: origin xyz, direction xyz, tmax, the bits of the ray's seed — the seed only enters
 * the stochastic alpha test) through ClosestHit (anyHit = 0: out = 4 floats per ray, t | bits of the flattened triangle index (0xffffffff: miss) | u | v;
 * tmax is ignored, the range is (0, 1e28)) or AnyHit (anyHit = 1: out[4 i] = 1 if anything accepts inside (0, tmax), else 0) of traceray_rq.glsl:108-185,
 * with the traversal the stage kernels use.  Host arrays in, host arrays out. */
int rt_trace_rays(rt_ctx* ctx, int n, const float* rays, float* out, int anyHit);
/* Which build of the ray-traced kernels (direct_stage / direct_gen / indirect_stage) a launch gets.  There are two, with identical
 * results: THROUGHPUT (majority-vote traversal rounds, 4-5 waves per SIMD — a full frame is bound by instruction issue) and LATENCY
 * (every ray advances every round, node fetches overlapped with the triangle work, register budget of 2 waves per SIMD — a row band
 * of a multi-GPU frame or a small image is bound by the dependent memory accesses of its slowest wave).  AUTO (default) picks per
 * launch from its size.  ABI 2.0: replaces the pipeline switch of ABI 1.0, whose second (queue-based) kernel organisation was removed. */
enum { RT_TRAVERSAL_AUTO = 0, RT_TRAVERSAL_THROUGHPUT = 1, RT_TRAVERSAL_LATENCY = 2 };
int rt_set_traversal(rt_ctx* ctx, int mode);
/* SampleExample::screenPicking (sample_example.cpp:456-497): nvvk::RayPickerKHR — one camera ray through the normalised
 * window position (pickX, pickY in [0,1]) built like raySpawn (origin = modelViewInv * (0,0,0,1), direction = modelViewInv *
 * normalize(perspectiveInv * (2*pick-1, 1, 1))), traced with the path's ClosestHit rules.  The result mirrors
 * nvvk::RayPickerKHR::PickResult; instanceID == -1 => nothing hit. */
typedef struct {
  rt_vec4 worldRayOrigin, worldRayDirection;
  float hitT; int32_t primitiveID; int32_t instanceID; int32_t instanceCustomIndex;
  rt_vec3 baryCoord;
} rt_pick_result;
int rt_pick(rt_ctx* ctx, const rt_mat4* modelViewInv, const rt_mat4* perspectiveInv, float pickX, float pickY, rt_pick_result* out);
/* SampleExample::updateUniformBuffer's vkCmdUpdateBuffer(m_sunAndSkyBuffer, ...) (sample_example.cpp:172): the procedural
 * sun & sky environment.  With in_use == 1 EnvRadiance / EnvSample / EnvEval (pathtrace.glsl:40-72, env_sampling.glsl:105-135)
 * evaluate sun_and_sky() (sun_and_sky.glsl:453-601) instead of the HDR map.  Default: in_use = 0. */
int rt_set_sun_and_sky(rt_ctx* ctx, const rt_sun_and_sky* ss);
/* Stream-level concurrency of rt_render_frame (results are identical in every mode):
 *   0 = every launch of Renderer::run's list (renderer.cpp:163-205) in order on the ctx stream;
 *   1 = the direct A-Trous chain runs beside the indirect stage on an internal stream and joins before compose;
 *   2 = (default) 1 + frames in flight: the call returns after enqueueing and the next frame's direct stage runs beside
 *       this frame's indirect stage and filters (internally triple-buffered G-buffer, double-buffered motion vectors).
 *       Throughput mode of a renderer that keeps submitting; one frame's latency is higher than in mode 1.
 *   3 = 2 with a third frame in flight (four G-buffers, three motion buffers, three direct images): direct(f) waits for frame f-3 instead of f-2, which takes
 *       the filter chain of frame f-2 off the path to direct(f).  Measured, not the default (profiles/r06_three_frames_ab.txt).
 * rt_sync / rt_readback / rt_get_counters wait for everything in flight.  In modes 2 and 3, rt_device_ptr results for the
 * G-buffers and the motion buffer (mode 3: and the direct result images) are invalidated by rt_render_frame (rt_run_stage never rotates buffers).
 * Also settable with RESTIR_OVERLAP=0|1|2|3 before rt_create. */
int rt_set_overlap(rt_ctx* ctx, int mode);
/* Priorities of the indirect-stage stream and the filter stream of mode 2 (levels: -1 low, 0 normal, +1 high).  The reference submits its dispatches to ONE queue
 * (src/renderer.cpp:154-206) and has no such choice; here three streams share the chip and the fastest setting depends on the workload
 * (profiles/r05_prio_by_config_ab.txt).  Unset (no call, no RESTIR_PRIO), the context decides on its first three mode-2 frames ("probe frames": every stage alone on
 * the main stream, timed): the filter stream becomes high next to the indirect stream when filters / (direct + indirect) of the LAST probe frame — warm caches, warm
 * history — is >= 0.30 (profiles/r06_prio_rule.txt); the two streams are created afterwards.  rt_resize, a new scene / tree and a denoise toggle re-open that decision (round 6; round 5 decided once,
 * on the cold first frame).  An explicit call is best made before the first frame: a stream created after others exist may share a hardware queue with them; it
 * stands for the context's lifetime.  Results are identical under every setting.  Drains the context. */
int rt_set_stream_priorities(rt_ctx* ctx, int indirectLevel, int filterLevel);
/* The levels in use; filterShare = the last probe frame's filters / (direct + indirect) (-1: not measured: no frame yet, or the levels were given); decided = 0 while a
 * context that decides for itself is still probing.  Any pointer may be NULL. */
int rt_get_stream_priorities(rt_ctx* ctx, int* indirectLevel, int* filterLevel, float* filterShare, int* decided);
/* The context's three streams of the frames-in-flight schedule (hipStream_t; owned by the context), for a host that issues the stages itself through rt_run_stage +
 * rt_set_stream (restir_amd/tiled.py wraps them in torch.cuda.ExternalStream; csrc/mgpu.cpp uses them per rank).  The reference has ONE queue (src/renderer.cpp:154-206);
 * here a host that brought its own streams (torch's pool of 32, created after RCCL's) ran on a stream layout the schedule was never measured on.  Creates the filter,
 * then the indirect stream with the current levels if they do not exist (only when one of the two is asked for: with both NULL nothing is created); call before anything
 * else in the process creates streams.  Any pointer may be NULL. */
int rt_get_streams(rt_ctx* ctx, void** mainStream, void** indirectStream, void** filterStream);
/* created = HIP streams this library has created in the process so far; index[0..2] = creation index of this context's main / indirect / filter stream (-1: none yet). */
int rt_get_stream_layout(rt_ctx* ctx, int* created, int index[3]);
/* ------------------------------------------------------------------------------------------------------------------
 * Multi-GPU context (csrc/mgpu.cpp): the row-tiled frame of BASELINE.json / SURVEY.md §8(e) for a host that owns all the
 * GPUs of the node from ONE process — "multi-GPU ctx internally drives 8 streams" (§8b threading row).  One worker thread,
 * one rt_ctx and one HIP stream per device; halos and history rows move with hipMemcpyPeerAsync over xGMI; band heights
 * are cost weighted from the ranks' measured stage times.  Call order == the single-GPU context's:
 *     rt_mgpu_create -> rt_mgpu_upload_scene (upload + BVH8 build on every device) -> rt_mgpu_resize
 *     per frame: rt_mgpu_set_camera, rt_mgpu_render_frame
 *     rt_mgpu_readback assembles a buffer of the last frame from the ranks that own its rows (same layout as rt_readback).
 * Frames in flight (default, rt_mgpu_set_pipeline): rt_mgpu_render_frame QUEUES the frame (at most two ahead) and returns; every rank runs the three-stream
 * schedule of rt_render_frame and every cross-rank dependency is a HIP event another rank's stream waits for — no host barrier per frame.  Frames must be
 * consecutive (`frames` + 1 per call); rt_mgpu_sync / rt_mgpu_readback / rt_mgpu_get_stats finish what is in flight first.  With rt_mgpu_set_pipeline(0),
 * in the spatial-reuse modes and under rt_mgpu_set_serialize the call returns when the frame is complete on every device (barrier schedule).
 * Every output is bit-identical to the single-GPU frame, in every ReSTIRState (the spatial-reuse modes run the direct stage in two
 * halves around an exchange of the cached reservoirs' boundary rows).  `devices` may name the same device more than once (used by the
 * tests on a one-GPU machine).  The caller's thread discipline is the reference's: one thread issues the calls.
 * (One-process-per-GPU hosts use restir_amd/tiled.py over torch.distributed / RCCL instead.)
 * ---------------------------------------------------------------------------------------------------------------- */
#define RT_MGPU_MAX_RANKS 16
typedef struct rt_mgpu rt_mgpu;
typedef struct {
  int32_t numRanks;
  uint32_t frames;                               /* frames rendered since creation */
  uint32_t historyFallbacks;                     /* frames whose temporal reuse left band + halo (full history pulled, stages re-run) */
  uint32_t pad;
  uint64_t haloBytes;                            /* bytes pulled from peers during the last frame, all ranks */
  int32_t bandBegin[RT_MGPU_MAX_RANKS], bandEnd[RT_MGPU_MAX_RANKS]; /* full-res row range of every rank in the last frame */
  float tracedMs[RT_MGPU_MAX_RANKS];             /* direct + indirect stage of the last frame, HIP events on the rank's stream */
  float filterMs[RT_MGPU_MAX_RANKS];             /* 4 + 5 A-Trous passes + compose */
  uint64_t haloBytesKind[6];                     /* haloBytes by purpose: history rows | filter halos | spatial-reuse rows | rows that changed owner | display gather | fallback pulls */
  uint64_t haloBytesRankKind[RT_MGPU_MAX_RANKS][6]; /* the same per rank (bytes that rank pulled for its latest complete frame) */
} rt_mgpu_stats;
int rt_mgpu_create(rt_mgpu** out, int numRanks, const int* devices /* NULL: devices 0..numRanks-1 */);
int rt_mgpu_destroy(rt_mgpu* m);
int rt_mgpu_upload_scene(rt_mgpu* m, const rt_scene_desc* scene);
int rt_mgpu_resize(rt_mgpu* m, int width, int height);
int rt_mgpu_set_camera(rt_mgpu* m, const rt_scene_camera* cam);
int rt_mgpu_render_frame(rt_mgpu* m, const rt_state* state, int frames);
int rt_mgpu_readback(rt_mgpu* m, int buffer, void* dst, size_t bytes);
int rt_mgpu_sync(rt_mgpu* m);
int rt_mgpu_set_balance(rt_mgpu* m, int mode);    /* 1 (default): cost-weighted band heights; 0: equal heights; 2: freeze the current partition */
int rt_mgpu_set_bands(rt_mgpu* m, const int* bands /* numRanks + 1 row boundaries, multiples of 16, 0 .. height */);   /* explicit partition; freezes it */
int rt_mgpu_set_serialize(rt_mgpu* m, int on);    /* measurement aid when several ranks share ONE device: ranks take turns on the GPU */
int rt_mgpu_set_pipeline(rt_mgpu* m, int on);     /* 1 (default): frames in flight per rank, event-ordered pulls; 0: barrier schedule */
int rt_mgpu_set_gather(rt_mgpu* m, int on);       /* 1 (default): rank 0 pulls the result bands every frame (display rank); 0: results stay with their owners */
int rt_mgpu_set_solo(rt_mgpu* m, int rank);       /* measurement aid: only `rank` renders (its pulls read the idle ranks' stale rows): the PERIOD of one rank on a GPU to itself; -1 = off */
int rt_mgpu_get_stats(rt_mgpu* m, rt_mgpu_stats* out);
/* Link statistics of the latest complete frame of the frames-in-flight schedule (ABI 2.1): every rank's four pull groups — 0 history rows (main stream),
 * 1 indirect-reservoir history (indirect stream), 2 direct filter halo, 3 indirect filter halo (filter stream) — timed with HIP events on the stream that
 * carried them, and their bytes; the device of every rank; peerAccess[puller][owner] = 1 when the puller's device has direct peer access to the owner's
 * (xGMI on an MI355X node; 0 = staged by the runtime).  This is the measured point bench.py's xgmi_model stands in for on a one-GPU box.
 * pullBytes / pullMs of a group are a pair of ONE frame (a frame whose events cannot be read yet leaves the previous pair standing); the barrier schedule
 * (rt_mgpu_set_pipeline(0), serialize, spatial modes) does not bracket its pulls: zeros. */
typedef struct {
  int32_t numRanks;
  int32_t devices[RT_MGPU_MAX_RANKS];
  uint8_t peerAccess[RT_MGPU_MAX_RANKS][RT_MGPU_MAX_RANKS];
  float pullMs[RT_MGPU_MAX_RANKS][4];
  uint64_t pullBytes[RT_MGPU_MAX_RANKS][4];
} rt_mgpu_link_stats;
int rt_mgpu_get_link_stats(rt_mgpu* m, rt_mgpu_link_stats* out);
/* rt_get_stream_layout of every rank's context: created = streams the library has created in the process; index[3 * rank + {0, 1, 2}] = creation index of that rank's
 * main / indirect / filter stream (-1: not created yet — a rank's two extra streams are created on its first frame in flight); levels[3] = their priority levels. */
int rt_mgpu_get_stream_layout(rt_mgpu* m, int* created, int* index, int levels[3]);
/* The partition rule on its own (pure host arithmetic): boundaries (numRanks + 1 rows, multiples of 16) that equalise the summed cost of the
 * 16-row stripes; with prevBands a boundary moves at most maxMoveStripes stripes (< 0: unlimited); every rank keeps >= one stripe. */
int rt_mgpu_plan_bands(int height, int numRanks, const float* stripeCost, const int* prevBands, int maxMoveStripes, int* outBands);
const char* rt_mgpu_last_error(rt_mgpu* m);

/* Measured VALU issue ceiling of the device the ctx lives on (csrc/microbench.hip): wave-level VALU instructions per second of a
 * chain-free loop.  variant 0 = the instruction mix of the traced kernels, 1 = v_fma_f32 only; wavesPerSimd in 1..8.
 * bench.py prices `roofline.valu` against this measurement instead of an assumed cycles-per-instruction figure.
 * (Introspection like rt_get_counters; no counterpart in the reference.) */
int rt_measure_valu_peak(rt_ctx* ctx, int variant, int wavesPerSimd, double* waveInstPerSec);
/* Wait for all work on the ctx stream. */
int rt_sync(rt_ctx* ctx);
/* Last error message of this ctx (or of rt_create when ctx == NULL). Never NULL. */
const char* rt_last_error(rt_ctx* ctx);
/* ABI version: (major<<16)|minor.  2.3 (round 6): + rt_get_streams, rt_get_stream_layout, rt_mgpu_get_stream_layout; the priority rule probes three frames and re-opens on resize / scene / denoise.  2.2 (round 5): + rt_set_stream_priorities, rt_get_stream_priorities.  2.1 (round 4): + rt_mgpu_get_link_stats.  2.0 (round 3): rt_set_pipeline -> rt_set_traversal; RT_STAGE_DIRECT levels 1 / 2 are rejected outside the
 * spatial modes; 1.1 would have been round 2's additions (rt_mgpu_*, rt_measure_valu_peak, the `level` halves of RT_STAGE_DIRECT). */
#define RT_ABI_VERSION_MAJOR 2u
#define RT_ABI_VERSION_MINOR 3u
uint32_t rt_abi_version(void);

#ifdef __cplusplus
} /* extern "C" */

static_assert(sizeof(rt_scene_camera) == 336, "SceneCamera host_device.h:153-165");
static_assert(sizeof(rt_vertex) == 32, "VertexAttributes host_device.h:167-174");
static_assert(sizeof(rt_material) == 80, "GltfShadeMaterial host_device.h:183-204");
static_assert(sizeof(rt_state) == 100, "RtxState host_device.h:207-238");
static_assert(sizeof(rt_direct_reservoir) == 36, "DirectReservoir host_device.h:273-277");
static_assert(sizeof(rt_indirect_reservoir) == 76, "IndirectReservoir host_device.h:279-284");
static_assert(sizeof(rt_impt_samp) == 16, "ImptSampData host_device.h:287-293");
static_assert(sizeof(rt_punc_light) == 80, "PuncLight host_device.h:295-312");
static_assert(sizeof(rt_trig_light) == 96, "TrigLight host_device.h:314-325");
static_assert(sizeof(rt_light_buf_info) == 16, "LightBufInfo host_device.h:327-333");
static_assert(sizeof(rt_tonemapper) == 48, "Tonemapper host_device.h:336-351");
static_assert(sizeof(rt_sun_and_sky) == 96, "SunAndSky host_device.h:353-377");
#endif

#endif /* RT_ABI_H */
