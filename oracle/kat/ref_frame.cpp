// ref_frame.cpp — frame driver for the reference's own compute shaders compiled on the CPU (TEST INFRASTRUCTURE, authoring
// container only; output oracle/_ref/libref_stages.so).  It plays the part of src/renderer.cpp + the Vulkan driver:
//   * owns the screen-space resources of renderer.cpp:227-302 / render_output.cpp:82-148 with the reference element layouts,
//   * binds "this"/"last" by frame parity exactly like m_descSet[(frames + 1) % 2] (renderer.cpp:157, 346-356),
//   * records the dispatch list of Renderer::run (renderer.cpp:154-206) — direct, indirect (half res), 4 + 5 filter passes with
//     denoiseLevel re-pushed, compose — as calls into the per-shader translation units (ref_tu.cpp).
// The scene side (flattened triangles, instance transforms and their inverses, texture filtering) is the oracle's orc::Scene:
// that is the part the reference hands to VK_KHR_ray_query / Vulkan samplers / nvpro_core and it stays unpinned.
// The C entry points mirror oracle/orc_capi.cpp so the tests can drive both through the same Python surface.
#include "glsl_cpu.h"
#include <memory>
namespace glsl {
#include "host_device.h"
#include "ref_ctx.h"
thread_local uint* rq_seed = nullptr;
thread_local uvec3 gl_GlobalInvocationID, gl_LocalInvocationID, gl_WorkGroupID;
thread_local uint gl_LocalInvocationIndex = 0;
uint rq_candidate_seed(uint raySeed, uint tri) { return orc::Scene::candidateSeed(raySeed, tri); }
static_assert(sizeof(RtxState) == sizeof(rt_state) && sizeof(SceneCamera) == sizeof(rt_scene_camera) && sizeof(GltfShadeMaterial) == sizeof(rt_material) &&
              sizeof(VertexAttributes) == sizeof(rt_vertex) && sizeof(DirectReservoir) == sizeof(rt_direct_reservoir) &&
              sizeof(IndirectReservoir) == sizeof(rt_indirect_reservoir) && sizeof(PuncLight) == sizeof(rt_punc_light) &&
              sizeof(TrigLight) == sizeof(rt_trig_light) && sizeof(ImptSampData) == sizeof(rt_impt_samp) && sizeof(SunAndSky) == sizeof(rt_sun_and_sky) &&
              sizeof(LightBufInfo) == sizeof(rt_light_buf_info), "rt_abi.h mirrors shaders/host_device.h");
}  // namespace glsl

extern "C" {
void ref_run_direct_stage(const glsl::RefCtx*, int, int);
void ref_run_direct_gen(const glsl::RefCtx*, int, int);
void ref_run_direct_reuse(const glsl::RefCtx*, int, int);
void ref_run_indirect_stage(const glsl::RefCtx*, int, int);
void ref_run_denoise_direct(const glsl::RefCtx*, int, int);
void ref_run_denoise_indirect(const glsl::RefCtx*, int, int);
void ref_run_compose(const glsl::RefCtx*, int, int);
}

namespace {
using namespace glsl;
struct Ref {
  orc::Scene scene;
  bool haveScene = false;
  int W = 0, H = 0;
  rt_scene_camera cam{};
  std::vector<InstanceData> geoInfo;
  std::vector<sampler2D> textures;
  // renderer.cpp:227-302
  std::vector<uint> gbuffer[2];
  std::vector<int16_t> motion;
  std::vector<DirectReservoir> directResv[2], directTemp;
  std::vector<IndirectReservoir> indirectResv[2], indirectTemp;
  std::vector<float> denoiseTemp[4], directResult[2], indirectResult[2];

  void resize(int w, int h)
  {
    W = w; H = h;
    const size_t n = size_t(w) * h, nh = size_t(w / 2) * (h / 2);
    for(int i = 0; i < 2; i++) {
      gbuffer[i].assign(n * 4, 0u);
      directResv[i].assign(n, DirectReservoir()); indirectResv[i].assign(nh, IndirectReservoir());
      directResult[i].assign(n * 4, 0.0f); indirectResult[i].assign(n * 4, 0.0f);
    }
    directTemp.assign(n, DirectReservoir()); indirectTemp.assign(nh, IndirectReservoir());
    motion.assign(n * 2, 0);
    for(auto& d : denoiseTemp) d.assign(n * 4, 0.0f);
  }
  RefCtx bind(const rt_state& st, int frames)
  {
    RefCtx c{};
    c.scene = &scene;
    memcpy(&c.rtxState, &st, sizeof(st));
    memcpy(&c.sceneCamera, &cam, sizeof(cam));
    memcpy(&c.sunAndSky, &scene.sunAndSky, sizeof(scene.sunAndSky));
    memcpy(&c.lightBufInfo, &scene.lightInfo, sizeof(scene.lightInfo));
    c.geoInfo = geoInfo.data();
    c.materials = reinterpret_cast<GltfShadeMaterial*>(scene.materials.data());
    c.puncLights = reinterpret_cast<PuncLight*>(scene.puncLights.data());
    c.trigLights = reinterpret_cast<TrigLight*>(scene.trigLights.data());
    c.envSamplingData = reinterpret_cast<ImptSampData*>(scene.envAccel.data());
    c.texturesMap = textures.data();
    c.environmentTexture = sampler2D{&scene, -1};
    const int cur = frames & 1, last = cur ^ 1;   // set index (frames+1)%2: last = [i], this = [!i]
    auto img = [&](std::vector<float>& v) { image2D im; im.data = v.data(); im.w = W; im.h = H; return im; };
    auto gimg = [&](std::vector<uint>& v) { uimage2D im; im.data = v.data(); im.w = W; im.h = H; return im; };
    c.lastDirectResultImage = img(directResult[last]); c.thisDirectResultImage = img(directResult[cur]);
    c.lastIndirectResultImage = img(indirectResult[last]); c.thisIndirectResultImage = img(indirectResult[cur]);
    c.lastGbuffer = gimg(gbuffer[last]); c.thisGbuffer = gimg(gbuffer[cur]);
    c.motionVector.data = motion.data(); c.motionVector.w = W; c.motionVector.h = H;
    c.lastDirectResv = directResv[last].data(); c.thisDirectResv = directResv[cur].data(); c.tempDirectResv = directTemp.data();
    c.lastIndirectResv = indirectResv[last].data(); c.thisIndirectResv = indirectResv[cur].data(); c.tempIndirectResv = indirectTemp.data();
    c.denoiseDirTempA = img(denoiseTemp[0]); c.denoiseDirTempB = img(denoiseTemp[1]);
    c.denoiseIndTempA = img(denoiseTemp[2]); c.denoiseIndTempB = img(denoiseTemp[3]);
    return c;
  }
  static int ceilDiv(int x, int y) { return (x + y - 1) / y; }   // CEIL_DIV, host_device.h:40
  void runStage(const rt_state& st, int frames, int stage, int level)
  {
    rt_state s = st; s.denoiseLevel = level;
    RefCtx c = bind(s, frames);
    const int gx = ceilDiv(st.size.x, 8), gy = ceilDiv(st.size.y, 8), hx = ceilDiv(st.size.x / 2, 8), hy = ceilDiv(st.size.y / 2, 8);
    switch(stage) {
      case RT_STAGE_DIRECT: ref_run_direct_stage(&c, gx, gy); break;
      case RT_STAGE_DIRECT_GEN: ref_run_direct_gen(&c, gx, gy); break;
      case RT_STAGE_DIRECT_REUSE: ref_run_direct_reuse(&c, gx, gy); break;
      case RT_STAGE_INDIRECT: ref_run_indirect_stage(&c, hx, hy); break;
      case RT_STAGE_DENOISE_DIRECT: ref_run_denoise_direct(&c, gx, gy); break;
      case RT_STAGE_DENOISE_INDIRECT: ref_run_denoise_indirect(&c, hx, hy); break;
      case RT_STAGE_COMPOSE: ref_run_compose(&c, gx, gy); break;
    }
  }
  void renderFrame(const rt_state& st, int frames)   // renderer.cpp:154-206
  {
    runStage(st, frames, RT_STAGE_DIRECT, st.denoiseLevel);
    runStage(st, frames, RT_STAGE_INDIRECT, st.denoiseLevel);
    if(st.denoise > 0) for(int i = 0; i < 4; i++) runStage(st, frames, RT_STAGE_DENOISE_DIRECT, i);
    if(st.denoise > 0) for(int i = 0; i < 5; i++) runStage(st, frames, RT_STAGE_DENOISE_INDIRECT, i);
    runStage(st, frames, RT_STAGE_COMPOSE, st.denoiseLevel);
  }
  void* buf(int id, size_t& bytes)
  {
    auto r = [&](auto& v) -> void* { bytes = v.size() * sizeof(v[0]); return v.data(); };
    switch(id) {
      case RT_BUF_GBUFFER0: return r(gbuffer[0]); case RT_BUF_GBUFFER1: return r(gbuffer[1]); case RT_BUF_MOTION: return r(motion);
      case RT_BUF_DIRECT_RESV0: return r(directResv[0]); case RT_BUF_DIRECT_RESV1: return r(directResv[1]); case RT_BUF_DIRECT_RESV_TEMP: return r(directTemp);
      case RT_BUF_INDIRECT_RESV0: return r(indirectResv[0]); case RT_BUF_INDIRECT_RESV1: return r(indirectResv[1]); case RT_BUF_INDIRECT_RESV_TEMP: return r(indirectTemp);
      case RT_BUF_DENOISE_DIR_A: return r(denoiseTemp[0]); case RT_BUF_DENOISE_DIR_B: return r(denoiseTemp[1]);
      case RT_BUF_DENOISE_IND_A: return r(denoiseTemp[2]); case RT_BUF_DENOISE_IND_B: return r(denoiseTemp[3]);
      case RT_BUF_DIRECT_RESULT0: return r(directResult[0]); case RT_BUF_DIRECT_RESULT1: return r(directResult[1]);
      case RT_BUF_INDIRECT_RESULT0: return r(indirectResult[0]); case RT_BUF_INDIRECT_RESULT1: return r(indirectResult[1]);
    }
    bytes = 0; return nullptr;
  }
};
}  // namespace

extern "C" {
void* ref_create() { return new Ref(); }
void ref_destroy(void* p) { delete static_cast<Ref*>(p); }
int ref_upload_scene(void* p, const rt_scene_desc* d)
{
  Ref* r = static_cast<Ref*>(p);
  r->scene.upload(d);
  r->scene.build();
  // InstanceData (host_device.h:242-247, scene.cpp:179-195): one row per primitive mesh, buffer device addresses
  r->geoInfo.clear();
  for(const rt_prim_mesh& pm : r->scene.primMeshes) {
    InstanceData g{};
    g.vertexAddress = uint64_t(uintptr_t(r->scene.vertices.data() + pm.vertexOffset));
    g.indexAddress = uint64_t(uintptr_t(r->scene.indices.data() + pm.firstIndex));
    g.materialIndex = pm.materialIndex;
    r->geoInfo.push_back(g);
  }
  r->textures.clear();
  for(size_t i = 0; i < r->scene.textures.size(); i++) r->textures.push_back(sampler2D{&r->scene, int(i)});
  r->haveScene = true;
  return RT_OK;
}
int ref_set_sun_and_sky(void* p, const rt_sun_and_sky* ss) { static_cast<Ref*>(p)->scene.sunAndSky = *ss; return RT_OK; }
int ref_resize(void* p, int w, int h) { static_cast<Ref*>(p)->resize(w, h); return RT_OK; }
int ref_set_camera(void* p, const rt_scene_camera* cam) { static_cast<Ref*>(p)->cam = *cam; return RT_OK; }
int ref_render_frame(void* p, const rt_state* st, int frames)
{
  Ref* r = static_cast<Ref*>(p);
  if(!r->haveScene) return RT_ERR_NO_SCENE;
  if(st->size.x != r->W || st->size.y != r->H) return RT_ERR_NO_TARGET;
  r->renderFrame(*st, frames);
  return RT_OK;
}
int ref_run_stage(void* p, const rt_state* st, int frames, int stage, int level)
{
  Ref* r = static_cast<Ref*>(p);
  if(!r->haveScene) return RT_ERR_NO_SCENE;
  if(st->size.x != r->W || st->size.y != r->H) return RT_ERR_NO_TARGET;
  r->runStage(*st, frames, stage, level);
  return RT_OK;
}
size_t ref_buffer_bytes(void* p, int id) { size_t b = 0; static_cast<Ref*>(p)->buf(id, b); return b; }
int ref_readback(void* p, int id, void* dst, size_t bytes)
{
  size_t b; void* src = static_cast<Ref*>(p)->buf(id, b);
  if(!src || b != bytes) return RT_ERR_INVALID_ARG;
  memcpy(dst, src, b); return RT_OK;
}
int ref_upload_history(void* p, int id, const void* src, size_t bytes)
{
  size_t b; void* dst = static_cast<Ref*>(p)->buf(id, b);
  if(!dst || b != bytes) return RT_ERR_INVALID_ARG;
  memcpy(dst, src, b); return RT_OK;
}
}
