// host_capi.cpp — plain-C handles over Scene / HdrSampling for the Python tests, bench.py and smoke().
#include "scene.hpp"
#include "alias_table.hpp"
#include <new>

using namespace rth;

namespace rth { bool decodeJpeg(const uint8_t* d, size_t n, TextureImage& img); }  // jpeg_decoder.cpp
namespace rth { bool decodePngImage(const uint8_t* d, size_t n, TextureImage& img); }   // gltf_loader.cpp
extern "C" {

void* rth_scene_create() { return new(std::nothrow) Scene(); }
void rth_scene_destroy(void* s) { delete static_cast<Scene*>(s); }
// no exception crosses the C boundary (the loaders allocate from sizes they read in files): a malformed or oversized input is an error code
#define RTH_GUARD(expr) do { try { return (expr); } catch(...) { return -1; } } while(0)
int rth_scene_load(void* s, const char* path) { RTH_GUARD(static_cast<Scene*>(s)->load(path) ? 0 : -1); }
// image decoders on a memory block (tests): returns 0 and fills w/h + BGRA8 pixels (caller passes a buffer of cap bytes)
int rth_decode_jpeg(const uint8_t* data, size_t n, int* w, int* h, uint8_t* out, size_t cap)
{
  TextureImage t;
  try { if(!rth::decodeJpeg(data, n, t)) return -1; } catch(...) { return -1; }
  *w = t.width; *h = t.height;
  if(t.bgra.size() > cap) return -2;
  memcpy(out, t.bgra.data(), t.bgra.size());
  return 0;
}
int rth_write_png(const char* path, const uint8_t* rgba, int w, int h, int keepAlpha) { return rth::writePng(path, rgba, w, h, keepAlpha != 0) ? 0 : -1; }
int rth_decode_png(const uint8_t* data, size_t n, int* w, int* h, uint8_t* out, size_t cap)
{
  TextureImage t;
  try { if(!rth::decodePngImage(data, n, t)) return -1; } catch(...) { return -1; }
  *w = t.width; *h = t.height;
  if(t.bgra.size() > cap) return -2;
  memcpy(out, t.bgra.data(), t.bgra.size());
  return 0;
}
int rth_scene_save_gltf(void* s, const char* path)
{
  std::string err;
  if(saveGltfFile(path, static_cast<Scene*>(s)->getScene(), err)) return 0;
  fprintf(stderr, "rth_scene_save_gltf: %s\n", err.c_str());
  return -1;
}
int rth_scene_make_procedural(void* s, int kind, float scale, uint32_t seed)
{
  const char* names[] = {"cornell", "helmet-class", "sponza-class", "bistro-exterior-class", "bistro-interior-class", "bistro-exterior-class (real footprint)", "sponza-class (1k textures)"};
  if(kind < 0 || kind > 6) return -1;
  return static_cast<Scene*>(s)->loadFromGltfScene(makeProceduralScene(ProcScene(kind), scale, seed), names[kind]) ? 0 : -1;
}
void rth_scene_set_camera(void* s, const float* eye, const float* center, const float* up, float fovDeg)
{
  static_cast<Scene*>(s)->setCamera(V3{eye[0], eye[1], eye[2]}, V3{center[0], center[1], center[2]}, V3{up[0], up[1], up[2]}, fovDeg);
}
void rth_scene_get_camera_pose(void* s, float* out10)
{
  Scene* sc = static_cast<Scene*>(s);
  out10[0] = sc->m_eye.x; out10[1] = sc->m_eye.y; out10[2] = sc->m_eye.z;
  out10[3] = sc->m_center.x; out10[4] = sc->m_center.y; out10[5] = sc->m_center.z;
  out10[6] = sc->m_up.x; out10[7] = sc->m_up.y; out10[8] = sc->m_up.z; out10[9] = sc->m_fov;
}
void rth_scene_update_camera(void* s, int w, int h) { static_cast<Scene*>(s)->updateCamera(w, h); }
void rth_scene_get_camera(void* s, rt_scene_camera* out) { *out = static_cast<Scene*>(s)->getCamera(); }
void rth_scene_light_weights(void* s, float* punc, float* trig) { *punc = static_cast<Scene*>(s)->m_puncLightWeight; *trig = static_cast<Scene*>(s)->m_trigLightWeight; }
// {triangles, instancedTriangles, vertices, primMeshes, nodes, materials, textures, puncLights, trigLights}
void rth_scene_stats(void* s, uint64_t* out9)
{
  const SceneStats& st = static_cast<Scene*>(s)->getStat();
  out9[0] = st.triangles; out9[1] = st.instancedTriangles; out9[2] = st.vertices; out9[3] = st.primMeshes; out9[4] = st.nodes;
  out9[5] = st.materials; out9[6] = st.textures; out9[7] = st.puncLights; out9[8] = st.trigLights;
}
void rth_scene_desc(void* s, void* env, rt_scene_desc* out) { *out = static_cast<Scene*>(s)->getDesc(static_cast<HdrSampling*>(env)); }

void* rth_env_create() { return new(std::nothrow) HdrSampling(); }
void rth_env_destroy(void* e) { delete static_cast<HdrSampling*>(e); }
int rth_env_load(void* e, const char* path) { RTH_GUARD(static_cast<HdrSampling*>(e)->loadEnvironment(path) ? 0 : -1); }
void rth_env_set(void* e, const float* rgba, int w, int h) { static_cast<HdrSampling*>(e)->setEnvironment(rgba, w, h); }
void rth_env_make_sky(void* e, int w, int h, float sunPeak, uint32_t seed) { static_cast<HdrSampling*>(e)->makeSyntheticSky(w, h, sunPeak, seed); }
float rth_env_integral(void* e) { return static_cast<HdrSampling*>(e)->getIntegral(); }
float rth_env_average(void* e) { return static_cast<HdrSampling*>(e)->getAverage(); }
int rth_env_width(void* e) { return static_cast<HdrSampling*>(e)->width(); }
int rth_env_height(void* e) { return static_cast<HdrSampling*>(e)->height(); }
void rth_env_get_accel(void* e, rt_impt_samp* out) { const auto& a = static_cast<HdrSampling*>(e)->accel(); memcpy(out, a.data(), a.size() * sizeof(rt_impt_samp)); }

// RtxState defaults of the reference application (sample_example.hpp:154-184); size/time/integrals left to the caller
void rth_default_state(rt_state* st)
{
  *st = rt_state{};
  st->frame = 0; st->maxDepth = 4; st->modulate = 1; st->fireflyClampThreshold = 1.f;
  st->hdrMultiplier = 1.f; st->debugging_mode = 0; st->environmentProb = 0.25f; st->time = 0;
  st->ReSTIRState = RT_RESTIR_TEMPORAL; st->RISSampleNum = 4; st->reservoirClamp = 80; st->accumulate = 0;
  st->envMapLuminIntegInv = 0.f; st->lightLuminIntegInv = 0.f; st->MIS = 1;
  st->sigLuminDirect = 0.4f; st->sigNormalDirect = 0.1f; st->sigDepthDirect = 0.02f; st->denoise = 1;
  st->sigLuminIndirect = 4.f; st->sigNormalIndirect = 0.4f; st->sigDepthIndirect = 1.f; st->denoiseLevel = 0;
}

void rth_alias_table(int n, const float* w, float* prob, int* failId)
{
  std::vector<AliasBucket> t = buildAliasTable(std::vector<float>(w, w + n));
  for(int i = 0; i < n; i++) { prob[i] = t[size_t(i)].prob; failId[i] = t[size_t(i)].failId; }
}

}  // extern "C"
