// orc_stages.cpp — CPU oracle restatement of the reference's 7 compute-shader entry points and of
// Renderer::run's dispatch schedule.  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load the library built from this directory.
//
//   direct_stage.comp   :150-288   -> Frame::directStage
//   direct_gen.comp     :77-149    -> Frame::directGen
//   direct_reuse.comp   :102-153   -> Frame::directReuse
//   indirect_stage.comp :129-309   -> Frame::indirectStage
//   denoise_direct.comp :19-71,139-174 / denoise_common.glsl -> Frame::denoiseDirect
//   denoise_indirect.comp :23-75,132-173                     -> Frame::denoiseIndirect
//   compose.comp        :23-43     -> Frame::compose
//   src/renderer.cpp    :154-206   -> Frame::renderFrame
//
// PARITY UNPINNED: the reference has no tests, golden vectors or fixtures and cannot be
// built here (Vulkan ray query + un-vendored nvpro_core).  The per-sample arithmetic the stages call (RNG, packing,
// reservoirs, BSDF, sky) is pinned by the known-answer vectors under tests/golden/ that were minted from the
// reference's own source files (oracle/kat/mint_kat.sh; tests/test_kat.py, tests/test_kat_float.py); the stage-level
// control flow in this file is a line-by-line restatement checked by property tests.
#include "orc_stages.h"
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace orc {

// ---- image helpers (Vulkan storage-image semantics: out-of-bounds loads return 0, stores are dropped) ---
static inline bool inImg(ivec2 c, int w, int h) { return c.x >= 0 && c.y >= 0 && c.x < w && c.y < h; }

uvec4 Frame::loadG(int which, ivec2 c) const
{
  if(!inImg(c, W, H)) return uvec4{0, 0, 0, 0};
  const uint32_t* p = &gbuffer[which][(size_t(c.y) * W + c.x) * 4];
  return uvec4{p[0], p[1], p[2], p[3]};
}
void Frame::storeG(int which, ivec2 c, uvec4 v)
{
  if(!inImg(c, W, H)) return;
  uint32_t* p = &gbuffer[which][(size_t(c.y) * W + c.x) * 4];
  p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
}
vec4 Frame::loadImg(const std::vector<float>& img, ivec2 c) const
{
  if(!inImg(c, W, H)) return V4(0, 0, 0, 0);
  const float* p = &img[(size_t(c.y) * W + c.x) * 4];
  return V4(p[0], p[1], p[2], p[3]);
}
void Frame::storeImg(std::vector<float>& img, ivec2 c, vec4 v)
{
  if(!inImg(c, W, H)) return;
  float* p = &img[(size_t(c.y) * W + c.x) * 4];
  p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
}
static inline int16_t sat16(int v) { return int16_t(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
void Frame::storeMotion(ivec2 c, ivec2 v)
{
  if(!inImg(c, W, H)) return;
  motion[(size_t(c.y) * W + c.x) * 2 + 0] = sat16(v.x);  // RG16_SINT (renderer.hpp:94)
  motion[(size_t(c.y) * W + c.x) * 2 + 1] = sat16(v.y);
}
ivec2 Frame::loadMotion(ivec2 c) const
{
  if(!inImg(c, W, H)) return ivec2{0, 0};
  return ivec2{motion[(size_t(c.y) * W + c.x) * 2], motion[(size_t(c.y) * W + c.x) * 2 + 1]};
}

void Frame::resize(int w, int h)
{
  W = w; H = h;
  // + 128 rows (64 half-res) of slack like the product's allocations, so the row-tiled multi-rank tests can all-gather in place
  const size_t n = size_t(w) * (h + 128), nh = size_t(w / 2) * (h / 2 + 64);
  for(int i = 0; i < 2; i++) {
    gbuffer[i].assign(n * 4, 0u);
    directResv[i].assign(n, rt_direct_reservoir{});
    indirectResv[i].assign(nh, rt_indirect_reservoir{});
    directResult[i].assign(n * 4, 0.0f);
    indirectResult[i].assign(n * 4, 0.0f);
  }
  directResvTemp.assign(n, rt_direct_reservoir{});
  indirectResvTemp.assign(nh, rt_indirect_reservoir{});
  motion.assign(n * 2, 0);
  for(int i = 0; i < 4; i++) denoiseTemp[i].assign(n * 4, 0.0f);
  for(int i = 0; i < 2; i++) lightId2[i].assign(n, 0xffffffffu);
  ldr.assign(n, 0u);
}

// ------------------------------------------------------------------------------------------------------------
// helpers shared by the direct stages
// ------------------------------------------------------------------------------------------------------------
static uvec4 encodeGeometryInfo(const State& state, float depth)  // direct_stage.comp:37-45
{
  uvec4 g;
  g.x = rt_f2u(depth);
  g.y = compress_unit_vec(state.normal);
  g.z = packUnorm4x8(V4(state.mat.metallic, state.mat.roughness, (state.mat.ior - 1.0f) / RT_MAX_IOR_MINUS_ONE, state.mat.transmission));
  g.w = packUnorm4x8(V4(state.mat.albedo, 1.0f)) & 0xFFFFFFu;
  g.w += hash8bit(state.matID);
  return g;
}
static void updateGeometryAlbedo(uvec4& g, vec3 albedo)  // direct_gen.comp:63-66
{
  uint32_t matId = g.w & 0xff000000u;
  g.w = (packUnorm4x8(V4(albedo, 1.0f)) & 0x00ffffffu) | matId;
}
static ivec2 createMotionIndex(const Shader& sh, vec3 wpos)  // direct_stage.comp:125-139
{
  vec4 proj = mul(Shader::M(sh.cam.lastProjView), V4(wpos, 1.0f));
  vec3 ndc = xyz(proj) / proj.w;
  vec2 mv = V2(ndc.x, ndc.y) * 0.5f + 0.5f;
  vec2 s = mv * V2(float(sh.rtx.size.x), float(sh.rtx.size.y));
  return ivec2{rt_ftoi(s.x), rt_ftoi(s.y)};
}
void Frame::loadLastGeometryInfo(int last, ivec2 c, vec3& normal, float& depth, uint32_t& matHash) const  // pathtrace.glsl:240-245
{
  uvec4 g = loadG(last, c);
  normal = decompress_unit_vec(g.y);
  depth = rt_u2f(g.x);
  matHash = g.w & 0xFF000000u;
}
// direct_stage.comp:47-84 (identical in direct_reuse.comp:52-89)
bool Frame::findTemporalNeighborDirect(const rt_state& st, int last, vec3 norm, float reprojDepth, uint32_t matId, ivec2 lastCoord,
                                       rt_direct_reservoir& resv, uint32_t& lid) const
{
  vec3 pnorm; float pdepth; uint32_t matHash;
  ivec2 size{st.size.x, st.size.y};
  if(!inBound(lastCoord, ivec2{2, 0}, size)) return false;
  if(lastCoord.y < histRow0 || lastCoord.y >= histRow1) histMiss = 1u;
  loadLastGeometryInfo(last, lastCoord, pnorm, pdepth, matHash);
  if(inBound(lastCoord, size)) {
    if(hash8bit(matId) == matHash) {
      if(dot(norm, pnorm) > 0.9f && reprojDepth < pdepth * 1.05f) {
        resv = directResv[last][size_t(lastCoord.y) * st.size.x + lastCoord.x];
        lid = lightId_last(last)[size_t(lastCoord.y) * st.size.x + lastCoord.x];
        return true;
      }
    }
  }
  return false;
}

// ------------------------------------------------------------------------------------------------------------
// direct_stage.comp  (the live direct pass)
// ------------------------------------------------------------------------------------------------------------
void Frame::directStage(const rt_state& st, int frames, int rowBegin, int rowEnd, int phase)
{
  const int cur = frames & 1, last = (frames + 1) & 1;
  // The reference does the spatial reuse inside the same dispatch, between barrier()s that only order one workgroup, reading
  // neighbours' tempDirectResv entries that other workgroups may or may not have written yet (a data race).  The race-free
  // reading restated here: every pixel caches its reservoir first (pass 1), then every pixel merges (pass 2).  `phase` 1 / 2
  // runs one pass only (row-tiled hosts exchange the neighbouring rows of the cache in between); the parked pixels live in the Frame.
  const bool spatial = st.ReSTIRState == RT_RESTIR_SPATIAL || st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL;
  std::vector<SpatialPending>& pend = spatialPend;
  if(spatial && pend.size() != size_t(st.size.x) * st.size.y) pend.assign(size_t(st.size.x) * st.size.y, SpatialPending());
  if(phase != 2)
    parallelRows(st.size.y, rowBegin, rowEnd, [&](int y) {
      for(int x = 0; x < st.size.x; x++) {
        Shader sh(*scene, st, cam);
        sh.imageCoords = ivec2{x, y};
        sh.seed = tea(uint32_t(st.size.x) * uint32_t(y) + uint32_t(x), st.time);  // :279
        Ray ray = sh.raySpawn(sh.imageCoords, ivec2{st.size.x, st.size.y});
        SpatialPending* P = spatial ? &pend[size_t(y) * st.size.x + x] : nullptr;
        if(P) P->active = false;
        vec3 radiance = ReSTIRDirect(sh, ray, cur, last, P);
        if(P && P->active) continue;
        vec3 pixelColor = sh.clampRadiance(radiance);
        storeImg(directResult[cur], sh.imageCoords, V4(pixelColor, 1.0f));  // :286
      }
    });
  if(!spatial || phase == 1) return;
  parallelRows(st.size.y, rowBegin, rowEnd, [&](int y) {
    for(int x = 0; x < st.size.x; x++) {
      const SpatialPending& P = pend[size_t(y) * st.size.x + x];
      if(!P.active) continue;
      Shader sh(*scene, st, cam);
      sh.imageCoords = ivec2{x, y};
      sh.seed = P.seed;
      vec3 radiance = finishSpatial(sh, P, cur);
      storeImg(directResult[cur], sh.imageCoords, V4(sh.clampRadiance(radiance), 1.0f));
    }
  });
}

// direct_stage.comp:86-121 + 229-262 for one pixel, after all pixels have run cacheTempReservoir
vec3 Frame::finishSpatial(Shader& sh, const SpatialPending& P, int cur)
{
  const rt_state& st = sh.rtx;
  const ivec2 size{st.size.x, st.size.y};
  rt_direct_reservoir resv = P.resv;
  float dummyPdf = 0;
  auto findSpatialNeighbor = [&](rt_direct_reservoir& out) -> bool {  // :86-107 (Radius is unused there; the geometry test looks at the pixel itself)
    float r0 = rnd(sh.seed), r1 = rnd(sh.seed);
    vec2 p = toConcentricDisk(V2(r0, r1));
    int px = rt_ftoi((float(sh.imageCoords.x) + p.x) + 0.5f);
    int py = rt_ftoi((float(sh.imageCoords.y) + p.y) + 0.5f);
    uvec4 g = loadG(cur, sh.imageCoords);  // loadThisGeometryInfo(imageCoords, ...)
    vec3 pnorm = decompress_unit_vec(g.y);
    float pdepth = rt_u2f(g.x);
    if(!inBound(ivec2{px, py}, size)) return false;
    if(dot(P.state.normal, pnorm) < 0.5f || rt_abs(P.hitT - pdepth) > P.hitT * 0.1f) return false;
    out = directResvTemp[size_t(py) * st.size.x + px];
    return true;
  };
  auto mergeSpatialNeighbors = [&](rt_direct_reservoir& out) -> bool {  // :109-121
    bool valid = false;
    memset(&out, 0, sizeof(out));  // `out` parameter + resvReset: sample undefined in GLSL, zero here (DESIGN.md deviation 3)
    for(int i = 0; i < 5; i++) {
      rt_direct_reservoir sp; memset(&sp, 0, sizeof(sp));
      if(findSpatialNeighbor(sp)) {
        if(!resvInvalid(sp)) { resvMerge(out, sp, rnd(sh.seed)); valid = true; }
      }
    }
    return valid;
  };
  rt_direct_reservoir spatial; memset(&spatial, 0, sizeof(spatial));
  for(int round = 0; round < 2; round++) {  // :236-252 (the reservoir cached between the rounds is the same one)
    rt_direct_reservoir agg;
    if(mergeSpatialNeighbors(agg)) {
      if(!resvInvalid(agg)) resvMerge(spatial, agg, rnd(sh.seed));
    }
  }
  if(!resvInvalid(spatial)) resvMerge(resv, spatial, rnd(sh.seed));
  vec3 direct = V3(0.0f);
  const rt_light_sample ls = resv.lightSample;
  if(!resvInvalid(resv)) {
    vec3 LiBsdf = toV(ls.Li) * sh.Eval(P.state, P.wo, P.state.ffnormal, toV(ls.wi), dummyPdf);
    direct = LiBsdf / resvToScalar(LiBsdf) * resv.weight / float(resv.num);
  }
  if(rt_isnan(direct.x) || rt_isnan(direct.y) || rt_isnan(direct.z)) direct = V3(0.0f);
  vec3 res = sh.clampRadiance(P.state.mat.emission + direct);
  return HDRToLDR(res);
}

vec3 Frame::ReSTIRDirect(Shader& sh, const Ray& r, int cur, int last, SpatialPending* pending)  // direct_stage.comp:150-270
{
  const rt_state& st = sh.rtx;
  const size_t index = size_t(sh.imageCoords.y) * st.size.x + sh.imageCoords.x;
  sh.ClosestHit(r);
  if(sh.hitT >= RT_INFINITY) {
    storeG(cur, sh.imageCoords, uvec4{rt_f2u(RT_INFINITY), 0, 0, RT_INVALID_MAT_ID});
    storeMotion(sh.imageCoords, ivec2{0, 0});
    return sh.EnvRadiance(r.direction);
  }
  State state = sh.GetState(r.direction);
  sh.GetMaterials(state, r);

  ivec2 motionIdx = createMotionIndex(sh, state.position);
  uvec4 gInfo = encodeGeometryInfo(state, sh.hitT);
  storeMotion(sh.imageCoords, motionIdx);
  storeG(cur, sh.imageCoords, gInfo);

  if(st.debugging_mode > RT_DBG_INDIRECT_STAGE) return sh.DebugInfo(state);
  if(state.isEmitter) return state.mat.emission;

  vec3 wo = -r.direction;
  vec3 direct = V3(0.0f);
  state.mat.albedo = V3(1.0f);
  float dummyPdf = 0;

  if(st.ReSTIRState == RT_RESTIR_NONE) {
    direct = sh.DirectLight(state, wo);
  } else {
    rt_direct_reservoir resv; memset(&resv, 0, sizeof(resv));
    uint32_t lid = 0xffffffffu;
    for(int i = 0; i < st.RISSampleNum; i++) {
      rt_light_sample ls;
      float p = sh.SampleDirectLightNoVisibility(state.position, ls);
      vec3 pHat = toV(ls.Li) * sh.Eval(state, wo, state.ffnormal, toV(ls.wi), dummyPdf) * rt_abs(dot(state.ffnormal, toV(ls.wi)));
      float weight = resvToScalar(pHat / p);
      if(Shader::IsPdfInvalid(p) || rt_isnan(weight)) weight = 0.0f;
      if(resvUpdate(resv, ls, weight, rnd(sh.seed))) lid = sh.lastLightId;
    }
    rt_light_sample ls = resv.lightSample;
    Ray shadowRay{OffsetRay(state.position, state.ffnormal), toV(ls.wi)};
    // a zero-weight reservoir cannot change (weight is only ever set to 0 here); skip its shadow ray
    if(resv.weight != 0.0f && sh.Occlusion(shadowRay, state, ls.dist)) resv.weight = 0.0f;

    if(st.ReSTIRState == RT_RESTIR_TEMPORAL || st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL) {
      float reprojDepth = length(toV(sh.cam.lastPosition) - state.position);
      rt_direct_reservoir temporal; uint32_t tlid = 0xffffffffu;
      if(findTemporalNeighborDirect(st, last, state.normal, reprojDepth, state.matID, motionIdx, temporal, tlid)) {
        if(!resvInvalid(temporal)) { if(resvMerge(resv, temporal, rnd(sh.seed))) lid = tlid; }
      }
    }
    rt_direct_reservoir tempResv = resv;
    resvCheckValidity(tempResv);
    resvClamp(tempResv, st.RISSampleNum * st.reservoirClamp);
    directResv[cur][index] = tempResv;  // saveNewReservoir
    lightId_cur(cur)[index] = lid;
    if(pending) {  // :229-234: resvCheckValidity(resv); cacheTempReservoir(resv); the rest happens in finishSpatial
      resvCheckValidity(resv);
      directResvTemp[index] = resv;
      pending->active = true; pending->state = state; pending->wo = wo; pending->hitT = sh.hitT; pending->seed = sh.seed; pending->resv = resv;
      return V3(0.0f);
    }
    ls = resv.lightSample;
    if(!resvInvalid(resv)) {
      vec3 LiBsdf = toV(ls.Li) * sh.Eval(state, wo, state.ffnormal, toV(ls.wi), dummyPdf);
      direct = LiBsdf / resvToScalar(LiBsdf) * resv.weight / float(resv.num);
    }
  }
  if(rt_isnan(direct.x) || rt_isnan(direct.y) || rt_isnan(direct.z)) direct = V3(0.0f);
  vec3 res = sh.clampRadiance(state.mat.emission + direct);
  return HDRToLDR(res);
}

// ------------------------------------------------------------------------------------------------------------
// direct_gen.comp + direct_reuse.comp (compiled by the reference, dispatch commented out: renderer.cpp:166-172)
// ------------------------------------------------------------------------------------------------------------
void Frame::directGen(const rt_state& st, int frames, int rowBegin, int rowEnd)
{
  const int cur = frames & 1;
  parallelRows(st.size.y, rowBegin, rowEnd, [&](int y) {
    for(int x = 0; x < st.size.x; x++) {
      Shader sh(*scene, st, cam);
      sh.imageCoords = ivec2{x, y};
      sh.seed = tea(uint32_t(st.size.x) * uint32_t(y) + uint32_t(x), st.time);  // direct_gen.comp:146
      Ray r = sh.raySpawn(sh.imageCoords, ivec2{st.size.x, st.size.y});
      const size_t index = size_t(y) * st.size.x + x;
      sh.ClosestHit(r);
      rt_direct_reservoir resv; memset(&resv, 0, sizeof(resv));
      uint32_t lid = 0xffffffffu;
      if(sh.hitT >= RT_INFINITY * 0.8f) {  // :86
        uvec4 g{rt_f2u(RT_INFINITY), 0, 0, RT_INVALID_MAT_ID};
        updateGeometryAlbedo(g, sh.EnvRadiance(r.direction));
        storeG(cur, sh.imageCoords, g);
        storeMotion(sh.imageCoords, ivec2{0, 0});
        directResv[cur][index] = resv;
        lightId_cur(cur)[index] = lid;
        continue;
      }
      State state = sh.GetState(r.direction);
      sh.GetMaterials(state, r);
      storeMotion(sh.imageCoords, createMotionIndex(sh, state.position));
      uvec4 g = encodeGeometryInfo(state, sh.hitT);
      if(st.debugging_mode > RT_DBG_INDIRECT_STAGE) updateGeometryAlbedo(g, sh.DebugInfo(state));
      else if(state.isEmitter) updateGeometryAlbedo(g, state.mat.emission);
      else {
        vec3 wo = -r.direction;
        state.mat.albedo = V3(1.0f);
        float dummyPdf = 0;
        for(int i = 0; i < st.RISSampleNum; i++) {
          rt_light_sample ls;
          float p = sh.SampleDirectLightNoVisibility(state.position, ls);
          vec3 pHat = toV(ls.Li) * sh.Eval(state, wo, state.ffnormal, toV(ls.wi), dummyPdf) * absDot(state.ffnormal, toV(ls.wi));
          float weight = resvToScalar(pHat / p);
          if(Shader::IsPdfInvalid(p) || rt_isnan(weight)) weight = 0.0f;
          if(resvUpdate(resv, ls, weight, rnd(sh.seed))) lid = sh.lastLightId;
        }
        rt_light_sample ls = resv.lightSample;
        Ray shadowRay{OffsetRay(state.position, state.ffnormal), toV(ls.wi)};
        if(resv.weight != 0.0f && sh.Occlusion(shadowRay, state, ls.dist)) resv.weight = 0.0f;
      }
      storeG(cur, sh.imageCoords, g);
      directResv[cur][index] = resv;
      lightId_cur(cur)[index] = lid;
    }
  });
}

// pathtrace.glsl:277-294
static bool getDirectStateFromGBuffer(uvec4 g, const Ray& ray, State& state, float& depth)
{
  depth = rt_u2f(g.x);
  if(depth >= RT_INFINITY * 0.8f) return false;
  state.position = ray.origin + ray.direction * depth;
  state.normal = decompress_unit_vec(g.y);
  state.ffnormal = dot(state.normal, ray.direction) <= 0.0f ? state.normal : -state.normal;
  state.mat.albedo = xyz(unpackUnorm4x8(g.w));
  vec4 matInfo = unpackUnorm4x8(g.z);
  state.mat.metallic = matInfo.x;
  state.mat.roughness = matInfo.y;
  state.mat.ior = matInfo.z * RT_MAX_IOR_MINUS_ONE + 1.f;
  state.mat.transmission = matInfo.w;
  state.matID = g.w >> 24;
  return true;
}

void Frame::directReuse(const rt_state& st, int frames, int rowBegin, int rowEnd)
{
  const int cur = frames & 1, last = (frames + 1) & 1;
  parallelRows(st.size.y, rowBegin, rowEnd, [&](int y) {
    for(int x = 0; x < st.size.x; x++) {
      Shader sh(*scene, st, cam);
      sh.imageCoords = ivec2{x, y};
      const int index = y * st.size.x + x;
      sh.seed = tea(uint32_t(index + st.size.x * st.size.y), st.time);  // direct_reuse.comp:109
      Ray ray = sh.raySpawn(sh.imageCoords, ivec2{st.size.x, st.size.y});
      State state; float depth;
      if(!getDirectStateFromGBuffer(loadG(cur, sh.imageCoords), ray, state, depth)) {
        storeImg(directResult[cur], sh.imageCoords, V4(0, 0, 0, 0));
        continue;
      }
      state.mat.albedo = V3(1.0f);
      vec3 direct = V3(0.0f);
      rt_direct_reservoir resv = directResv[cur][index];
      uint32_t lid = lightId_cur(cur)[index];
      rt_light_sample ls = resv.lightSample;  // NB: captured before the merge (direct_reuse.comp:124)
      vec3 wo = -ray.direction; (void)wo;
      ivec2 motionIdx = loadMotion(sh.imageCoords);
      if(st.ReSTIRState == RT_RESTIR_TEMPORAL || st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL) {
        float reprojDepth = length(toV(cam.lastPosition) - state.position);
        rt_direct_reservoir temporal; uint32_t tlid = 0xffffffffu;
        if(findTemporalNeighborDirect(st, last, state.normal, reprojDepth, state.matID, motionIdx, temporal, tlid)) {
          if(!resvInvalid(temporal)) { if(resvMerge(resv, temporal, rnd(sh.seed))) lid = tlid; }
        }
      }
      if(!resvInvalid(resv)) direct = toV(ls.Li);  // :143
      resvClamp(resv, st.RISSampleNum * st.reservoirClamp);
      resvCheckValidity(resv);
      if(rt_isnan(direct.x) || rt_isnan(direct.y) || rt_isnan(direct.z)) direct = V3(0.0f);
      directResv[cur][index] = resv;
      lightId_cur(cur)[index] = lid;
      storeImg(directResult[cur], sh.imageCoords, V4(HDRToLDR(sh.clampRadiance(direct)), 1.0f));
    }
  });
}

// ------------------------------------------------------------------------------------------------------------
// indirect_stage.comp
// ------------------------------------------------------------------------------------------------------------
static bool getIndirectStateFromGBuffer(uvec4 g, const Ray& ray, State& state, float& depth)  // pathtrace.glsl:296-313
{
  depth = rt_u2f(g.x);
  if(depth >= RT_INFINITY * 0.8f) return false;
  state.position = ray.origin + ray.direction * depth;
  state.normal = decompress_unit_vec(g.y);
  state.ffnormal = dot(state.normal, ray.direction) <= 0.0f ? state.normal : -state.normal;
  state.mat.albedo = xyz(unpackUnorm4x8(g.w));
  vec4 matInfo = unpackUnorm4x8(g.z);
  state.mat.metallic = matInfo.x;
  state.mat.roughness = matInfo.y;
  state.mat.ior = matInfo.z * RT_MAX_IOR_MINUS_ONE + 1.f;
  state.mat.transmission = matInfo.w;
  state.matID = g.w >> 24;
  return true;
}
static rt_gi_sample newGISample()  // indirect_stage.comp:110-115 (other fields: undefined in GLSL, 0 here)
{
  rt_gi_sample s; memset(&s, 0, sizeof(s));
  s.nv = rt_vec3{100.0f, 100.0f, 100.0f};
  return s;
}
static bool GISampleValid(const rt_gi_sample& s) { return s.nv.x < 1.1f && !hasNan(toV(s.L)); }  // :117-119
static float MIS(const rt_state& st, float f, float g) { return (st.MIS > 0) ? powerHeuristic(f, g) : 1.0f; }  // :57-59

// indirect_stage.comp:129-226
static void pathTraceIndirect(Shader& sh, State state, Ray ray, bool multiBounce, float& primSamplePdf, vec3& primWo, State& primState, rt_gi_sample& gi)
{
  const rt_state& st = sh.rtx;
  vec3 throughput = V3(multiBounce ? 4.0f : 1.0f);
  primWo = -ray.direction;
  primState = state;
  gi = newGISample();
  primSamplePdf = 0.0f;
  state.mat.albedo = V3(1.0f);
  static const char* dbgEnv = getenv("ORC_DEBUG_PIXEL");
  int dx = -1, dy = -1; if(dbgEnv) sscanf(dbgEnv, "%d,%d", &dx, &dy);
  const bool dbg = sh.imageCoords.x == dx && sh.imageCoords.y == dy;
  auto addL = [&](vec3 v) { gi.L = toR(toV(gi.L) + v); if(dbg) fprintf(stderr, "ORC   addL %08x %08x %08x -> L %g %g %g\n", rt_f2u(v.x), rt_f2u(v.y), rt_f2u(v.z), gi.L.x, gi.L.y, gi.L.z); };
  sh.dbgPrint = dbg;
  if(dbg) fprintf(stderr, "ORC pixel %d %d multiBounce %d seed %08x\n", dx, dy, int(multiBounce), sh.seed);

  for(int depth = 1; depth <= st.maxDepth; depth++) {
    vec3 wo = -ray.direction;
    if(depth > 1 && st.MIS > 0) {
      vec3 Li = V3(0.0f), wi = V3(0.0f);
      float lightPdf = sh.SampleDirectLight(state, Li, wi);
      if(dbg) fprintf(stderr, "ORC  depth %d NEE lightPdf %08x Li %g %g %g wi %08x %08x %08x lid %08x seed %08x\n", depth, rt_f2u(lightPdf), Li.x, Li.y, Li.z, rt_f2u(wi.x), rt_f2u(wi.y), rt_f2u(wi.z), sh.lastLightId, sh.seed);
      if(!Shader::IsPdfInvalid(lightPdf)) {
        float BSDFPdf = sh.Pdf(state, wo, state.ffnormal, wi);
        float weight = MIS(st, lightPdf, BSDFPdf);
        addL(Li * sh.BSDF(state, wo, state.ffnormal, wi) * absDot(state.ffnormal, wi) * throughput / lightPdf * weight);
      }
    }
    vec3 sampleWi = V3(0.0f);
    float samplePdf = 0.0f;
    vec3 sampleBSDF = sh.Sample(state, wo, state.ffnormal, sampleWi, samplePdf);
    if(Shader::IsPdfInvalid(samplePdf)) break;

    if(depth > 1) {
      if(!multiBounce) return;
      throughput *= sampleBSDF / samplePdf * absDot(state.ffnormal, sampleWi);
    } else {
      primSamplePdf = samplePdf;
      gi.xv = toR(state.position);
      gi.nv = toR(state.ffnormal);
    }
    ray.origin = OffsetRay(state.position, state.ffnormal);
    ray.direction = sampleWi;
    sh.ClosestHit(ray);
    if(dbg) fprintf(stderr, "ORC  depth %d bounce dir %08x %08x %08x pdf %08x hitT %08x seed %08x\n", depth, rt_f2u(sampleWi.x), rt_f2u(sampleWi.y), rt_f2u(sampleWi.z), rt_f2u(samplePdf), rt_f2u(sh.hitT), sh.seed);

    if(sh.hitT >= RT_INFINITY - 1e-4f) {
      if(depth > 1) {
        float lightPdf;
        vec3 Li = sh.EnvEval(sampleWi, lightPdf);
        float weight = MIS(st, samplePdf, lightPdf);
        addL(Li * throughput * weight);
      } else {
        gi.xs = toR(state.position + sampleWi * RT_INFINITY * 0.8f);
        gi.ns = toR(-sampleWi);
      }
      break;
    }
    state = sh.GetState(ray.direction);
    sh.GetMaterials(state, ray);

    if(state.isEmitter) {
      if(depth > 1) {
        float lightPdf;
        vec3 Li = sh.LightEval(state, sh.hitT, sampleWi, lightPdf);
        float weight = MIS(st, samplePdf, lightPdf);
        addL(Li * throughput * weight);
      } else {
        gi.xs = toR(state.position);
        gi.ns = toR(state.ffnormal);
      }
      break;
    }
    if(depth == 1) {
      gi.xs = toR(state.position);
      gi.ns = toR(state.ffnormal);
    }
    // Russian roulette: `#ifndef RR` block (:218-224) is compiled out because pathtrace.glsl:2 defines RR
  }
}

void Frame::indirectStage(const rt_state& st, int frames, int rowBegin, int rowEnd)
{
  const int cur = frames & 1, last = (frames + 1) & 1;
  const ivec2 indSize{st.size.x / 2, st.size.y / 2};
  parallelRows(indSize.y, rowBegin, rowEnd, [&](int y) {
    for(int x = 0; x < indSize.x; x++) {
      Shader sh(*scene, st, cam);
      sh.imageCoords = ivec2{x, y};
      sh.seed = tea(uint32_t(indSize.x) * uint32_t(y) + uint32_t(x), st.time);  // :280
      Ray ray = sh.raySpawn(sh.imageCoords, indSize);
      // TILED_MULTIBOUNCE (:283-288): local invocation 0 of each 8x8 workgroup draws the tile flag from ITS stream
      bool multiBounce;
      {
        int tx = x & ~7, ty = y & ~7;
        if(tx == x && ty == y) multiBounce = rnd(sh.seed) < 0.25f;
        else { uint32_t s0 = tea(uint32_t(indSize.x) * uint32_t(ty) + uint32_t(tx), st.time); multiBounce = rnd(s0) < 0.25f; }
      }
      State state; float depth;
      if(!getIndirectStateFromGBuffer(loadG(cur, ivec2{x * 2, y * 2}), ray, state, depth)) {
        storeImg(denoiseTemp[2], sh.imageCoords, V4(0, 0, 0, 0));
        continue;
      }
      state.position += state.ffnormal * 2e-2f;  // :299
      float primSamplePdf; vec3 primWo; State primState; rt_gi_sample gi;
      pathTraceIndirect(sh, state, ray, multiBounce, primSamplePdf, primWo, primState, gi);
      vec3 pixelColor = ReSTIRIndirect(sh, depth, primSamplePdf, primWo, primState, gi, cur, last);
      pixelColor = sh.clampRadiance(pixelColor);
      storeImg(denoiseTemp[2], sh.imageCoords, V4(pixelColor, 1.0f));
    }
  });
}

// indirect_stage.comp:228-268 (+ findTemporalNeighbor :74-108)
vec3 Frame::ReSTIRIndirect(Shader& sh, float dist, float primSamplePdf, vec3 primWo, State primState, rt_gi_sample gi, int cur, int last)
{
  (void)dist;
  const rt_state& st = sh.rtx;
  const ivec2 indSize{st.size.x / 2, st.size.y / 2};
  vec3 indirect = V3(0.0f);
  rt_indirect_reservoir resv; memset(&resv, 0, sizeof(resv));
  if(st.ReSTIRState == RT_RESTIR_TEMPORAL || st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL) {
    float reprojDepth = length(toV(sh.cam.lastPosition) - primState.position);
    ivec2 motionIdx = loadMotion(ivec2{sh.imageCoords.x * 2, sh.imageCoords.y * 2});
    vec3 pnorm; float pdepth; uint32_t matHash;
    if(motionIdx.x >= 0 && motionIdx.x < W && motionIdx.y >= 0 && motionIdx.y < H && (motionIdx.y < histRow0 || motionIdx.y >= histRow1)) histMissInd = 1u;
    loadLastGeometryInfo(last, motionIdx, pnorm, pdepth, matHash);
    ivec2 coord{motionIdx.x / 2, motionIdx.y / 2};
    if(inBound(coord, indSize)) {
      if(hash8bit(primState.matID) == matHash) {
        if(dot(primState.ffnormal, pnorm) > 0.5f && reprojDepth < pdepth * 1.1f) resv = indirectResv[last][size_t(coord.y) * indSize.x + coord.x];
      }
    }
  }
  float sampleWeight = 0.0f;
  if(GISampleValid(gi)) {
    gi.pHat = resvToScalar(toV(gi.L));  // pHatIndirect :61-66
    sampleWeight = gi.pHat / primSamplePdf;
    if(rt_isnan(sampleWeight) || sampleWeight < 0.0f) sampleWeight = 0.0f;
  }
  resvUpdate(resv, gi, sampleWeight, rnd(sh.seed));
  resvCheckValidity(resv);
  resvClamp(resv, st.reservoirClamp * 2);
  indirectResv[cur][size_t(sh.imageCoords.y) * indSize.x + sh.imageCoords.x] = resv;  // saveNewReservoir

  gi = resv.giSample;
  if(!resvInvalid(resv) && GISampleValid(gi)) {
    vec3 primWi = normalize(toV(gi.xs) - toV(gi.xv));
    primState.mat.albedo = V3(1.0f);
    float bigW = resv.weight / (resvToScalar(toV(resv.giSample.L)) * float(resv.num));  // bigWIndirect :68-70
    indirect = toV(gi.L) * sh.BSDF(primState, primWo, toV(gi.nv), primWi) * satDot(toV(gi.nv), primWi) * bigW;
  }
  vec3 res = sh.clampRadiance(indirect);
  return HDRToLDR(res);
}

// ------------------------------------------------------------------------------------------------------------
// denoise_common.glsl + denoise_direct.comp + denoise_indirect.comp
// ------------------------------------------------------------------------------------------------------------
static const float Gaussian5x5[5][5] = {{.0030f, .0133f, .0219f, .0133f, .0030f},
                                        {.0133f, .0596f, .0983f, .0596f, .0133f},
                                        {.0219f, .0983f, .1621f, .0983f, .0219f},
                                        {.0133f, .0596f, .0983f, .0596f, .0133f},
                                        {.0030f, .0133f, .0219f, .0133f, .0030f}};  // denoise_common.glsl:15-21

// denoise_common.glsl:27-40 — note: direction is NOT re-normalised after the view transform
static vec3 getCameraPosDenoise(const rt_scene_camera& cam, ivec2 coord, float dist, ivec2 imageSize)
{
  const vec2 pixelCenter = V2(float(coord.x), float(coord.y)) + 0.5f;
  const vec2 inUV = pixelCenter / V2(float(imageSize.x), float(imageSize.y));
  vec2 d = inUV * 2.0f - 1.0f;
  vec4 origin = mul(Shader::M(cam.viewInverse), V4(0, 0, 0, 1));
  vec4 target = mul(Shader::M(cam.projInverse), V4(d.x, d.y, 1, 1));
  vec4 direction = mul(Shader::M(cam.viewInverse), V4(normalize(xyz(target)), 0));
  return xyz(origin) + xyz(direction) * dist;
}
void Frame::loadThisGeometry(int cur, ivec2 coord, vec3& normal, vec3& pos, uint32_t& matHash, ivec2 imageSize) const  // :42-47
{
  uvec4 g = loadG(cur, coord);
  normal = decompress_unit_vec(g.y);
  pos = getCameraPosDenoise(cam, coord, rt_u2f(g.x), imageSize);
  matHash = g.w & 0xFF000000u;
}

// denoise_direct.comp:19-71 (indirect=false) and denoise_indirect.comp:23-75 (indirect=true)
vec3 Frame::waveletFilter(const rt_state& st, int cur, const std::vector<float>& inImage, ivec2 coord, vec3 norm, vec3 pos, uint32_t matHash,
                          float sigLumin, float sigNormal, float sigDepth, int level, bool indirect) const
{
  if(matHash == RT_INVALID_MAT_ID) return V3(0.0f);
  const int step = 1 << level;
  const ivec2 bound = indirect ? ivec2{st.size.x / 2, st.size.y / 2} : ivec2{st.size.x, st.size.y};
  vec3 sum = V3(0.0f);
  float sumWeight = 0.0f;
  vec3 color = xyz(loadImg(inImage, coord));
  for(int j = -2; j <= 2; j++) {
    for(int i = -2; i <= 2; i++) {
      ivec2 q{coord.x + i * step, coord.y + j * step};
      if(q.x >= bound.x || q.y >= bound.y || q.x < 0 || q.y < 0) continue;
      vec3 normQ, posQ; uint32_t matHashQ;
      if(indirect) loadThisGeometry(cur, ivec2{q.x * 2, q.y * 2}, normQ, posQ, matHashQ, bound);
      else loadThisGeometry(cur, q, normQ, posQ, matHashQ, bound);
      vec3 colorQ = xyz(loadImg(inImage, q));
      if(matHash != matHashQ || matHashQ == RT_INVALID_MAT_ID) continue;
      float var = sigLumin;
      float distColor = indirect ? dot(color - colorQ, color - colorQ) : rt_abs(luminance(color) - luminance(colorQ));
      float wColor = rt_exp(-distColor / var) + 1e-2f;
      float distNorm2 = dot(norm - normQ, norm - normQ);
      float wNorm = rt_min(1.0f, rt_exp(-distNorm2 / sigNormal));
      float distPos2 = dot(pos - posQ, pos - posQ);
      float wDepth = rt_exp(-distPos2 / sigDepth) + 1e-2f;
      float weight = wColor * wNorm * wDepth * Gaussian5x5[i + 2][j + 2];
      sum += colorQ * weight;
      sumWeight += weight;
    }
  }
  vec3 res = (sumWeight < 1e-5f) ? V3(0.0f) : sum / sumWeight;
  if(hasNan(res) || res.x < 0 || res.y < 0 || res.z < 0 || res.x > 1e8f || res.y > 1e8f || res.z > 1e8f) res = V3(0.0f);
  return res;
}

void Frame::denoiseDirect(const rt_state& st, int frames, int level, int rowBegin, int rowEnd)  // denoise_direct.comp:139-174
{
  const int cur = frames & 1;
  std::vector<float>* chain[5] = {&directResult[cur], &denoiseTemp[0], &denoiseTemp[1], &denoiseTemp[0], &directResult[cur]};
  if(level < 0 || level > 3) return;
  const std::vector<float>& src = *chain[level];
  std::vector<float>& dst = *chain[level + 1];
  const ivec2 size{st.size.x, st.size.y};
  parallelRows(size.y, rowBegin, rowEnd, [&](int y) {
    for(int x = 0; x < size.x; x++) {
      ivec2 coord{x, y};
      vec3 norm, pos; uint32_t matHash;
      loadThisGeometry(cur, coord, norm, pos, matHash, size);
      vec3 res = waveletFilter(st, cur, src, coord, norm, pos, matHash, st.sigLuminDirect, st.sigNormalDirect, st.sigDepthDirect, level, false);
      if(level == 3) res = LDRToHDR(res);
      storeImg(dst, coord, V4(res, 1.0f));
    }
  });
}

void Frame::denoiseIndirect(const rt_state& st, int frames, int level, int rowBegin, int rowEnd)  // denoise_indirect.comp:132-173
{
  const int cur = frames & 1;
  if(st.denoise == 0 || level < 0 || level > 4) return;
  // IndTempA -> B -> A -> thisIndirectResultImage (scratch!) -> A -> B
  std::vector<float>* chain[6] = {&denoiseTemp[2], &denoiseTemp[3], &denoiseTemp[2], &indirectResult[cur], &denoiseTemp[2], &denoiseTemp[3]};
  const std::vector<float>& src = *chain[level];
  std::vector<float>& dst = *chain[level + 1];
  const ivec2 ind{st.size.x / 2, st.size.y / 2};
  parallelRows(ind.y, rowBegin, rowEnd, [&](int y) {
    for(int x = 0; x < ind.x; x++) {
      ivec2 coord{x, y};
      vec3 norm, pos; uint32_t matHash;
      loadThisGeometry(cur, ivec2{x * 2, y * 2}, norm, pos, matHash, ind);
      vec3 res = waveletFilter(st, cur, src, coord, norm, pos, matHash, st.sigLuminIndirect, st.sigNormalIndirect, st.sigDepthIndirect, level, true);
      if(level == 4) res = LDRToHDR(res);
      storeImg(dst, coord, V4(res, 1.0f));
    }
  });
}

// ------------------------------------------------------------------------------------------------------------
// compose.comp:23-43
// ------------------------------------------------------------------------------------------------------------
void Frame::compose(const rt_state& st, int frames, int rowBegin, int rowEnd)
{
  const int cur = frames & 1;
  const std::vector<float>& indSrc = (st.denoise > 0) ? denoiseTemp[3] : denoiseTemp[2];
  parallelRows(st.size.y, rowBegin, rowEnd, [&](int y) {
    for(int x = 0; x < st.size.x; x++) {
      ivec2 coord{x, y};
      if(st.modulate == 0) {
        storeImg(indirectResult[cur], coord, loadImg(indSrc, ivec2{x / 2, y / 2}));
      } else {
        vec3 albedo = xyz(unpackUnorm4x8(loadG(cur, coord).w));
        vec3 direct = xyz(loadImg(directResult[cur], coord)) * albedo;
        vec3 indirect = xyz(loadImg(indSrc, ivec2{x / 2, y / 2})) * albedo;
        storeImg(directResult[cur], coord, V4(direct, 1.0f));
        storeImg(indirectResult[cur], coord, V4(indirect, 1.0f));
      }
    }
  });
}

// ------------------------------------------------------------------------------------------------------------
// Renderer::run, renderer.cpp:154-206
// ------------------------------------------------------------------------------------------------------------
void Frame::runStage(const rt_state& st, int frames, int stage, int level, int rowBegin, int rowEnd)
{
  switch(stage) {
    case RT_STAGE_DIRECT: directStage(st, frames, rowBegin, rowEnd, level); break;
    case RT_STAGE_INDIRECT: indirectStage(st, frames, rowBegin, rowEnd); break;
    case RT_STAGE_DENOISE_DIRECT: denoiseDirect(st, frames, level, rowBegin, rowEnd); break;
    case RT_STAGE_DENOISE_INDIRECT: denoiseIndirect(st, frames, level, rowBegin, rowEnd); break;
    case RT_STAGE_COMPOSE: compose(st, frames, rowBegin, rowEnd); break;
    case RT_STAGE_DIRECT_GEN: directGen(st, frames, rowBegin, rowEnd); break;
    case RT_STAGE_DIRECT_REUSE: directReuse(st, frames, rowBegin, rowEnd); break;
  }
}
void Frame::renderFrame(const rt_state& st, int frames)
{
  runStage(st, frames, RT_STAGE_DIRECT, 0, 0, 0);
  runStage(st, frames, RT_STAGE_INDIRECT, 0, 0, 0);
  if(st.denoise > 0) {
    for(int i = 0; i < 4; i++) runStage(st, frames, RT_STAGE_DENOISE_DIRECT, i, 0, 0);
    for(int i = 0; i < 5; i++) runStage(st, frames, RT_STAGE_DENOISE_INDIRECT, i, 0, 0);
  }
  runStage(st, frames, RT_STAGE_COMPOSE, 0, 0, 0);
}

}  // namespace orc
