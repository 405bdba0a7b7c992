// direct_phases.h — the direct stage cut at its shadow ray (direct_stage.comp:150-270, direct_gen.comp:77-149):
//   directShadePixel    everything between the primary hit and the shadow ray: G-buffer, motion vector, M-candidate RIS (or the
//                       single DirectLight sample); parks what the second half needs in the per-pixel scratch records
//   directResolvePixel  visibility of the winner, temporal reuse, reservoir store, shading
// Used by the wavefront organisation (one kernel per half, rays in HBM queues) and by the fused K-tiles-per-wave direct
// kernel (both halves in one kernel, rays in the wave's LDS pool).  Same arithmetic and RNG draw order as k_direct_stage.
#pragma once
#include "stage_common.h"

namespace rt {

enum : uint32_t { ST_DONE = 0u, ST_RIS = 1u, ST_NONE = 2u };

// c.seed must hold the pixel's stream (tea(...) of direct_stage.comp:279), c.hit the primary hit.  Returns whether a shadow
// ray has to be traced; its origin/tmax and direction/seed come back in shadowO / shadowD.
RT_DEV bool directShadePixel(Ctx& c, const DevFrame& F, const rt_state& st, i2 px, int genOnly, float4& shadowO, float4& shadowD)
{
  c.imageCoords = px;
  const size_t index = size_t(px.y) * st.size.x + px.x;
  bool wantShadow = false;
  const Ray r = c.raySpawn(px, i2{st.size.x, st.size.y});
  uint32_t status = ST_DONE;
  f3 radiance = mk3(0.0f);
  const bool miss = genOnly ? (c.hit.t >= RT_INFINITY * 0.8f) : (c.hit.t >= RT_INFINITY);  // direct_gen.comp:86 vs direct_stage.comp:155
  if(miss) {
    uint4 g = make_uint4(rt_f2u(RT_INFINITY), 0u, 0u, RT_INVALID_MAT_ID);
    radiance = c.EnvRadiance(r.direction);
    if(genOnly) { updateGeometryAlbedo(g, radiance); F.thisDirectResv[index] = zeroDirectResv(); F.thisLightId[index] = 0xffffffffu; }
    F.thisG[index] = g;
    storeMotion(F, px, i2{0, 0});
  } else {
    State state = c.GetState(r.direction);
    c.GetMaterials(state, r);
    const i2 motionIdx = createMotionIndex(c, state.position);
    uint4 gInfo = encodeGeometryInfo(state, c.hit.t);
    storeMotion(F, px, motionIdx);
    bool ris = false;
    if(st.debugging_mode > RT_DBG_INDIRECT_STAGE) { radiance = c.DebugInfo(state); if(genOnly) updateGeometryAlbedo(gInfo, radiance); }
    else if(state.isEmitter) { radiance = state.mat.emission; if(genOnly) updateGeometryAlbedo(gInfo, radiance); }
    else ris = true;
    F.thisG[index] = gInfo;
    if(genOnly && !ris) { F.thisDirectResv[index] = zeroDirectResv(); F.thisLightId[index] = 0xffffffffu; }  // direct_gen.comp:136-137
    if(ris) {
      const f3 wo = -r.direction;
      state.mat.albedo = mk3(1.0f);
      rt_direct_reservoir resv = zeroDirectResv();
      uint32_t lid = 0xffffffffu;
      rt_light_sample ls;
      if(!genOnly && st.ReSTIRState == RT_RESTIR_NONE) {
        // DirectLight (pathtrace.glsl:205-220): the sample's pdf travels in resv.weight, the sample in resv.lightSample
        const float pdf = c.SampleDirectLightNoVisibility(state.position, ls);
        resv.lightSample = ls; resv.weight = pdf;
        status = ST_NONE;
        wantShadow = !Ctx::IsPdfInvalid(pdf);
      } else {
        for(int i = 0; i < st.RISSampleNum; i++) {  // direct_stage.comp:189-200
          const float p = c.SampleDirectLightNoVisibility(state.position, ls);
          const f3 pHat = mk3(ls.Li) * metallicWorkflowBSDF(state.mat, state.ffnormal, wo, mk3(ls.wi)) * rt_abs(dot(state.ffnormal, mk3(ls.wi)));
          float weight = resvToScalar(pHat / p);
          if(Ctx::IsPdfInvalid(p) || rt_isnan(weight)) weight = 0.0f;
          if(resvUpdate(resv, ls, weight, rnd(c.seed))) lid = c.lastLightId;
        }
        status = ST_RIS;
        wantShadow = resv.weight != 0.0f;  // a zero-weight reservoir cannot change: its shadow ray is skipped
      }
      ls = resv.lightSample;
      const f3 org = OffsetRay(state.position, state.ffnormal);
      const float tmax = ((ls.dist - rt_abs(org.x - state.position.x)) - rt_abs(org.y - state.position.y)) - rt_abs(org.z - state.position.z);  // Occlusion, pathtrace.glsl:18-22
      shadowO = make_float4(org.x, org.y, org.z, tmax);
      shadowD = make_float4(ls.wi.x, ls.wi.y, ls.wi.z, rt_u2f(c.seed));
      F.cand[index] = resv;
      F.candLid[index] = lid;
      SurfRec sr;
      sr.position = toR(state.position); sr.normal = toR(state.normal); sr.ffnormal = toR(state.ffnormal); sr.emission = toR(state.mat.emission);
      sr.roughness = state.mat.roughness; sr.metallic = state.mat.metallic; sr.matID = state.matID; sr.seed = c.seed;
      F.surf[index] = sr;
    }
  }
  F.status[index] = status;
  if(status == ST_DONE && !genOnly) storeImg(F.thisDirectResult, F, px, mk4(c.clampRadiance(radiance), 1.0f));  // direct_stage.comp:285-286
  return wantShadow;
}

// `occluded`: result of the shadow ray parked by directShadePixel (false when none was needed).
RT_DEV void directResolvePixel(Ctx& c, const DevFrame& F, const rt_state& st, const rt_scene_camera& cam, i2 px, bool occluded, int genOnly)
{
  const size_t index = size_t(px.y) * st.size.x + px.x;
  const uint32_t status = F.status[index];
  if(status == ST_DONE) return;
  c.imageCoords = px;
  const SurfRec sr = F.surf[index];
  c.seed = sr.seed;
  rt_direct_reservoir resv = F.cand[index];
  uint32_t lid = F.candLid[index];
  if(genOnly) {  // direct_gen.comp:127-137
    if(occluded) resv.weight = 0.0f;
    F.thisDirectResv[index] = resv;
    F.thisLightId[index] = lid;
    return;
  }
  const Ray r = c.raySpawn(px, i2{st.size.x, st.size.y});
  const f3 wo = -r.direction;
  Material mat;
  mat.albedo = mk3(1.0f); mat.emission = mk3(sr.emission); mat.metallic = sr.metallic; mat.roughness = sr.roughness; mat.ior = 0.f; mat.transmission = 0.f;
  const f3 position = mk3(sr.position), normal = mk3(sr.normal), ffnormal = mk3(sr.ffnormal);
  f3 direct = mk3(0.0f);
  if(status == ST_NONE) {
    const float pdf = resv.weight;
    const rt_light_sample ls = resv.lightSample;
    if(!Ctx::IsPdfInvalid(pdf) && !occluded)
      direct = mk3(ls.Li) * metallicWorkflowBSDF(mat, ffnormal, wo, mk3(ls.wi)) * rt_max(dot(ffnormal, mk3(ls.wi)), 0.0f) / pdf;
  } else {
    if(occluded) resv.weight = 0.0f;
    if(st.ReSTIRState == RT_RESTIR_TEMPORAL || st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL) {
      const float reprojDepth = length(mk3(cam.lastPosition) - position);
      const i2 motionIdx = loadMotion(F, px);  // RG16_SINT-saturated; equivalent to the unsaturated index for sizes <= 32767
      rt_direct_reservoir temporal; uint32_t tlid = 0xffffffffu;
      if(findTemporalNeighborDirect(F, st, normal, reprojDepth, sr.matID, motionIdx, temporal, tlid)) {
        if(!resvInvalidW(temporal.weight)) { if(resvMerge(resv, temporal, rnd(c.seed))) lid = tlid; }
      }
    }
    rt_direct_reservoir tempResv = resv;
    if(resvInvalidW(tempResv.weight)) { tempResv.num = 0; tempResv.weight = 0.f; }
    resvClamp(tempResv, st.RISSampleNum * st.reservoirClamp);
    F.thisDirectResv[index] = tempResv;
    F.thisLightId[index] = lid;
    const rt_light_sample ls = resv.lightSample;
    if(!resvInvalidW(resv.weight)) {
      const f3 LiBsdf = mk3(ls.Li) * metallicWorkflowBSDF(mat, ffnormal, wo, mk3(ls.wi));
      direct = LiBsdf / resvToScalar(LiBsdf) * resv.weight / float(resv.num);
    }
  }
  if(rt_isnan(direct.x) || rt_isnan(direct.y) || rt_isnan(direct.z)) direct = mk3(0.0f);
  const f3 radiance = HDRToLDR(c.clampRadiance(mat.emission + direct));
  storeImg(F.thisDirectResult, F, px, mk4(c.clampRadiance(radiance), 1.0f));
}

}  // namespace rt
