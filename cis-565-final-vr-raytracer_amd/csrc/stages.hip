// stages.hip — the per-frame kernels of the ReSTIR DI+GI path on gfx950.
//
//   k_direct_stage     direct_stage.comp:150-288  (live)     primary ray + G-buffer + M-candidate RIS + shadow ray + temporal reuse + shade
//   k_direct_gen       direct_gen.comp:77-149     (compiled by the reference, dispatch disabled: renderer.cpp:166-168)
//   k_direct_reuse     direct_reuse.comp:102-153  (idem, renderer.cpp:170-171)
//   k_indirect_stage   indirect_stage.comp:129-309, half resolution, tile-level multi-bounce
//   k_denoise_lds<IND,FAST> denoise_direct.comp / denoise_indirect.comp: one edge-avoiding A-Trous level
//   k_compose          compose.comp:23-43
//
// Launch shape: one wave64 per 8x8 pixel tile (= the reference's 8x8 workgroups, host_device.h:31-38), 1-D grid with
// an XCD-aware tile order: consecutive workgroup ids land on different XCDs (observed b % 8), so XCD x gets the x-th
// contiguous band of tile rows and its private 4 MiB L2 sees one compact screen region + the BVH subtrees under it.
// This file is compiled six times: as is, and through stages_{sky,cnt,sky_cnt,lat,sky_lat}.hip with RT_SKY / RT_COUNT / RT_LAT set.
//   RT_SKY = 1    procedural sun & sky code paths compiled in.  Keeping sun_and_sky() out of the default kernels saves 13 VGPRs
//                 in k_direct_stage — the procedural sky is the rarely used mode (default in_use = 0, sample_example.hpp:202).
//   RT_COUNT = 1  traversal / shading counters (rt_set_counting) are flushed at the end of the traced kernels.  Without the
//                 flush the compiler removes the per-round counter increments from the traversal loops: -1.7 % frame time.
//   RT_LAT = 1    the traced kernels for SMALL launches (row bands of a multi-GPU frame, small images): the latency-mode traversal round of
//                 traverse.h and two waves per SIMD worth of registers.  Only the three traced kernels are compiled; every other stage forwards to
//                 the throughput build.  rt_api.cpp picks per launch (launch size; rt_set_traversal forces).  Same bits.
#ifndef RT_SKY
#define RT_SKY 0
#endif
#ifndef RT_COUNT
#define RT_COUNT 0
#endif
#ifndef RT_LAT
#define RT_LAT 0
#endif
#if RT_LAT && RT_COUNT
#error "the counting build exists for the throughput kernels only"
#endif
#if RT_LAT && RT_SKY
#define RT_VARIANT sky_lat
#define RT_FORWARD sky
#elif RT_LAT
#define RT_VARIANT base_lat
#define RT_FORWARD base
#elif RT_SKY && RT_COUNT
#define RT_VARIANT sky_cnt
#elif RT_SKY
#define RT_VARIANT sky
#elif RT_COUNT
#define RT_VARIANT base_cnt
#else
#define RT_VARIANT base
#endif
#include "stage_common.h"
#include <algorithm>

namespace rt {
namespace RT_VARIANT {

// ------------------------------------------------------------------------------------------------------------
// direct_stage.comp
// ------------------------------------------------------------------------------------------------------------
// launch bounds of the two traced kernels, re-measured in round 2 under the final schedule (frames in flight, ms/frame, same box):
// (direct, indirect) waves per SIMD = (5,5) 3.155, (4,5) 3.121, (4,4) 3.29, (5,4) 3.59, (3,5) 3.16, (4,6) 3.15  =>  (4, 5)
// and again in round 4 on the build whose traversal stacks and filters had given LDS back (profiles/r04_launch_bounds_ab.txt, two boxes):
// (4,5) 2.987 / 2.993, (5,5) 2.901 / 2.893, (6,5) 2.891, (5,4) 2.923, (4,4) 2.995, (5,6) 2.904, (6,6) 2.920  =>  (5, 5): 96 VGPRs and 31 spilled
// values (60 B of scratch per lane, none of it inside a traversal loop) buy a fifth wave per SIMD.
#ifndef RT_DIRECT_LB
#define RT_DIRECT_LB 5
#endif
// Register cap of the traced kernels below what the launch bounds give (96 at 5 waves per SIMD).  Why one would: 5 x 96 = 480 of a SIMD's 512 registers — the 32 left
// hold no wave of the filter chain (k_denoise_lds: 50-57), which therefore only gets a slot when a traversal wave retires and stretches 6x with frames in flight;
// 5 x 88 + 56 = 496 would fit one.  Measured in round 6: profiles/r06_vgpr_room_ab.txt.
#ifdef RT_DIRECT_VGPR
#define RT_DIRECT_VGPR_ATTR __attribute__((amdgpu_num_vgpr((RT_DIRECT_VGPR) / 2)))   // (gfx90a and later: the backend doubles the attribute — unified VGPR / AGPR file)
#else
#define RT_DIRECT_VGPR_ATTR
#endif
#ifdef RT_INDIRECT_VGPR
#define RT_INDIRECT_VGPR_ATTR __attribute__((amdgpu_num_vgpr((RT_INDIRECT_VGPR) / 2)))
#else
#define RT_INDIRECT_VGPR_ATTR
#endif
// ReSTIRDirect (direct_stage.comp:150-270) cut at its one shadow ray, so that the ray can be traced by whatever the build uses (the lane's own
// traversal loop in the throughput build, the workgroup's ray pool in the latency build) while both builds share every line of shading:
//   directPre   primary hit -> G-buffer, motion vector, M-candidate RIS (or the single DirectLight sample); returns whether a shadow ray is needed
//   directPost  visibility of the winner, temporal reuse, reservoir store, shading, result store
struct DirectCont {
  State state; f3 wo, radiance; i2 motionIdx;
  rt_direct_reservoir resv; uint32_t lid;
  float pdf;     // kind 2: pdf of the single light sample (kept in resv.lightSample)
  int kind;      // 0: radiance is final (miss, emitter, debug view), 1: RIS reservoir, 2: single sample (ReSTIRState none)
};
RT_DEV float occlusionDist(const Ray& ray, f3 statePos, float dist)   // Occlusion, pathtrace.glsl:18-22
{
  return ((dist - rt_abs(ray.origin.x - statePos.x)) - rt_abs(ray.origin.y - statePos.y)) - rt_abs(ray.origin.z - statePos.z);
}
RT_DEV bool directPre(Ctx& c, const DevFrame& F, const rt_state& st, i2 px, const Ray& r, DirectCont& K, Ray& shadowRay, float& shadowDist)
{
  const size_t index = size_t(px.y) * st.size.x + px.x;
  K.kind = 0; K.radiance = mk3(0.0f);
  if(c.hit.t >= RT_INFINITY) {  // :155-159
    F.thisG[index] = make_uint4(rt_f2u(RT_INFINITY), 0u, 0u, RT_INVALID_MAT_ID);
    storeMotion(F, px, i2{0, 0});
    K.radiance = c.EnvRadiance(r.direction);
    return false;
  }
  State& state = K.state;
  state = c.GetState(r.direction);
  c.GetMaterials(state, r);
  K.motionIdx = createMotionIndex(c, state.position);
  const uint4 gInfo = encodeGeometryInfo(state, c.hit.t);
  storeMotion(F, px, K.motionIdx);
  F.thisG[index] = gInfo;
  if(st.debugging_mode > RT_DBG_INDIRECT_STAGE) { K.radiance = c.DebugInfo(state); return false; }
  if(state.isEmitter) { K.radiance = state.mat.emission; return false; }
  K.wo = -r.direction;
  state.mat.albedo = mk3(1.0f);
  K.resv = zeroDirectResv();
  K.lid = 0xffffffffu;
  if(st.ReSTIRState == RT_RESTIR_NONE) {  // DirectLight, pathtrace.glsl:205-220
    K.kind = 2;
    rt_light_sample ls;
    K.pdf = c.SampleDirectLightNoVisibility(state.position, ls);
    K.resv.lightSample = ls;
    if(Ctx::IsPdfInvalid(K.pdf)) return false;
    shadowRay = Ray{OffsetRay(state.position, state.ffnormal), mk3(ls.wi)};
    shadowDist = occlusionDist(shadowRay, state.position, ls.dist);
    return true;
  }
  K.kind = 1;
  return risCandidatesNoVisibility(c, state, K.wo, K.resv, K.lid, shadowRay, shadowDist);
}
RT_DEV void directPost(Ctx& c, const DevFrame& F, const rt_state& st, const rt_scene_camera& cam, i2 px, DirectCont& K, bool occluded)
{
  const size_t index = size_t(px.y) * st.size.x + px.x;
  const bool spatial = st.ReSTIRState == RT_RESTIR_SPATIAL || st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL;
  bool deferred = false;
  f3 radiance = K.radiance;
  if(K.kind != 0) {
    State& state = K.state;
    const f3 wo = K.wo;
    f3 direct = mk3(0.0f);
    if(K.kind == 2) {
      const rt_light_sample ls = K.resv.lightSample;
      if(!Ctx::IsPdfInvalid(K.pdf) && !occluded)
        direct = mk3(ls.Li) * metallicWorkflowBSDF(state.mat, state.ffnormal, wo, mk3(ls.wi)) * rt_max(dot(state.ffnormal, mk3(ls.wi)), 0.0f) / K.pdf;
    } else {
      rt_direct_reservoir& resv = K.resv;
      uint32_t lid = K.lid;
      if(occluded) resv.weight = 0.0f;
      if(st.ReSTIRState == RT_RESTIR_TEMPORAL || st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL) {
        const float reprojDepth = length(mk3(cam.lastPosition) - state.position);
        rt_direct_reservoir temporal; uint32_t tlid = 0xffffffffu;
        if(findTemporalNeighborDirect(F, st, state.normal, reprojDepth, state.matID, K.motionIdx, temporal, tlid)) {
          if(!resvInvalidW(temporal.weight)) { if(resvMerge(resv, temporal, rnd(c.seed))) lid = tlid; }
        }
      }
      rt_direct_reservoir tempResv = resv;
      if(resvInvalidW(tempResv.weight)) { tempResv.num = 0; tempResv.weight = 0.f; }
      resvClamp(tempResv, st.RISSampleNum * st.reservoirClamp);
      F.thisDirectResv[index] = tempResv;  // saveNewReservoir
      F.thisLightId[index] = lid;
      if(spatial) {
        // Spatial / spatiotemporal reuse (:224-255) reads the reservoirs its neighbours cache here.  The reference orders
        // that with workgroup barriers only (neighbours in other workgroups race); this build finishes the pixel in a
        // second kernel (k_direct_spatial) once every pixel of the launch has cached its reservoir.
        if(resvInvalidW(resv.weight)) { resv.num = 0; resv.weight = 0.f; }  // resvCheckValidity(resv) :231
        F.tempDirectResv[index] = resv;                                      // cacheTempReservoir :234
        SurfRec sr;
        sr.position = toR(state.position); sr.normal = toR(state.normal); sr.ffnormal = toR(state.ffnormal); sr.emission = toR(state.mat.emission);
        sr.roughness = state.mat.roughness; sr.metallic = state.mat.metallic; sr.matID = state.matID; sr.seed = c.seed;
        F.surf[index] = sr;
        deferred = true;
      } else {
        const rt_light_sample ls = resv.lightSample;
        if(!resvInvalidW(resv.weight)) {
          f3 LiBsdf = mk3(ls.Li) * metallicWorkflowBSDF(state.mat, state.ffnormal, wo, mk3(ls.wi));
          direct = LiBsdf / resvToScalar(LiBsdf) * resv.weight / float(resv.num);
        }
      }
    }
    if(rt_isnan(direct.x) || rt_isnan(direct.y) || rt_isnan(direct.z)) direct = mk3(0.0f);
    radiance = HDRToLDR(c.clampRadiance(state.mat.emission + direct));
  }
  if(spatial) F.status[index] = deferred ? 1u : 0u;
  if(!deferred) {
    const f3 pixelColor = c.clampRadiance(radiance);
    storeImg(F.thisDirectResult, F, px, mk4(pixelColor, 1.0f));  // :286
  }
}

#if RT_LAT
// ---- latency build: a workgroup of NW waves per 8x8 tile.  Wave 0 owns the 64 pixels (everything that is not traversal runs there, one lane per pixel,
// exactly the code of the throughput build); rays go through the workgroup's LDS pool and every wave traces them eight lanes per ray (tracePoolWide).
struct WideLds { uint2* stacks; float4* pool; unsigned char* list; uint32_t* ctrl; };   // ctrl: [0] rays listed, [1] cursor, [2] wave 0 has finished
__host__ __device__ inline size_t wideWaveStride(int stackEntries) { return size_t(stackEntries) * WIDE_RAYS + GANG_BOX_UINT2; }   // per wave: 8 stack columns + the gang mailbox (uint2 units)
RT_DEV WideLds wideLds(uint2* base, int stackEntries, int nWaves)
{
  WideLds L;
  L.stacks = base;
  L.pool = reinterpret_cast<float4*>(base + size_t(nWaves) * wideWaveStride(stackEntries));
  L.list = reinterpret_cast<unsigned char*>(L.pool + 128 * POOL_SLOT_F4);
  L.ctrl = reinterpret_cast<uint32_t*>(L.list + 128);
  return L;
}
inline size_t wideLdsBytes(int stackEntries, int nWaves) { return size_t(nWaves) * wideWaveStride(stackEntries) * sizeof(uint2) + 128 * 32 + 128 + 16; }
// wave 0, all 64 lanes: list the slots its lanes filled (slot 2 * lane: closest-hit ray, 2 * lane + 1: any-hit ray)
RT_DEV void poolPublish(const WideLds& L, bool hasC, bool hasS)
{
  const int lane = int(threadIdx.x) & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned long long mC = __ballot(hasC ? 1 : 0), mS = __ballot(hasS ? 1 : 0);
  const int nC = __popcll(mC);
  if(hasC) L.list[__popcll(mC & lt)] = (unsigned char)(lane * 2);
  if(hasS) L.list[nC + __popcll(mS & lt)] = (unsigned char)((lane * 2 + 1) | 0x80);
  if(lane == 0) { L.ctrl[0] = uint32_t(nC + __popcll(mS)); L.ctrl[1] = 0u; L.ctrl[2] = 0u; }
}
// every wave of the workgroup
RT_DEV void groupTrace(const DevScene& S, const WideLds& L, TravCounters& tc)
{
  __syncthreads();
  const int wave = int(threadIdx.x) >> 6;
  tracePoolWide(S, L.pool, L.list, int(L.ctrl[0]), &L.ctrl[1], L.stacks + size_t(wave) * wideWaveStride(S.stackEntries), tc);
  __syncthreads();
}

#ifndef RT_LAT_DIRECT_WAVES
#define RT_LAT_DIRECT_WAVES 4   // 128 VGPRs: two workgroups per CU (151 VGPRs and one workgroup at 2: 15-30 % slower on every band, scripts/variants_ab.sh with -DRT_LAT_DIRECT_WAVES=2)
#endif
__global__ __launch_bounds__(512, RT_LAT_DIRECT_WAVES) void k_direct_stage(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY)
{
  extern __shared__ uint2 s_stack[];
  const TileCoord tile = tileOf(tilesX, tilesY);
  if(!tile.valid) return;
  const int wave = int(threadIdx.x) >> 6, lane = int(threadIdx.x) & 63;
  const WideLds L = wideLds(s_stack, S.stackEntries, int(blockDim.x) >> 6);
  const i2 px{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + (lane >> 3)};
  const bool mine = wave == 0 && !(px.x >= st.size.x || px.y >= rowEnd);
#if RT_WAVEPROF
  const uint64_t p0 = clock64(), w0 = wall_clock64();
  uint64_t p1 = 0, p2 = 0, p3 = 0, p4 = 0;
#endif
  Ctx c(S, st, cam, nullptr);
  Ray r{mk3(0.0f), mk3(0.0f)};
  if(mine) {
    c.imageCoords = px;
    c.seed = tea(uint32_t(st.size.x) * uint32_t(px.y) + uint32_t(px.x), st.time);  // :279
    r = c.raySpawn(px, i2{st.size.x, st.size.y});
    c.nClosest++;
    poolPut(L.pool, lane * 2, r.origin, r.direction, RT_INFINITY, c.seed);
  }
  if(wave == 0) poolPublish(L, mine, false);
#if RT_WAVEPROF
  p1 = clock64();
#endif
  groupTrace(S, L, c.tc);
#if RT_WAVEPROF
  p2 = clock64();
#endif
  DirectCont K;
  Ray shadowRay{mk3(0.0f), mk3(0.0f)};
  float shadowDist = 0.0f;
  bool wantShadow = false;
  if(mine) {
    c.hit = poolGet(L.pool, lane * 2);
    wantShadow = directPre(c, F, st, px, r, K, shadowRay, shadowDist);
    if(wantShadow) { c.nAny++; poolPut(L.pool, lane * 2 + 1, shadowRay.origin, shadowRay.direction, shadowDist, c.seed); }
  }
  if(wave == 0) poolPublish(L, false, wantShadow);
#if RT_WAVEPROF
  p3 = clock64();
#endif
  groupTrace(S, L, c.tc);
#if RT_WAVEPROF
  p4 = clock64();
#endif
  if(mine) directPost(c, F, st, cam, px, K, wantShadow && poolGet(L.pool, lane * 2 + 1).gid != 0xffffffffu);
#if RT_WAVEPROF
  {  // record: 0 x | 1 y | 2 workgroup cycles | 3 primary trace | 4 shadow trace | 5 node-only rounds (max over waves) | 6 rounds with triangles | 7 raygen | 8 cyc node rounds | 9 cyc tri rounds | 10 pre | 11 post | 15 ticks
    uint32_t* rec = F.waveProf + size_t(blockIdx.x) * 16;
    const uint64_t p5 = clock64();
    if(wave == 0 && lane == 0) {
      rec[0] = uint32_t(tile.x); rec[1] = uint32_t(tile.y); rec[2] = uint32_t(p5 - p0); rec[3] = uint32_t(p2 - p1); rec[4] = uint32_t(p4 - p3);
      rec[7] = uint32_t(p1 - p0); rec[10] = uint32_t(p3 - p2); rec[11] = uint32_t(p5 - p4); rec[15] = uint32_t(wall_clock64() - w0);
    }
    if(lane == 0) { atomicMax(&rec[5], c.tc.rN); atomicMax(&rec[6], c.tc.rT); atomicMax(&rec[8], c.tc.cN); atomicMax(&rec[9], c.tc.cT); atomicAdd(&rec[12], c.tc.rC); atomicAdd(&rec[13], c.tc.rN + c.tc.rT);   // 12 / (8 x 13) = share of the ray slots in use
      uint32_t* acc = F.waveProf + size_t(65535) * 16;   // launch totals: cycles by phase of the triangle step, [7] = steps, [8] = cycles of rounds with a triangle step
      for(int k = 0; k < 7; k++) atomicAdd(&acc[k], c.tc.ph[k] >> 4);
      atomicAdd(&acc[7], c.tc.ph[7]);
      atomicAdd(&acc[8], c.tc.cT >> 4);
      for(int k = 0; k < 3; k++) atomicAdd(&acc[9 + k], c.tc.nph[k] >> 4);
      atomicAdd(&acc[12], c.tc.nph[3]); }
  }
#endif
}
#else
__global__ __launch_bounds__(64, RT_DIRECT_LB) RT_DIRECT_VGPR_ATTR void k_direct_stage(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY)
{
  extern __shared__ uint2 s_stack[];
#if RT_WAVEPROF
  const uint64_t prof_c0 = clock64(), prof_w0 = wall_clock64();
#endif
  const uint64_t row_c0 = F.rowCost ? clock64() : 0ull;
  const TileCoord tile = tileOfOrdered(F.rowOrder, tilesX, tilesY);
  if(!tile.valid) return;
  const int lane = int(threadIdx.x);
  const i2 px{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + (lane >> 3)};
  if(px.x >= st.size.x || px.y >= rowEnd) return;

  Ctx c(S, st, cam, s_stack + lane);
  c.imageCoords = px;
  c.seed = tea(uint32_t(st.size.x) * uint32_t(px.y) + uint32_t(px.x), st.time);  // :279
  const Ray r = c.raySpawn(px, i2{st.size.x, st.size.y});
  c.ClosestHit(r);
  DirectCont K;
  Ray shadowRay{mk3(0.0f), mk3(0.0f)};
  float shadowDist = 0.0f;
  const bool wantShadow = directPre(c, F, st, px, r, K, shadowRay, shadowDist);
  const bool occluded = wantShadow && c.AnyHit(shadowRay, shadowDist);
  directPost(c, F, st, cam, px, K, occluded);
  flushCounters(F, c);
#if RT_WAVEPROF
  waveProfFlush(F, c, tile.x, tile.y, prof_c0, prof_w0);
#endif
  // what this tile cost, for the order of the NEXT launch's tile rows (every in-image lane of the wave arrives here together: one atomic per wave)
  if(F.rowCost) {
    const unsigned long long act = __ballot(1);
    if(lane == __builtin_ctzll(act)) atomicAdd(&F.rowCost[tile.y], uint32_t((clock64() - row_c0) >> 8));
  }
}

// tile rows by descending cost of the previous launch (stable: equal costs keep the screen order, the first frame's zeros give the identity); clears the costs
__global__ __launch_bounds__(256) void k_row_order(uint32_t* rowCost, uint16_t* rowOrder, int tilesY)
{
  extern __shared__ uint32_t s_cost[];
  for(int i = int(threadIdx.x); i < tilesY; i += int(blockDim.x)) s_cost[i] = rowCost[i];
  __syncthreads();
  for(int i = int(threadIdx.x); i < tilesY; i += int(blockDim.x)) {
    const uint32_t ci = s_cost[i];
    int rank = 0;
    for(int j = 0; j < tilesY; j++) { const uint32_t cj = s_cost[j]; rank += (cj > ci || (cj == ci && j < i)) ? 1 : 0; }
    rowOrder[rank] = uint16_t(i);
    rowCost[i] = 0u;
  }
}
#endif

#if !RT_LAT
// Second half of direct_stage.comp's ReSTIRDirect for the spatial modes (:86-121, 236-262): two rounds of five neighbour
// merges from the cached reservoirs, the final merge and the shading.  No rays.
// The 10 x 10 block of cached reservoirs around the 8 x 8 tile (every neighbour lies within one pixel) is staged in LDS once and the
// ten neighbour reads of a pixel come from there (north_star design list: LDS neighbour-reservoir tile).  Against the global gather,
// Sponza-class 1080p, serial: 135 -> 90 us per launch, VMEM reads 1.46 M -> 0.46 M, SQ_WAIT_ANY -62 % (profiles/r02_spatial_lds_ab.txt).
__global__ __launch_bounds__(64) void k_direct_spatial(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY)
{
  __shared__ uint32_t s_nb[100 * 9];
  const TileCoord tile = tileOf(tilesX, tilesY);
  if(!tile.valid) return;
  const int lane = int(threadIdx.x);
  const i2 px{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + (lane >> 3)};
  const int nbx0 = tile.x * 8 - 1, nby0 = rowBegin + tile.y * 8 - 1;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(F.tempDirectResv);
    for(int k = lane; k < 100; k += 64) {
      const int rx = nbx0 + k % 10, ry = nby0 + k / 10;
      if(rx >= 0 && ry >= 0 && rx < st.size.x && ry < st.size.y) {
        const size_t o = (size_t(ry) * st.size.x + rx) * 9;
#pragma unroll
        for(int w = 0; w < 9; w++) s_nb[k * 9 + w] = src[o + w];
      }
    }
    __syncthreads();
  }
  if(px.x >= st.size.x || px.y >= rowEnd) return;
  const size_t index = size_t(px.y) * st.size.x + px.x;
  if(F.status[index] != 1u) return;
  Ctx c(S, st, cam, nullptr);
  c.imageCoords = px;
  const SurfRec sr = F.surf[index];
  c.seed = sr.seed;
  rt_direct_reservoir resv = F.tempDirectResv[index];
  const uint4 g = F.thisG[index];
  const f3 pnorm = decompress_unit_vec(g.y);          // loadThisGeometryInfo(imageCoords, ...): the pixel's own G-buffer entry
  const float pdepth = rt_u2f(g.x), depth = pdepth;   // prd.hitT is what encodeGeometryInfo stored in .x
  const f3 normal = mk3(sr.normal), ffnormal = mk3(sr.ffnormal);
  const i2 size{st.size.x, st.size.y};
  rt_direct_reservoir spatial = zeroDirectResv();
  for(int round = 0; round < 2; round++) {
    rt_direct_reservoir agg = zeroDirectResv();
    bool valid = false;
    for(int i = 0; i < 5; i++) {
      const float r0 = rnd(c.seed), r1 = rnd(c.seed);
      const f2 p = toConcentricDisk(mk2(r0, r1));
      const i2 q{rt_ftoi((float(px.x) + p.x) + 0.5f), rt_ftoi((float(px.y) + p.y) + 0.5f)};
      if(!inBound(q, size)) continue;
      if(dot(normal, pnorm) < 0.5f || rt_abs(depth - pdepth) > depth * 0.1f) continue;
      rt_direct_reservoir nb;
      {
        const uint32_t* e = s_nb + ((q.y - nby0) * 10 + (q.x - nbx0)) * 9;
        uint32_t* d = reinterpret_cast<uint32_t*>(&nb);
#pragma unroll
        for(int w = 0; w < 9; w++) d[w] = e[w];
      }
      if(!resvInvalidW(nb.weight)) { resvMerge(agg, nb, rnd(c.seed)); valid = true; }
    }
    if(valid && !resvInvalidW(agg.weight)) resvMerge(spatial, agg, rnd(c.seed));
  }
  if(!resvInvalidW(spatial.weight)) resvMerge(resv, spatial, rnd(c.seed));
  const Ray r = c.raySpawn(px, size);
  const f3 wo = -r.direction;
  Material mat;
  mat.albedo = mk3(1.0f); mat.emission = mk3(sr.emission); mat.metallic = sr.metallic; mat.roughness = sr.roughness; mat.ior = 0.f; mat.transmission = 0.f;
  f3 direct = mk3(0.0f);
  const rt_light_sample ls = resv.lightSample;
  if(!resvInvalidW(resv.weight)) {
    const f3 LiBsdf = mk3(ls.Li) * metallicWorkflowBSDF(mat, ffnormal, wo, mk3(ls.wi));
    direct = LiBsdf / resvToScalar(LiBsdf) * resv.weight / float(resv.num);
  }
  if(rt_isnan(direct.x) || rt_isnan(direct.y) || rt_isnan(direct.z)) direct = mk3(0.0f);
  const f3 radiance = HDRToLDR(c.clampRadiance(mat.emission + direct));
  storeImg(F.thisDirectResult, F, px, mk4(c.clampRadiance(radiance), 1.0f));
}

#endif  // !RT_LAT

// ------------------------------------------------------------------------------------------------------------
// direct_gen.comp / direct_reuse.comp
// ------------------------------------------------------------------------------------------------------------
#if !RT_LAT
__global__ __launch_bounds__(64) void k_direct_gen(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY)
{
  extern __shared__ uint2 s_stack[];
  const TileCoord tile = tileOf(tilesX, tilesY);
  if(!tile.valid) return;
  const int lane = int(threadIdx.x);
  const i2 px{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + (lane >> 3)};
  if(px.x >= st.size.x || px.y >= rowEnd) return;
  Ctx c(S, st, cam, s_stack + lane);
  c.imageCoords = px;
  const size_t index = size_t(px.y) * st.size.x + px.x;
  c.seed = tea(uint32_t(st.size.x) * uint32_t(px.y) + uint32_t(px.x), st.time);  // direct_gen.comp:146
  const Ray r = c.raySpawn(px, i2{st.size.x, st.size.y});
  c.ClosestHit(r);
  rt_direct_reservoir resv = zeroDirectResv();
  uint32_t lid = 0xffffffffu;
  if(c.hit.t >= RT_INFINITY * 0.8f) {  // :86
    uint4 g = make_uint4(rt_f2u(RT_INFINITY), 0u, 0u, RT_INVALID_MAT_ID);
    updateGeometryAlbedo(g, c.EnvRadiance(r.direction));
    F.thisG[index] = g;
    storeMotion(F, px, i2{0, 0});
  } else {
    State state = c.GetState(r.direction);
    c.GetMaterials(state, r);
    storeMotion(F, px, createMotionIndex(c, state.position));
    uint4 g = encodeGeometryInfo(state, c.hit.t);
    if(st.debugging_mode > RT_DBG_INDIRECT_STAGE) updateGeometryAlbedo(g, c.DebugInfo(state));
    else if(state.isEmitter) updateGeometryAlbedo(g, state.mat.emission);
    else {
      state.mat.albedo = mk3(1.0f);
      risCandidates(c, state, -r.direction, resv, lid);
    }
    F.thisG[index] = g;
  }
  F.thisDirectResv[index] = resv;
  F.thisLightId[index] = lid;
  flushCounters(F, c);
}

__global__ __launch_bounds__(64) void k_direct_reuse(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY)
{
  const TileCoord tile = tileOf(tilesX, tilesY);
  if(!tile.valid) return;
  const int lane = int(threadIdx.x);
  const i2 px{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + (lane >> 3)};
  if(px.x >= st.size.x || px.y >= rowEnd) return;
  Ctx c(S, st, cam, nullptr);
  c.imageCoords = px;
  const int index = px.y * st.size.x + px.x;
  c.seed = tea(uint32_t(index + st.size.x * st.size.y), st.time);  // direct_reuse.comp:109
  const Ray ray = c.raySpawn(px, i2{st.size.x, st.size.y});
  GState state; float depth;
  if(!stateFromGBuffer(F.thisG[index], ray, state, depth)) { storeImg(F.thisDirectResult, F, px, mk4(0, 0, 0, 0)); return; }
  f3 direct = mk3(0.0f);
  rt_direct_reservoir resv = F.thisDirectResv[index];
  uint32_t lid = F.thisLightId[index];
  const rt_light_sample ls = resv.lightSample;  // captured before the merge (direct_reuse.comp:124)
  const i2 motionIdx = loadMotion(F, px);
  if(st.ReSTIRState == RT_RESTIR_TEMPORAL || st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL) {
    const float reprojDepth = length(mk3(cam.lastPosition) - state.position);
    rt_direct_reservoir temporal; uint32_t tlid = 0xffffffffu;
    if(findTemporalNeighborDirect(F, st, state.normal, reprojDepth, state.matID, motionIdx, temporal, tlid)) {
      if(!resvInvalidW(temporal.weight)) { if(resvMerge(resv, temporal, rnd(c.seed))) lid = tlid; }
    }
  }
  if(!resvInvalidW(resv.weight)) direct = mk3(ls.Li);  // :143
  resvClamp(resv, st.RISSampleNum * st.reservoirClamp);
  if(resvInvalidW(resv.weight)) { resv.num = 0; resv.weight = 0.f; }
  if(rt_isnan(direct.x) || rt_isnan(direct.y) || rt_isnan(direct.z)) direct = mk3(0.0f);
  F.thisDirectResv[index] = resv;
  F.thisLightId[index] = lid;
  storeImg(F.thisDirectResult, F, px, mk4(HDRToLDR(c.clampRadiance(direct)), 1.0f));
}

#endif  // !RT_LAT

// ------------------------------------------------------------------------------------------------------------
// indirect_stage.comp
// ------------------------------------------------------------------------------------------------------------
// Longest-first tile order for the indirect stage: the 25 % multi-bounce tiles (indirect_stage.comp:283-288) run up to
// maxDepth bounces while the rest stop after one, so they are dispatched first and the short tiles fill the tail.
// One thread per tile recomputes the tile flag exactly as the stage does and appends the tile to its XCD's list
// (multi-bounce from the front, single-bounce from the back).  Only the order changes, never a result.
__global__ __launch_bounds__(256) void k_ind_tile_order(rt_state st, int rowBegin, int tilesX, int tilesY, int cap, uint32_t* lists, uint32_t* counts)
{
  // one workgroup per XCD list; list positions come from LDS counters (8 k global atomics on 16 addresses cost 95 us)
  __shared__ uint32_t s_cnt[2];
  const int xcd = int(blockIdx.x);
  if(threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  uint32_t* list = lists + size_t(xcd) * cap;
  const int indW = st.size.x / 2;
  const int G = tileChunk(tilesX, tilesY), nTiles = tilesX * tilesY;
  const int chunks = (nTiles + G - 1) / G;
  for(int s = xcd; s < chunks; s += 8) {
    for(int i = int(threadIdx.x); i < G; i += int(blockDim.x)) {
      const int ti = s * G + i;
      if(ti >= nTiles) continue;
      const int ty = ti / tilesX, tx = ti - ty * tilesX;
      uint32_t seed = tea(uint32_t(indW) * uint32_t(rowBegin + ty * 8) + uint32_t(tx * 8), st.time);
      const bool mb = rnd(seed) < 0.25f;
      const uint32_t t = uint32_t(ty * tilesX + tx);
      if(mb) list[atomicAdd(&s_cnt[0], 1u)] = t;
      else list[cap - 1 - int(atomicAdd(&s_cnt[1], 1u))] = t;
    }
  }
  __syncthreads();
  if(threadIdx.x < 2) counts[xcd * 2 + threadIdx.x] = s_cnt[threadIdx.x];
  if(threadIdx.x == 2) counts[16 + xcd] = 0u;   // the pixel cursor of this XCD's persistent multi-bounce waves (indirectMultiBouncePersistent)
}

// ReSTIRIndirect, indirect_stage.comp:228-268 (+ findTemporalNeighbor :74-108): temporal lookup, reservoir update, shading
// of the half-resolution pixel.  Shared by the generic indirect kernel body and the multi-tile single-bounce body.
RT_DEV void restirIndirectFinish(Ctx& c, const DevFrame& F, const rt_state& st, const rt_scene_camera& cam, i2 px, i2 indSize, const GState& primState, f3 primWo,
                                 rt_gi_sample gi, float primSamplePdf)
{
  f3 indirect = mk3(0.0f);
  rt_indirect_reservoir resv;
  resv.giSample.L = rt_vec3{0, 0, 0}; resv.giSample.xv = rt_vec3{0, 0, 0}; resv.giSample.nv = rt_vec3{0, 0, 0};
  resv.giSample.xs = rt_vec3{0, 0, 0}; resv.giSample.ns = rt_vec3{0, 0, 0}; resv.giSample.pHat = 0.f;
  resv.num = 0; resv.weight = 0.f; resv.bigW = 0.f;
  if(st.ReSTIRState == RT_RESTIR_TEMPORAL || st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL) {
    const float reprojDepth = length(mk3(cam.lastPosition) - primState.position);
    const i2 motionIdx = loadMotion(F, i2{px.x * 2, px.y * 2});
    if(motionIdx.x >= 0 && motionIdx.x < F.W && motionIdx.y >= 0 && motionIdx.y < F.H && (motionIdx.y < F.histRow0 || motionIdx.y >= F.histRow1)) *F.histMiss = 1u;
    const uint4 lg = loadG(F.lastG, F, motionIdx);
    const f3 pnorm = decompress_unit_vec(lg.y);
    const float pdepth = rt_u2f(lg.x);
    const uint32_t matHash = lg.w & 0xFF000000u;
    const i2 coord{motionIdx.x / 2, motionIdx.y / 2};
    if(inBound(coord, indSize)) {
      if(hash8bit(primState.matID) == matHash) {
        if(dot(primState.ffnormal, pnorm) > 0.5f && reprojDepth < pdepth * 1.1f) resv = F.lastIndirectResv[size_t(coord.y) * indSize.x + coord.x];
      }
    }
  }
  float sampleWeight = 0.0f;
  if(GISampleValid(gi)) {
    gi.pHat = resvToScalar(mk3(gi.L));  // pHatIndirect :61-66
    sampleWeight = gi.pHat / primSamplePdf;
    if(rt_isnan(sampleWeight) || sampleWeight < 0.0f) sampleWeight = 0.0f;
  }
  {  // resvUpdate :54-60
    const float rr = rnd(c.seed);
    resv.weight += sampleWeight; resv.num += 1;
    if(rr * resv.weight < sampleWeight) resv.giSample = gi;
  }
  if(resvInvalidW(resv.weight)) { resv.num = 0; resv.weight = 0.f; resv.bigW = 0.f; }
  resvClamp(resv, st.reservoirClamp * 2);
  F.thisIndirectResv[size_t(px.y) * indSize.x + px.x] = resv;  // saveNewReservoir

  gi = resv.giSample;
  if(!resvInvalidW(resv.weight) && GISampleValid(gi)) {
    const f3 primWi = normalize(mk3(gi.xs) - mk3(gi.xv));
    Material pm = primState.mat;
    pm.albedo = mk3(1.0f);
    const float bigW = resv.weight / (resvToScalar(mk3(resv.giSample.L)) * float(resv.num));  // bigWIndirect :68-70
    indirect = mk3(gi.L) * metallicWorkflowBSDF(pm, mk3(gi.nv), primWo, primWi) * satDot(mk3(gi.nv), primWi) * bigW;
  }
  f3 pixelColor = HDRToLDR(c.clampRadiance(indirect));
  pixelColor = c.clampRadiance(pixelColor);
  storeImg(F.denoiseIndA, F, px, mk4(pixelColor, 1.0f));
}

#if !RT_LAT
// ---- single-bounce tiles, K tiles per wave ----------------------------------------------------------------------------------
// 75 % of the tiles stop after one bounce + one NEE shadow ray (TILED_MULTIBOUNCE, :283-288): per path one closest-hit ray,
// then at most one any-hit ray.  With one tile per wave the ray pool holds fewer rays than the wave has lanes (sky pixels,
// paths that left the scene), and every lane then waits for the slowest ray: 77 % of the lane-rounds of these tiles were
// idle.  Here a wave owns K such tiles; what a path carries between its three phases is small (seed, pdf, direction /
// hit point, normal, pending NEE term), so K paths per lane stay in registers, both traces run over a pool of up to 64 K
// rays, and lanes pull rays until the pool is dry.  Same arithmetic, same RNG draw order per path as the generic body.
template <int K>
RT_DEV void indirectSingleBounceTiles(const DevScene& S, const DevFrame& F, const rt_state& st, const rt_scene_camera& cam, int rowBegin, int rowEnd, int tilesX,
                                      const uint32_t* tiles /* K entries */, int nTilesHere, uint2* s_stack)
{
  const int lane = int(threadIdx.x);
  const i2 indSize{st.size.x / 2, st.size.y / 2};
  float4* pool = reinterpret_cast<float4*>(s_stack + size_t(S.stackEntries) * 64);
#if RT_WAVEPROF
  const uint64_t prof_c0 = clock64(), prof_w0 = wall_clock64();
#endif
  Ctx c(S, st, cam, s_stack + lane);
  struct Path { i2 px; uint32_t seed; float primSamplePdf; f3 a, b, pend; bool hasSurface, nvSet, shadow; };
  // a: sampleWi between phase 0 and 1, gi.xs afterwards; b: gi.ns; pend: the NEE term waiting for its visibility
  Path P[K];
  uint32_t have = 0;
  // ---- phase 0: G-buffer decode, BSDF sample, park the bounce ray (depth 1 of pathTraceIndirect, :129-226) ---------------------
#pragma unroll
  for(int k = 0; k < K; k++) {
    Path& p = P[k];
    p.hasSurface = false; p.nvSet = false; p.shadow = false; p.primSamplePdf = 0.0f; p.a = mk3(0.0f); p.b = mk3(0.0f); p.pend = mk3(0.0f);
    const int t = int(tiles[k < nTilesHere ? k : 0]);
    const int ty = t / tilesX, tx = t - ty * tilesX;
    p.px = i2{tx * 8 + (lane & 7), rowBegin + ty * 8 + (lane >> 3)};
    p.seed = tea(uint32_t(indSize.x) * uint32_t(p.px.y) + uint32_t(p.px.x), st.time);  // :280
    if(lane == 0) (void)rnd(p.seed);  // invocation 0 drew the tile flag from its own stream (:283-288); the flag is known here
    const bool inImage = k < nTilesHere && !(p.px.x >= indSize.x || p.px.y >= indSize.y || p.px.y >= rowEnd);
    if(!inImage) continue;
    const Ray ray = c.raySpawn(p.px, indSize);
    GState g0; float depth;
    if(!stateFromGBuffer(loadG(F.thisG, F, i2{p.px.x * 2, p.px.y * 2}), ray, g0, depth)) { storeImg(F.denoiseIndA, F, p.px, mk4(0, 0, 0, 0)); continue; }
    p.hasSurface = true;
    g0.position += g0.ffnormal * 2e-2f;  // :299
    Material m = g0.mat; m.albedo = mk3(1.0f);
    if(st.maxDepth < 1) continue;
    c.seed = p.seed;
    f3 wi = mk3(0.0f); float pdf = 0.0f;
    (void)c.Sample(m, -ray.direction, g0.ffnormal, wi, pdf);
    p.seed = c.seed;
    if(Ctx::IsPdfInvalid(pdf)) continue;
    p.primSamplePdf = pdf; p.nvSet = true; p.a = wi;
    c.nClosest++;
    poolPut(pool, k * 64 + lane, OffsetRay(g0.position, g0.ffnormal), wi, RT_INFINITY, p.seed);
    have |= 1u << k;
  }
  tracePoolTiles<K, false>(S, pool, have, c.stack, c.tc);
  // ---- phase 1: the hit; NEE sample of depth 2 -> shadow ray; the BSDF sample depth 2 draws before it stops (:169-176) ------
  uint32_t haveS = 0;
#pragma unroll
  for(int k = 0; k < K; k++) {
    Path& p = P[k];
    if(!((have >> k) & 1u)) continue;
    const f3 wi = p.a;
    c.hit = poolGet(pool, k * 64 + lane);
    const Ray ray0 = c.raySpawn(p.px, indSize);
    GState g0; float depth;
    stateFromGBuffer(loadG(F.thisG, F, i2{p.px.x * 2, p.px.y * 2}), ray0, g0, depth);
    g0.position += g0.ffnormal * 2e-2f;
    if(c.hit.t >= RT_INFINITY - 1e-4f) {  // :183-195, depth 1
      p.a = g0.position + wi * RT_INFINITY * 0.8f; p.b = -wi;
      continue;
    }
    const Ray ray{OffsetRay(g0.position, g0.ffnormal), wi};
    State state = c.GetState(ray.direction);
    c.GetMaterials(state, ray);
    p.a = state.position; p.b = state.ffnormal;  // gi.xs / gi.ns, emitter or not (:197-215)
    if(state.isEmitter || st.maxDepth < 2) continue;
    c.seed = p.seed;
    const f3 wo = -ray.direction;
    if(st.MIS > 0) {  // SampleDirectLight (pathtrace.glsl:185-203) minus its visibility test; throughput is 1 in these tiles
      rt_light_sample ls;
      const float lightPdf = c.SampleDirectLightNoVisibility(state.position, ls);
      if(!Ctx::IsPdfInvalid(lightPdf)) {
        const f3 lwi = mk3(ls.wi);
        const f3 so = OffsetRay(state.position, state.ffnormal);
        const float maxDist = ((ls.dist - rt_abs(so.x - state.position.x)) - rt_abs(so.y - state.position.y)) - rt_abs(so.z - state.position.z);  // Occlusion :18-22
        c.nAny++;
        poolPut(pool, k * 64 + lane, so, lwi, maxDist, c.seed);
        haveS |= 1u << k; p.shadow = true;
        const float BSDFPdf = metallicWorkflowPdf(state.mat, state.ffnormal, wo, lwi);
        const float weight = MISw(st, lightPdf, BSDFPdf);
        p.pend = mk3(ls.Li) * metallicWorkflowBSDF(state.mat, state.ffnormal, wo, lwi) * absDot(state.ffnormal, lwi) * mk3(1.0f) / lightPdf * weight;
      }
    }
    f3 w2 = mk3(0.0f); float pdf2 = 0.0f;
    (void)c.Sample(state.mat, wo, state.ffnormal, w2, pdf2);  // depth 2 samples the BSDF, then `if(!multiBounce) break`
    p.seed = c.seed;
  }
  // (slots of the bounce rays were read above; the shadow rays reuse them)
  tracePoolTiles<K, true>(S, pool, haveS, c.stack, c.tc);
  // ---- phase 2: visibility of the NEE term, then ReSTIRIndirect ------------------------------------------------------------------
#pragma unroll
  for(int k = 0; k < K; k++) {
    Path& p = P[k];
    if(!p.hasSurface) continue;
    rt_gi_sample gi = newGISample();
    const Ray ray0 = c.raySpawn(p.px, indSize);
    GState primState; float depth;
    stateFromGBuffer(loadG(F.thisG, F, i2{p.px.x * 2, p.px.y * 2}), ray0, primState, depth);
    primState.position += primState.ffnormal * 2e-2f;
    if(p.nvSet) { gi.xv = toR(primState.position); gi.nv = toR(primState.ffnormal); gi.xs = toR(p.a); gi.ns = toR(p.b); }
    if(p.shadow && poolGet(pool, k * 64 + lane).gid == 0xffffffffu) gi.L = toR(mk3(gi.L) + p.pend);  // not occluded
    c.seed = p.seed;
    c.imageCoords = p.px;
    restirIndirectFinish(c, F, st, cam, p.px, indSize, primState, -ray0.direction, gi, p.primSamplePdf);
  }
#if RT_WAVEPROF
  // (round 5: the K-tile waves were missing from the profile — 75 % of the stage's tiles; their record is keyed by the first tile, bit 17 marks the kind)
  { const int t0 = int(tiles[0]); waveProfFlush(F, c, (t0 % tilesX) | 0x20000, t0 / tilesX, prof_c0, prof_w0); }
#endif
  flushCounters(F, c);
}

// ---- multi-bounce tiles, persistent waves with per-lane path regeneration (round 6) ---------------------------------------------------------------
// A wave that owns ONE multi-bounce tile traces its paths vertex by vertex: 64 + 0 rays at the first vertex, then whatever survived — the pool runs dry (round 5:
// 47 % of the traversal rounds of the stage executed <= 8 lanes, 36 % had <= 8 LIVE lanes; ordering the pool did nothing, profiles/r05_pool_order_ab.txt).  Here a
// wave is not tied to a tile: a lane whose path has ended finishes its pixel (ReSTIRIndirect + the stores are per pixel, seeds are per pixel) and takes the next
// pixel of its XCD's multi-bounce list from an atomic cursor, so every vertex round of the wave parks rays for (nearly) all 64 lanes — lanes sit at different depths
// of different pixels.  No ray queue in HBM (the wavefront organisation that lost in round 1): a path lives in its lane's registers from its first vertex to its
// reservoir.  Per path the arithmetic and the order of the RNG draws are those of the generic body below, hence the same bits.  `persist` tiles' worth of pixels per
// wave (RESTIR_IND_PERSIST; 0 = the generic body).
RT_DEV void indirectMultiBouncePersistent(const DevScene& S, const DevFrame& F, const rt_state& st, const rt_scene_camera& cam, int rowBegin, int rowEnd, int tilesX,
                                          const uint32_t* list /* this XCD's multi-bounce tiles */, int nTiles, uint32_t* cursor, uint2* s_stack)
{
  const int lane = int(threadIdx.x);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const i2 indSize{st.size.x / 2, st.size.y / 2};
  float4* pool = reinterpret_cast<float4*>(s_stack + size_t(S.stackEntries) * 64);
#if RT_WAVEPROF
  const uint64_t prof_c0 = clock64(), prof_w0 = wall_clock64();
#endif
  Ctx c(S, st, cam, s_stack + lane);
  const uint32_t total = uint32_t(nTiles) * 64u;
  // the path a lane carries between two vertex rounds
  bool have = false, alive = false;      // have: the lane owns a pixel with a surface; alive: its path goes on
  i2 px{0, 0};
  int depthI = 1;
  f3 throughput = mk3(4.0f);
  Ray ray{mk3(0.0f), mk3(0.0f)};
  rt_gi_sample gi = newGISample();
  float primSamplePdf = 0.0f;
  f3 vPos = mk3(0.0f), vFfn = mk3(0.0f);  // the vertex: position (primary: offset along its normal, :299), shading normal, material
  Material vMat; vMat.albedo = vMat.emission = mk3(0.0f); vMat.metallic = vMat.ior = vMat.roughness = vMat.transmission = 0.0f;
  bool exhausted = false;                 // wave-uniform: the cursor has passed the end of the list
  for(;;) {
    // ---- 1. lanes without a path take the next pixels of the list (again while pixels without a surface — sky — leave lanes empty) ----------------------------
    while(!exhausted) {
      const unsigned long long needM = __ballot(have ? 0 : 1);
      if(needM == 0ull) break;
      const int nNeed = __popcll(needM);
      uint32_t base = 0u;
      if(lane == __builtin_ctzll(needM)) base = atomicAdd(cursor, uint32_t(nNeed));
      base = uint32_t(__builtin_amdgcn_readlane(int(base), __builtin_ctzll(needM)));
      if(base + uint32_t(nNeed) >= total) exhausted = true;
      const uint32_t idx = base + uint32_t(__popcll(needM & lt));
      if(!have && idx < total) {
        const int t = int(list[idx >> 6]), it = int(idx & 63u);
        const int ty = t / tilesX, tx = t - ty * tilesX;
        px = i2{tx * 8 + (it & 7), rowBegin + ty * 8 + (it >> 3)};
        c.seed = tea(uint32_t(indSize.x) * uint32_t(px.y) + uint32_t(px.x), st.time);  // :280
        c.imageCoords = px;
        if(it == 0) (void)rnd(c.seed);  // invocation 0 of the workgroup drew the tile flag from its own stream (:283-288); the flag is known here (the list)
        const bool inImage = !(px.x >= indSize.x || px.y >= indSize.y || px.y >= rowEnd);
        if(inImage) {
          ray = c.raySpawn(px, indSize);
          GState g0; float depth;
          if(stateFromGBuffer(loadG(F.thisG, F, i2{px.x * 2, px.y * 2}), ray, g0, depth)) {
            have = true; alive = st.maxDepth >= 1;   // (the depth loop of :129-226 does not run at all with maxDepth 0)
            depthI = 1;
            throughput = mk3(4.0f);
            gi = newGISample();
            primSamplePdf = 0.0f;
            vPos = g0.position + g0.ffnormal * 2e-2f;  // :299
            vFfn = g0.ffnormal; vMat = g0.mat; vMat.albedo = mk3(1.0f);
          } else storeImg(F.denoiseIndA, F, px, mk4(0, 0, 0, 0));
        }
      }
    }
    if(__ballot(have ? 1 : 0) == 0ull) break;
    // ---- 2. one path vertex per lane (pathTraceIndirect, :129-226; the body of the generic kernel's depth loop) ------------------------------------------------
    bool hasShadow = false, hasBounce = false;
    f3 pendingAdd = mk3(0.0f), sampleWi = mk3(0.0f);
    float samplePdf = 0.0f;
    if(alive) {
      const f3 wo = -ray.direction;
      if(depthI > 1 && st.MIS > 0) {  // SampleDirectLight (pathtrace.glsl:185-203) minus its visibility test
        rt_light_sample ls;
        const float lightPdf = c.SampleDirectLightNoVisibility(vPos, ls);
        if(!Ctx::IsPdfInvalid(lightPdf)) {
          const f3 wi = mk3(ls.wi);
          const f3 so = OffsetRay(vPos, vFfn);
          const float maxDist = ((ls.dist - rt_abs(so.x - vPos.x)) - rt_abs(so.y - vPos.y)) - rt_abs(so.z - vPos.z);  // Occlusion :18-22
          c.nAny++;
          poolPut(pool, lane * 2 + 1, so, wi, maxDist, c.seed);
          hasShadow = true;
          const float BSDFPdf = metallicWorkflowPdf(vMat, vFfn, wo, wi);
          const float weight = MISw(st, lightPdf, BSDFPdf);
          pendingAdd = mk3(ls.Li) * metallicWorkflowBSDF(vMat, vFfn, wo, wi) * absDot(vFfn, wi) * throughput / lightPdf * weight;
        }
      }
      const f3 sampleBSDF = c.Sample(vMat, wo, vFfn, sampleWi, samplePdf);
      if(Ctx::IsPdfInvalid(samplePdf)) alive = false;
      else {
        if(depthI > 1) throughput *= sampleBSDF / samplePdf * absDot(vFfn, sampleWi);
        else {
          primSamplePdf = samplePdf;
          gi.xv = toR(vPos);
          gi.nv = toR(vFfn);
        }
        ray.origin = OffsetRay(vPos, vFfn);
        ray.direction = sampleWi;
        c.nClosest++;
        poolPut(pool, lane * 2, ray.origin, ray.direction, RT_INFINITY, c.seed);
        hasBounce = true;
      }
    }
    // ---- 3. the wave's rays -------------------------------------------------------------------------------------------------------------------------------------
    tracePool(S, pool, hasBounce, hasShadow, c.stack, c.tc);   // (returns at once when nothing was parked)
    // ---- 4. results, in the reference's order ---------------------------------------------------------------------------------------------------------------------
    if(hasShadow && poolGet(pool, lane * 2 + 1).gid == 0xffffffffu) gi.L = toR(mk3(gi.L) + pendingAdd);  // not occluded
    if(hasBounce) {
      c.hit = poolGet(pool, lane * 2);
      if(c.hit.t >= RT_INFINITY - 1e-4f) {
        if(depthI > 1) {
          float lightPdf;
          const f3 Li = c.EnvEval(sampleWi, lightPdf);
          const float weight = MISw(st, samplePdf, lightPdf);
          gi.L = toR(mk3(gi.L) + Li * throughput * weight);
        } else {
          gi.xs = toR(vPos + sampleWi * RT_INFINITY * 0.8f);
          gi.ns = toR(-sampleWi);
        }
        alive = false;
      } else {
        State state = c.GetState(ray.direction);
        c.GetMaterials(state, ray);
        if(state.isEmitter) {
          if(depthI > 1) {
            float lightPdf;
            const f3 Li = c.LightEval(state, c.hit.t, sampleWi, lightPdf);
            const float weight = MISw(st, samplePdf, lightPdf);
            gi.L = toR(mk3(gi.L) + Li * throughput * weight);
          } else {
            gi.xs = toR(state.position);
            gi.ns = toR(state.ffnormal);
          }
          alive = false;
        } else if(depthI == 1) { gi.xs = toR(state.position); gi.ns = toR(state.ffnormal); }
        vPos = state.position; vFfn = state.ffnormal; vMat = state.mat;
      }
    }
    // Russian roulette (:218-224) is compiled out in the reference (`#ifndef RR`, pathtrace.glsl:2)
    if(alive) { depthI++; if(depthI > st.maxDepth) alive = false; }
    // ---- 5. paths that ended: ReSTIRIndirect of their pixel; the lane is free for the next one -----------------------------------------------------------------------
    if(have && !alive) {
      const Ray ray0 = c.raySpawn(px, indSize);
      GState primState; float depth;
      stateFromGBuffer(loadG(F.thisG, F, i2{px.x * 2, px.y * 2}), ray0, primState, depth);
      primState.position += primState.ffnormal * 2e-2f;
      restirIndirectFinish(c, F, st, cam, px, indSize, primState, -ray0.direction, gi, primSamplePdf);
      have = false;
    }
  }
#if RT_WAVEPROF
  waveProfFlush(F, c, int(blockIdx.x & 0xffff) | 0x10000, 0, prof_c0, prof_w0);
#endif
  flushCounters(F, c);
}

#endif  // !RT_LAT

// 5 waves/SIMD (96 VGPRs, a few more spills) instead of 4: the stage alone is no faster, but with frames in flight its waves
// share the SIMDs with the next frame's direct stage and the frame is 2.7 % shorter (3, 4, 6 measured: 3.96 / 3.47 / 3.42 vs 3.37 ms)
#ifndef RT_INDIRECT_LB
#define RT_INDIRECT_LB 5
#endif
#ifndef RT_IND_PERSIST_DEFAULT
#define RT_IND_PERSIST_DEFAULT 0
#endif
#if RT_LAT
// latency build: a workgroup of NW waves per half-res tile; wave 0 runs the paths of the 64 pixels (the body below), the other waves only serve the
// workgroup's ray pool: after every path vertex wave 0 lists the vertex's rays, all waves trace them eight lanes per ray, wave 0 goes on shading
// (4 waves per SIMD = 128 VGPRs: two of these workgroups per CU, and beside one of them two direct-stage waves per SIMD — with 165 registers the next
//  frame's direct stage, which runs beside this kernel when frames are in flight, had one wave slot per SIMD left: profiles/r03_mgpu_period_ab.txt)
__global__ __launch_bounds__(512, 4) void k_indirect_stage(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY, int cap,
                                                       const uint32_t* lists, uint32_t* counts, int subShift, int sbK, int genericBlocks, int persist)
#else
__global__ __launch_bounds__(64, RT_INDIRECT_LB) RT_INDIRECT_VGPR_ATTR void k_indirect_stage(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY, int cap,
                                                          const uint32_t* lists, uint32_t* counts, int subShift, int sbK, int genericBlocks, int persist)
#endif
{
  // subShift > 0 (small launches: row bands of a multi-GPU frame, small images): a tile is split over 2 or 4 waves that own
  // 32 / 16 of its pixels each; the other lanes of each wave have no path and only serve the wave's ray pool.  With fewer
  // paths per wave the majority-vote loop runs fewer rounds per ray, and the idle SIMDs of an under-filled chip get work:
  // the latency of the longest multi-bounce tile — the floor of a small launch — drops.  Results do not change.
  extern __shared__ uint2 s_stack[];
#if RT_WAVEPROF
  const uint64_t prof_c0 = clock64(), prof_w0 = wall_clock64();
#endif
  TileCoord tile;
  const int part = (int(blockIdx.x) >> 3) & ((1 << subShift) - 1);
  {
    const int L = int(blockIdx.x), xcd = L & 7;
    const int nf = int(counts[xcd * 2]), nb = int(counts[xcd * 2 + 1]);
#if !RT_LAT
    if(sbK > 0 && L >= genericBlocks) {  // single-bounce tiles of this XCD, sbK per wave (from the back of its list)
      const int w = (L - genericBlocks) >> 3, first = w * sbK;
      if(first >= nb) return;
      uint32_t t[3] = {0u, 0u, 0u};
      const int n = min(sbK, nb - first);
      for(int k = 0; k < n; k++) t[k] = lists[size_t(xcd) * cap + (cap - 1 - (first + k))];
      if(sbK == 3) indirectSingleBounceTiles<3>(S, F, st, cam, rowBegin, rowEnd, tilesX, t, n, s_stack);
      else indirectSingleBounceTiles<2>(S, F, st, cam, rowBegin, rowEnd, tilesX, t, n, s_stack);  // (4 per wave was measured: slower, spills)
      return;
    }
    if(sbK > 0 && persist > 0) {  // multi-bounce tiles of this XCD: persistent waves, `persist` tiles' worth of pixels each, paths regenerated per lane
      if((L >> 3) * persist >= nf) return;
      indirectMultiBouncePersistent(S, F, st, cam, rowBegin, rowEnd, tilesX, lists + size_t(xcd) * cap, nf, counts + 16 + xcd, s_stack);
      return;
    }
#endif
    (void)persist;
    const int k = (L >> 3) >> subShift;
    tile.valid = sbK > 0 ? (k < nf) : (k < nf + nb);
    const uint32_t t = tile.valid ? lists[size_t(xcd) * cap + (k < nf ? k : cap - 1 - (k - nf))] : 0u;
    tile.y = int(t) / tilesX; tile.x = int(t) - tile.y * tilesX;
  }
  if(!tile.valid) return;
#if RT_LAT
  const WideLds WL = wideLds(s_stack, S.stackEntries, int(blockDim.x) >> 6);
  if((int(threadIdx.x) >> 6) != 0) {   // helper waves: trace whatever wave 0 lists until it says it is done
    TravCounters htc{};
    for(;;) {
      __syncthreads();
      if(WL.ctrl[2] != 0u) break;
      tracePoolWide(S, WL.pool, WL.list, int(WL.ctrl[0]), &WL.ctrl[1], WL.stacks + size_t(int(threadIdx.x) >> 6) * wideWaveStride(S.stackEntries), htc);
      __syncthreads();
    }
    return;
  }
#endif
  const int lane = int(threadIdx.x);
  const i2 indSize{st.size.x / 2, st.size.y / 2};
  const int rowsPerPart = 8 >> subShift;
  const bool hasPixel = (lane >> 3) < rowsPerPart;
  const i2 px{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + part * rowsPerPart + (lane >> 3)};
#if RT_LAT
  Ctx c(S, st, cam, nullptr);
#else
  Ctx c(S, st, cam, s_stack + lane);
#endif
  c.imageCoords = px;
  c.seed = tea(uint32_t(indSize.x) * uint32_t(px.y) + uint32_t(px.x), st.time);  // :280
  // TILED_MULTIBOUNCE (:283-288): invocation 0 of the workgroup draws the tile flag from its own stream (and so advances
  // it); the flag is wave-uniform here, so it costs one readfirstlane instead of shared memory + barrier.  Waves that own
  // another part of the tile recompute the flag from a copy of that pixel's stream.
  int mb = 0;
  if(lane == 0) {
    if(part == 0) mb = rnd(c.seed) < 0.25f ? 1 : 0;
    else { uint32_t s0 = tea(uint32_t(indSize.x) * uint32_t(rowBegin + tile.y * 8) + uint32_t(tile.x * 8), st.time); mb = rnd(s0) < 0.25f ? 1 : 0; }
  }
  const bool multiBounce = __builtin_amdgcn_readfirstlane(mb) != 0;
  // Lanes without a pixel / without a surface stay in the kernel: they trace rays of the other lanes (tracePool).
#if RT_LAT
  float4* pool = WL.pool;
#else
  float4* pool = reinterpret_cast<float4*>(s_stack + size_t(S.stackEntries) * 64);
#endif
  const bool inImage = hasPixel && !(px.x >= indSize.x || px.y >= indSize.y || px.y >= rowEnd);
  Ray ray = c.raySpawn(px, indSize);

  GState g0; float depth;
  bool alive = inImage && stateFromGBuffer(loadG(F.thisG, F, i2{px.x * 2, px.y * 2}), ray, g0, depth);
  const bool hasSurface = alive;
  if(inImage && !hasSurface) storeImg(F.denoiseIndA, F, px, mk4(0, 0, 0, 0));
  if(!alive) { g0 = GState{}; }
  g0.position += g0.ffnormal * 2e-2f;  // :299

  // ---- pathTraceIndirect, :129-226 --------------------------------------------------------------------------
  // Per path vertex the reference traces the NEE shadow ray, then samples the BSDF and traces the bounce ray.  Neither
  // trace advances prd.seed here (DESIGN.md deviation 1), so both rays of a vertex are known before either is traced:
  // they go to the wave's ray pool together, and the results are applied in the reference's order afterwards.
  f3 throughput = mk3(multiBounce ? 4.0f : 1.0f);
  const f3 primWo = -ray.direction;
  const GState primState = g0;
  rt_gi_sample gi = newGISample();
  float primSamplePdf = 0.0f;
  State state = zeroState();
  state.position = g0.position; state.normal = g0.normal; state.ffnormal = g0.ffnormal; state.mat = g0.mat; state.matID = g0.matID;
  state.mat.albedo = mk3(1.0f);
  for(int depthI = 1; depthI <= st.maxDepth; depthI++) {
    bool hasShadow = false, hasBounce = false;
    f3 pendingAdd = mk3(0.0f), sampleWi = mk3(0.0f);
    float samplePdf = 0.0f;
    if(alive) {
      const f3 wo = -ray.direction;
      if(depthI > 1 && st.MIS > 0) {  // SampleDirectLight (pathtrace.glsl:185-203) minus its visibility test
        rt_light_sample ls;
        const float lightPdf = c.SampleDirectLightNoVisibility(state.position, ls);
        if(!Ctx::IsPdfInvalid(lightPdf)) {
          const f3 wi = mk3(ls.wi);
          const f3 so = OffsetRay(state.position, state.ffnormal);
          const float maxDist = ((ls.dist - rt_abs(so.x - state.position.x)) - rt_abs(so.y - state.position.y)) - rt_abs(so.z - state.position.z);  // Occlusion :18-22
          c.nAny++;
          poolPut(pool, lane * 2 + 1, so, wi, maxDist, c.seed);
          hasShadow = true;
          const float BSDFPdf = metallicWorkflowPdf(state.mat, state.ffnormal, wo, wi);
          const float weight = MISw(st, lightPdf, BSDFPdf);
          pendingAdd = mk3(ls.Li) * metallicWorkflowBSDF(state.mat, state.ffnormal, wo, wi) * absDot(state.ffnormal, wi) * throughput / lightPdf * weight;
        }
      }
      const f3 sampleBSDF = c.Sample(state.mat, wo, state.ffnormal, sampleWi, samplePdf);
      if(Ctx::IsPdfInvalid(samplePdf)) alive = false;
      else if(depthI > 1 && !multiBounce) alive = false;
      else {
        if(depthI > 1) throughput *= sampleBSDF / samplePdf * absDot(state.ffnormal, sampleWi);
        else {
          primSamplePdf = samplePdf;
          gi.xv = toR(state.position);
          gi.nv = toR(state.ffnormal);
        }
        ray.origin = OffsetRay(state.position, state.ffnormal);
        ray.direction = sampleWi;
        c.nClosest++;
        poolPut(pool, lane * 2, ray.origin, ray.direction, RT_INFINITY, c.seed);
        hasBounce = true;
      }
    }
    if(__ballot((hasShadow || hasBounce) ? 1 : 0) == 0ull) break;  // wave-uniform
#if RT_LAT
    poolPublish(WL, hasBounce, hasShadow);
#if RT_WAVEPROF
    { const uint64_t g0 = clock64(); groupTrace(S, WL, c.tc); c.cycClosest += uint32_t(clock64() - g0); c.cycAny++; }   // [3] cycles in the pool traces, [4] their number
#else
    groupTrace(S, WL, c.tc);
#endif
#else
    tracePool(S, pool, hasBounce, hasShadow, c.stack, c.tc);
#endif
    if(hasShadow && poolGet(pool, lane * 2 + 1).gid == 0xffffffffu) gi.L = toR(mk3(gi.L) + pendingAdd);  // not occluded
    if(hasBounce) {
      c.hit = poolGet(pool, lane * 2);
      if(c.hit.t >= RT_INFINITY - 1e-4f) {
        if(depthI > 1) {
          float lightPdf;
          const f3 Li = c.EnvEval(sampleWi, lightPdf);
          const float weight = MISw(st, samplePdf, lightPdf);
          gi.L = toR(mk3(gi.L) + Li * throughput * weight);
        } else {
          gi.xs = toR(state.position + sampleWi * RT_INFINITY * 0.8f);
          gi.ns = toR(-sampleWi);
        }
        alive = false;
      } else {
        state = c.GetState(ray.direction);
        c.GetMaterials(state, ray);
        if(state.isEmitter) {
          if(depthI > 1) {
            float lightPdf;
            const f3 Li = c.LightEval(state, c.hit.t, sampleWi, lightPdf);
            const float weight = MISw(st, samplePdf, lightPdf);
            gi.L = toR(mk3(gi.L) + Li * throughput * weight);
          } else {
            gi.xs = toR(state.position);
            gi.ns = toR(state.ffnormal);
          }
          alive = false;
        } else if(depthI == 1) { gi.xs = toR(state.position); gi.ns = toR(state.ffnormal); }
      }
    }
    // Russian roulette (:218-224) is compiled out in the reference (`#ifndef RR`, pathtrace.glsl:2)
  }
#if RT_LAT
  if(lane == 0) WL.ctrl[2] = 1u;   // release the helper waves
  __syncthreads();
#endif
#if RT_WAVEPROF
  waveProfFlush(F, c, tile.x | (multiBounce ? 0x10000 : 0), tile.y, prof_c0, prof_w0);
#endif
  if(!hasSurface) { flushCounters(F, c); return; }

  restirIndirectFinish(c, F, st, cam, px, indSize, primState, primWo, gi, primSamplePdf);
  flushCounters(F, c);
}

#if !RT_LAT
// ------------------------------------------------------------------------------------------------------------
// denoise_common.glsl + denoise_direct.comp + denoise_indirect.comp
// ------------------------------------------------------------------------------------------------------------

// denoise_common.glsl:27-40: the direction is not re-normalised after the view transform
RT_DEV f3 cameraPosDenoise(const rt_scene_camera& cam, i2 coord, float dist, i2 imageSize)
{
  const f2 pixelCenter = mk2(float(coord.x), float(coord.y)) + 0.5f;
  const f2 inUV = pixelCenter / mk2(float(imageSize.x), float(imageSize.y));
  const f2 d = inUV * 2.0f - 1.0f;
  const f4 origin = mul(cam.viewInverse, mk4(0, 0, 0, 1));
  const f4 target = mul(cam.projInverse, mk4(d.x, d.y, 1, 1));
  const f4 direction = mul(cam.viewInverse, mk4(normalize(xyz(target)), 0));
  return xyz(origin) + xyz(direction) * dist;
}

// loadThisGeometry (denoise_common.glsl:42-55) hoisted out of the 25-tap loop: every pixel's normal, material hash and
// reconstructed position are decoded once per frame instead of once per tap per level (the reference re-derives them
// 25 x 9 times per pixel: 2 mat-vec + 2 normalisations each).  Same expressions => same bits.
template <bool IND>
__global__ __launch_bounds__(64) void k_denoise_geom(DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY)
{
  const TileCoord tile = tileOf(tilesX, tilesY);
  if(!tile.valid) return;
  const int lane = int(threadIdx.x);
  const i2 bound = IND ? i2{st.size.x / 2, st.size.y / 2} : i2{st.size.x, st.size.y};
  const i2 coord{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + (lane >> 3)};
  if(coord.x >= bound.x || coord.y >= bound.y || coord.y >= rowEnd) return;
  const i2 gc = IND ? i2{coord.x * 2, coord.y * 2} : coord;
  const uint4 g = loadG(F.thisG, F, gc);
  const f3 norm = decompress_unit_vec(g.y);
  const f3 pos = cameraPosDenoise(cam, gc, rt_u2f(g.x), bound);
  const size_t idx = size_t(coord.y) * bound.x + coord.x;
  (IND ? F.geomNh : F.geomN)[idx] = make_float4(norm.x, norm.y, norm.z, rt_u2f(g.w & 0xFF000000u));
  (IND ? F.geomPh : F.geomP)[idx] = make_float4(pos.x, pos.y, pos.z, 0.f);
}

// exp(x) for x <= 0, bit-identical to rt_exp() on that domain (same range reduction, polynomial and scaling; the
// x > 88.7 early-out of rt_exp cannot trigger) but with selects instead of branches.
RT_DEV float expNonPositive(float x)
{
  const float z = rt_floor(rt_fma(x, 1.44269504088896341f, 0.5f));
  // n = (int)z wherever it matters: z <= 0 here, and below -127 (x < -88.4) the result is forced to 0 further down, so the
  // conversion is clamped instead of range-checked; a NaN clamps to -127 and e = NaN * 0
  const int n = int(fmaxf(z, -127.0f));
  float r = rt_fma(z, -0.693359375f, x);
  r = rt_fma(z, 2.12194440e-4f, r);
  const float rr = r * r;
  float p = 1.9875691500E-4f;
  p = rt_fma(p, r, 1.3981999507E-3f);
  p = rt_fma(p, r, 8.3334519073E-3f);
  p = rt_fma(p, r, 4.1665795894E-2f);
  p = rt_fma(p, r, 1.6666665459E-1f);
  p = rt_fma(p, r, 5.0000001201E-1f);
  p = rt_fma(p, rr, r);
  p = p + 1.0f;
  // rt_exp scales in two steps, (p * 2^a) * 2^b with a + b = n, so that results below the normal range round once; for
  // x >= -87.34 n >= -126, 2^n is a normal number and p * 2^n is the same single rounding.  Below that the result is 0.
  float e = p * rt_u2f(uint32_t(n + 127) << 23);
  // NaN in => NaN out through the arithmetic (rt_exp returns its argument, i.e. possibly another NaN payload; every caller
  // turns a NaN weight into the same cleared pixel)
  return (x < -87.33654475055310f) ? 0.0f : e;
}

// a / b for a uniform divisor b with y = RN(1 / b) precomputed by an IEEE division: q = RN(a*y), r = a - b*q (exact, fma),
// q' = RN(q + r*y) is the correctly rounded quotient (Markstein 1990, Theorem 7.1: holds when y is the correctly rounded
// reciprocal and no intermediate leaves the normal range).  launchStage enables this path only for 1e-6 <= b <= 1e6; then the
// residual r is a normal number whenever |a/b| >= 2^-25, and below that exp(-a/b) is exactly 1 whatever the last bit of the
// quotient.  Above 1e30 (a * y could overflow for the smallest divisors) and for a non-finite a the plain division runs.
template <bool FAST>
RT_DEV float divUniform(float a, float b, float y)
{
  if(!FAST) return a / b;
  const float q = a * y;
  const float r = __builtin_fmaf(-b, q, a);
  const float q2 = __builtin_fmaf(r, y, q);
  return (a <= 1.0e30f) ? q2 : a / b;   // (kept as a branch: measured 10 % faster filters than a select of `a`)
}

__device__ constexpr float kGauss[5][5] = {{.0030f, .0133f, .0219f, .0133f, .0030f},
                                {.0133f, .0596f, .0983f, .0596f, .0133f},
                                {.0219f, .0983f, .1621f, .0983f, .0219f},
                                {.0133f, .0596f, .0983f, .0596f, .0133f},
                                {.0030f, .0133f, .0219f, .0133f, .0030f}};  // denoise_common.glsl:15-21

// ------------------------------------------------------------------------------------------------------------
// waveletFilter on chip.  The weight of tap q in pixel p's sum is, bit for bit, the weight of tap p in pixel q's sum:
// |a - b| = |b - a|, (a - b)^2 = (b - a)^2 component by component, the material test is symmetric and the 5x5 kernel is
// point symmetric.  On ONE a-trous sub-lattice (the pixels with equal coordinates modulo 2^level) the dilated stencil is a dense 5 x 5,
// so a workgroup that stages a lattice tile and its 2-pixel ring in LDS can evaluate every pair weight once: 12 "forward" offsets per pixel,
// backward taps from the neighbour's forward weights.  (A 256-thread / 16 x 16 tile version and the per-pixel gather were the first two
// forms of this filter; both are superseded by k_denoise_lds below and were removed in round 3 — profiles/r02_denoise_*_ab.txt.)
// ------------------------------------------------------------------------------------------------------------
constexpr int DT_NFWD = 12;
// forward half of the stencil in (i, j), j > 0 or (j == 0 and i > 0)
__device__ constexpr int8_t kFwdI[DT_NFWD] = {1, 2, -2, -1, 0, 1, 2, -2, -1, 0, 1, 2};
__device__ constexpr int8_t kFwdJ[DT_NFWD] = {0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2};
RT_DEV int fwdIndex(int i, int j) { return j == 0 ? i - 1 : (j == 1 ? 4 + i : 9 + i); }   // inverse of the two tables

template <bool IND, bool FAST>
RT_DEV float denoisePairWeight(f3 color, float lum, f3 norm, f3 pos, f3 colorQ, float lumQ, f3 normQ, f3 posQ, float gauss, float sigL, float sigN, float sigD, float yL,
                               float yN, float yD)
{
  const float distColor = IND ? dot(color - colorQ, color - colorQ) : rt_abs(lum - lumQ);
  const float wColor = expNonPositive(-divUniform<FAST>(distColor, sigL, yL)) + 1e-2f;
  const float distNorm2 = dot(norm - normQ, norm - normQ);
  const float wNorm = rt_min(1.0f, expNonPositive(-divUniform<FAST>(distNorm2, sigN, yN)));
  const float distPos2 = dot(pos - posQ, pos - posQ);
  const float wDepth = expNonPositive(-divUniform<FAST>(distPos2, sigD, yD)) + 1e-2f;
  return wColor * wNorm * wDepth * gauss;
}

// k_denoise_lds: a 64-thread workgroup takes an 8 x 8 tile of one a-trous sub-lattice, stages colour (+ luminance), normal + material hash and
// position of the tile and its 2-pixel ring (12 x 12 lattice pixels, 6.9 KB) and every lane takes its 25 taps from there: 7 global loads per lane
// instead of the 75 of a per-pixel gather (which kept the texture-address path 66-82 % busy, profiles/r02_denoise_lds_ab.txt); its one-wave
// workgroups fit into any free wave slot beside the traversal kernels of the frames in flight.
// Every pair weight is evaluated once: a lane owns its pixel's 12 forward weights and the centre weight; a backward
// tap takes the neighbour's forward weight when that neighbour is one of the tile's 64 pixels (71 % of the backward taps), the other 222 (pixel, offset)
// pairs of a tile are the same for every tile — a compile-time list, fetched before the first barrier and worked off in four full-wave passes.  17 weight
// evaluations per lane instead of 25 (-20 % executed VALU instructions); the weights travel through the LDS that held normals and positions, which are
// dead by then.  Same expressions and the accumulation order of the reference's loops (j outer, i inner): bit-identical to a per-pixel gather.
constexpr int DL_T = 8, DL_S = DL_T + 4;
// the (pixel p, forward offset k) pairs whose backward partner q = p - offset(k) lies outside the tile, offsets ascending, pixels ascending within an offset:
// pair = LDS index of p | LDS index of q << 8 in the staged 12 x 12 block, gauss = the kernel factor of the offset, base[k] = first pair of offset k
struct BwdPairs { uint32_t pair[256]; float gauss[256]; int16_t base[DT_NFWD + 1]; };
constexpr float kGaussFwdC[DT_NFWD] = {.0983f, .0219f, .0133f, .0596f, .0983f, .0596f, .0133f, .0030f, .0133f, .0219f, .0133f, .0030f};
constexpr BwdPairs makeBwdPairs()
{
  BwdPairs t{};
  int n = 0;
  for(int k = 0; k < DT_NFWD; k++) {
    const int i = kFwdI[k], j = kFwdJ[k];
    t.base[k] = int16_t(n);
    for(int p = 0; p < 64; p++) {
      const int qx = (p & 7) - i, qy = (p >> 3) - j;
      if(qx < 0 || qx > 7 || qy < 0) {
        const int pidx = ((p >> 3) + 2) * 12 + ((p & 7) + 2), qidx = pidx - j * 12 - i;
        t.pair[n] = uint32_t(pidx | (qidx << 8)); t.gauss[n] = kGaussFwdC[k]; n++;
      }
    }
  }
  t.base[DT_NFWD] = int16_t(n);
  return t;
}
__device__ constexpr BwdPairs kBwdPairs = makeBwdPairs();
constexpr int DL_NBWD = makeBwdPairs().base[DT_NFWD];
static_assert(DL_NBWD == 222, "backward pair list");
// slot of pixel (x, y)'s directly evaluated backward weight for forward offset k = (i, j): its rank in kBwdPairs.  Rows y < j lie outside with all 8 pixels,
// the rows below with |i| pixels each (x < i for i > 0, x > 7 + i for i < 0)
RT_DEV int bwdSlot(int k, int i, int j, int x, int y)
{
  const int ai = i < 0 ? -i : i;
  const int inRow = i > 0 ? x : x - (8 - ai);
  const int r = y < j ? 8 * y + x : 8 * j + (y - j) * ai + inRow;
  return kBwdPairs.base[k] + r;
}

template <bool IND, bool FAST>
__global__ __launch_bounds__(64) void k_denoise_lds(DevFrame F, rt_state st, const float4* src, float4* dst, int level, int rowBegin, int rowEnd, int tilesX, int tilesY,
                                                     float yL, float yN, float yD)
{
  __shared__ float4 sC[DL_S * DL_S];
  __shared__ float4 sNP[2 * DL_S * DL_S];
  float4* sN = sNP; float4* sP = sNP + DL_S * DL_S;
  const int step = 1 << level;
  const int total = tilesX * tilesY * step * step, perXcd = (total + 7) / 8;
  const int local = int(blockIdx.x >> 3);
  const int work = int(blockIdx.x & 7u) * perXcd + local;
  if(local >= perXcd || work >= total) return;
  const int sub = work % (step * step), tileIdx = work / (step * step);
  const int a = sub % step, b = sub / step;
  const int X0 = (tileIdx % tilesX) * DL_T, Y0 = (tileIdx / tilesX) * DL_T;
  const i2 bound = IND ? i2{st.size.x / 2, st.size.y / 2} : i2{st.size.x, st.size.y};
  const float sigLumin = IND ? st.sigLuminIndirect : st.sigLuminDirect;
  const float sigNormal = IND ? st.sigNormalIndirect : st.sigNormalDirect;
  const float sigDepth = IND ? st.sigDepthIndirect : st.sigDepthDirect;
  const int last = IND ? 4 : 3;
  const float4* gN = IND ? F.geomNh : F.geomN;
  const float4* gP = IND ? F.geomPh : F.geomP;
  const int lane = int(threadIdx.x);
#pragma unroll
  for(int it = 0; it < 3; it++) {
    const int idx = lane + 64 * it;
    if(idx < DL_S * DL_S) {
      const int lx = idx % DL_S, ly = idx / DL_S;
      const int px = a + step * (X0 - 2 + lx), py = rowBegin + b + step * (Y0 - 2 + ly);
      float4 c = make_float4(0.f, 0.f, 0.f, 0.f), n = make_float4(0.f, 0.f, 0.f, rt_u2f(RT_INVALID_MAT_ID)), q = make_float4(0.f, 0.f, 0.f, 0.f);
      if(px >= 0 && py >= 0 && px < bound.x && py < bound.y) {
        const size_t gi = size_t(py) * bound.x + px;
        n = gN[gi]; q = gP[gi];
        c = src[size_t(py) * F.W + px];
        c.w = IND ? 0.0f : luminance(mk3(c.x, c.y, c.z));
      }
      sC[idx] = c; sN[idx] = n; sP[idx] = q;
    }
  }
  uint32_t pr[4]; float pg[4];
#pragma unroll
  for(int r = 0; r < 4; r++) { const int idx = min(r * 64 + lane, DL_NBWD - 1); pr[r] = kBwdPairs.pair[idx]; pg[r] = kBwdPairs.gauss[idx]; }
  __syncthreads();
  const int ux = lane & 7, uy = lane >> 3;
  const int cidx = (uy + 2) * DL_S + (ux + 2);
  const float4 cN = sN[cidx], cC = sC[cidx], cP = sP[cidx];
  const uint32_t hash = rt_f2u(cN.w);
  const f3 color = mk3(cC.x, cC.y, cC.z), norm = mk3(cN.x, cN.y, cN.z), pos = mk3(cP.x, cP.y, cP.z);
  float wf[DT_NFWD];
#pragma unroll
  for(int k = 0; k < DT_NFWD; k++) {
    const int qidx = cidx + kFwdJ[k] * DL_S + kFwdI[k];
    const float4 qN = sN[qidx];
    float w = -1.0f;
    if(hash != RT_INVALID_MAT_ID && rt_f2u(qN.w) == hash) {
      const float4 qP = sP[qidx], qC = sC[qidx];
      w = denoisePairWeight<IND, FAST>(color, cC.w, norm, pos, mk3(qC.x, qC.y, qC.z), qC.w, mk3(qN.x, qN.y, qN.z), mk3(qP.x, qP.y, qP.z), kGauss[kFwdI[k] + 2][kFwdJ[k] + 2],
                                       sigLumin, sigNormal, sigDepth, yL, yN, yD);
    }
    wf[k] = w;
  }
  float wb[4];
#pragma unroll
  for(int r = 0; r < 4; r++) {
    float w = -1.0f;
    if(r * 64 + lane < DL_NBWD) {
      const int pidx = int(pr[r] & 0xffu), qidx = int(pr[r] >> 8);
      const float4 pN = sN[pidx], qN = sN[qidx];
      const uint32_t ph = rt_f2u(pN.w);
      if(ph != RT_INVALID_MAT_ID && rt_f2u(qN.w) == ph) {
        const float4 pC = sC[pidx], pP = sP[pidx], qC = sC[qidx], qP = sP[qidx];
        w = denoisePairWeight<IND, FAST>(mk3(pC.x, pC.y, pC.z), pC.w, mk3(pN.x, pN.y, pN.z), mk3(pP.x, pP.y, pP.z), mk3(qC.x, qC.y, qC.z), qC.w, mk3(qN.x, qN.y, qN.z),
                                         mk3(qP.x, qP.y, qP.z), pg[r], sigLumin, sigNormal, sigDepth, yL, yN, yD);
      }
    }
    wb[r] = w;
  }
  __syncthreads();
  float* sW = reinterpret_cast<float*>(sNP);
#pragma unroll
  for(int k = 0; k < DT_NFWD; k++) sW[k * 64 + lane] = wf[k];
#pragma unroll
  for(int r = 0; r < 4; r++) if(r * 64 + lane < DL_NBWD) sW[DT_NFWD * 64 + r * 64 + lane] = wb[r];
  __syncthreads();
  const i2 coord{a + step * (X0 + ux), rowBegin + b + step * (Y0 + uy)};
  if(coord.x >= bound.x || coord.y >= bound.y || coord.y >= rowEnd) return;
  f3 res = mk3(0.0f);
  if(hash != RT_INVALID_MAT_ID) {
    f3 sum = mk3(0.0f);
    float sumWeight = 0.0f;
#pragma unroll
    for(int j = -2; j <= 2; j++)
#pragma unroll
      for(int i = -2; i <= 2; i++) {
        float w;
        if(i == 0 && j == 0) {
          // the pixel with itself: every distance is x - x = 0 for finite inputs (NaN otherwise), every exponential exp(-0) = 1, the weight a constant — evaluated
          // with the general expression's own float operations, in its order.  A wave with a non-finite pixel takes the general path (scalar branch).
          const float dl = IND ? dot(color - color, color - color) : rt_abs(cC.w - cC.w);
#ifndef RT_NO_CENTRE_SHORTCUT
          // Only in the FAST instantiation: there every sigma has been checked to lie in [1e-6, 1e6] (uniformDivOk), so 0 / sigma = 0.  With sigma = 0 or NaN the
          // general expression is 0 / 0 = NaN and the reference clears the pixel — that case must take the general path.
          if(FAST && __ballot((dl == 0.0f && dot(norm - norm, norm - norm) == 0.0f && dot(pos - pos, pos - pos) == 0.0f) ? 0 : 1) == 0ull) w = (((1.0f + 1e-2f) * 1.0f) * (1.0f + 1e-2f)) * kGauss[2][2];
          else
#endif
          w = denoisePairWeight<IND, FAST>(color, cC.w, norm, pos, color, cC.w, norm, pos, kGauss[2][2], sigLumin, sigNormal, sigDepth, yL, yN, yD);
        }
        else if(j > 0 || (j == 0 && i > 0)) w = wf[fwdIndex(i, j)];
        else {
          const int k = fwdIndex(-i, -j), qx = ux + i, qy = uy + j;
          const bool inside = qx >= 0 && qx <= 7 && qy >= 0;
          w = sW[inside ? k * 64 + qy * 8 + qx : DT_NFWD * 64 + bwdSlot(k, -i, -j, ux, uy)];
        }
        if(w != -1.0f) {
          const float4 q = sC[cidx + j * DL_S + i];
          sum += mk3(q.x, q.y, q.z) * w;
          sumWeight += w;
        }
      }
    res = (sumWeight < 1e-5f) ? mk3(0.0f) : sum / sumWeight;
    if(hasNan(res) || res.x < 0 || res.y < 0 || res.z < 0 || res.x > 1e8f || res.y > 1e8f || res.z > 1e8f) res = mk3(0.0f);
  }
  if(level == last) res = LDRToHDR(res);
  storeImg(dst, F, coord, mk4(res, 1.0f));
}

// ------------------------------------------------------------------------------------------------------------
// compose.comp:23-43
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_compose(DevFrame F, rt_state st, int rowBegin, int rowEnd, int tilesX, int tilesY)
{
  const TileCoord tile = tileOf(tilesX, tilesY);
  if(!tile.valid) return;
  const int lane = int(threadIdx.x);
  const i2 coord{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + (lane >> 3)};
  if(coord.x >= st.size.x || coord.y >= st.size.y || coord.y >= rowEnd) return;
  const float4* indSrc = (st.denoise > 0) ? F.denoiseIndB : F.denoiseIndA;
  const i2 half{coord.x / 2, coord.y / 2};
  if(st.modulate == 0) {
    storeImg(F.thisIndirectResult, F, coord, loadImg(indSrc, F, half));
  } else {
    const f3 albedo = xyz(unpackUnorm4x8(loadG(F.thisG, F, coord).w));
    const f3 direct = xyz(loadImg(F.thisDirectResult, F, coord)) * albedo;
    const f3 indirect = xyz(loadImg(indSrc, F, half)) * albedo;
    storeImg(F.thisDirectResult, F, coord, mk4(direct, 1.0f));
    storeImg(F.thisIndirectResult, F, coord, mk4(indirect, 1.0f));
  }
}

#endif  // !RT_LAT

// ------------------------------------------------------------------------------------------------------------
// host-side launch (one entry of Renderer::run's dispatch list, renderer.cpp:163-205)
// ------------------------------------------------------------------------------------------------------------
// divUniform's fast path: the divisor must keep every intermediate in the normal range (see there)
static bool uniformDivOk(float s) { return s >= 1e-6f && s <= 1e6f; }

#if !RT_LAT
// one level of an a-trous chain on the one-wave LDS-staged kernel
template <bool IND>
static void launchDenoiseLevel(hipStream_t stream, const DevFrame& F, const rt_state& st, const float4* src, float4* dst, int level, int rowBegin, int rowEnd, int gw)
{
  const float sL = IND ? st.sigLuminIndirect : st.sigLuminDirect, sN = IND ? st.sigNormalIndirect : st.sigNormalDirect, sD = IND ? st.sigDepthIndirect : st.sigDepthDirect;
  const int stp = 1 << level, ltx = ((gw + stp - 1) / stp + DL_T - 1) / DL_T, lty = ((rowEnd - rowBegin + stp - 1) / stp + DL_T - 1) / DL_T;
  const unsigned nwg = 8u * unsigned((ltx * lty * stp * stp + 7) / 8);
  if(uniformDivOk(sL) && uniformDivOk(sN) && uniformDivOk(sD))
    hipLaunchKernelGGL((k_denoise_lds<IND, true>), dim3(nwg), dim3(64), 0, stream, F, st, src, dst, level, rowBegin, rowEnd, ltx, lty, 1.0f / sL, 1.0f / sN, 1.0f / sD);
  else
    hipLaunchKernelGGL((k_denoise_lds<IND, false>), dim3(nwg), dim3(64), 0, stream, F, st, src, dst, level, rowBegin, rowEnd, ltx, lty, 0.f, 0.f, 0.f);
}
#endif

hipError_t launchStage(hipStream_t stream, const DevScene& Sin, const DevFrame& F, const rt_state& st, const rt_scene_camera& cam, int stage, int level,
                       int rowBegin, int rowEnd)
{
#if RT_LAT
  // only two kernels exist in this build
  if(stage != RT_STAGE_DIRECT && stage != RT_STAGE_INDIRECT) return rt::RT_FORWARD::launchStage(stream, Sin, F, st, cam, stage, level, rowBegin, rowEnd);
#endif
  DevScene S = Sin;
#if RT_LAT
  S.stackEntries = S.stackTotal + (S.gangMax > 0 ? GANG_EXTRA : 0);   // one column per RAY (8 per wave): the whole stack fits in LDS (+ the room gang mode expands into)
  const int nwEnv = 8;   // waves per workgroup of the latency kernels (3 / 4 / 6 measured: slower, profiles/r03_band_chunk_ab.txt)
  const int nWaves = nwEnv;        // waves per tile workgroup
#else
  S.stackEntries = F.stackLds > 0 ? std::min(F.stackLds, S.stackTotal) : S.stackTotal;   // LDS part of the traversal stack for this launch
#endif
  if(stage == RT_STAGE_INDIRECT) S.stackOvf = Sin.stackOvfInd;   // a direct-kind kernel of the next frame can be in flight beside it
  const bool needOvf = S.stackTotal > S.stackEntries;
  const bool half = (stage == RT_STAGE_INDIRECT || stage == RT_STAGE_DENOISE_INDIRECT);
  const int gw = half ? st.size.x / 2 : st.size.x, gh = half ? st.size.y / 2 : st.size.y;
  if(rowEnd <= 0 || rowEnd > gh) rowEnd = gh;
  if(rowBegin < 0) rowBegin = 0;
  if(rowBegin >= rowEnd || gw <= 0) return hipSuccess;
  const int tilesX = (gw + 7) / 8, tilesY = (rowEnd - rowBegin + 7) / 8;
  const dim3 grid(tileGrid(tilesX, tilesY)), block(64);
  const size_t lds = size_t(S.stackEntries) * 64 * sizeof(uint2);
  const bool spatial = st.ReSTIRState == RT_RESTIR_SPATIAL || st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL;
  switch(stage) {
    case RT_STAGE_DIRECT:
      // (a K-tiles-per-wave variant of this kernel — primary and shadow rays through the LDS ray pool, pixel state in the scratch
      //  records between the traces — was measured: 1.77 -> 2.6 ms; the coherent primary rays gain nothing from the pool and the
      //  split shading costs more than the shorter tail saves.  The single-bounce indirect tiles are where pooling pays.)
      // `level` selects the halves of the stage for hosts that must exchange the cached reservoirs of neighbouring rows in between
      // (row-tiled multi-GPU frames with spatial reuse): 0 = the whole stage, 1 = k_direct_stage only, 2 = k_direct_spatial only
      if(level < 0 || level > 2 || (level != 0 && !spatial)) return hipErrorInvalidValue;
#if RT_LAT
      if(level != 2) hipLaunchKernelGGL(k_direct_stage, grid, dim3(64 * nWaves), wideLdsBytes(S.stackEntries, nWaves), stream, S, F, st, cam, rowBegin, rowEnd, tilesX, tilesY);
      if(level != 1 && spatial) return rt::RT_FORWARD::launchStage(stream, Sin, F, st, cam, stage, 2, rowBegin, rowEnd);
#else
      if(needOvf && grid.x * 64u > S.stackOvfThreads) return hipErrorInvalidConfiguration;   // overflow area missing / too small: an internal sizing error, not the caller's
      if(level != 2) {
        // full-frame launches take their tile rows in the order of what they cost last time (DevFrame::rowCost); row bands and chunked launches keep the screen order
        DevFrame Fd = F;
        const bool ordered = F.rowOrder && rowBegin == 0 && rowEnd == gh && tileChunk(tilesX, tilesY) == tilesX && tilesY <= 4096;
        if(ordered) hipLaunchKernelGGL(k_row_order, dim3(1), dim3(256), size_t(tilesY) * sizeof(uint32_t), stream, F.rowCost, F.rowOrder, tilesY);
        else { Fd.rowCost = nullptr; Fd.rowOrder = nullptr; }
        hipLaunchKernelGGL(k_direct_stage, grid, block, lds, stream, S, Fd, st, cam, rowBegin, rowEnd, tilesX, tilesY);
      }
      if(level != 1 && spatial) hipLaunchKernelGGL(k_direct_spatial, grid, block, 0, stream, S, F, st, cam, rowBegin, rowEnd, tilesX, tilesY);
#endif
      break;
#if !RT_LAT
    case RT_STAGE_DIRECT_GEN:
      if(needOvf && grid.x * 64u > S.stackOvfThreads) return hipErrorInvalidConfiguration;   // overflow area missing / too small: an internal sizing error, not the caller's
      hipLaunchKernelGGL(k_direct_gen, grid, block, lds, stream, S, F, st, cam, rowBegin, rowEnd, tilesX, tilesY);
      break;
#endif
    case RT_STAGE_INDIRECT: {
      // per-XCD tile lists: capacity = the tiles one XCD can own under the striped mapping
      const int cap = int(tileGrid(tilesX, tilesY) / 8);
      hipLaunchKernelGGL(k_ind_tile_order, dim3(8), dim3(256), 0, stream, st, rowBegin, tilesX, tilesY, cap, F.tileOrder, F.qcount + 192);
#if RT_LAT
      // one workgroup per tile, multi-bounce tiles first (same lists)
      hipLaunchKernelGGL(k_indirect_stage, grid, dim3(64 * nWaves), wideLdsBytes(S.stackEntries, nWaves), stream, S, F, st, cam, rowBegin, rowEnd, tilesX, tilesY, cap,
                         (const uint32_t*)F.tileOrder, F.qcount + 192, 0, 0, int(grid.x), 0);
#else
      // under ~2 waves per SIMD (1024 SIMDs) the launch is latency bound: split tiles over more waves
      static const int subEnv = getenv("RESTIR_IND_SUB") ? atoi(getenv("RESTIR_IND_SUB")) : -1;
      const int nTiles = tilesX * tilesY;
      const int subShift = subEnv >= 0 ? subEnv : (nTiles <= 1536 ? 2 : (nTiles <= 3072 ? 1 : 0));
      // throughput-bound launches: single-bounce tiles go K per wave (indirectSingleBounceTiles; 3 per wave: fewer wave instructions than 2, frames
      // in flight -1.3 %, the stage alone +4 %); latency-bound ones keep one (part of a) tile per wave
      const int sbK = subShift > 0 ? 0 : 3;
      const unsigned genericBlocks = grid.x << subShift;
      const unsigned sbBlocks = sbK > 0 ? 8u * unsigned((cap + sbK - 1) / sbK) : 0u;
      const size_t poolBytes = std::max<size_t>(POOL_BYTES, size_t(sbK) * 64 * 33);
      if(needOvf && (genericBlocks + sbBlocks) * 64u > S.stackOvfThreads) return hipErrorInvalidConfiguration;
      // multi-bounce tiles of large launches: persistent waves with per-lane path regeneration, RESTIR_IND_PERSIST tiles' worth of pixels per wave (0 = one wave per tile,
      // the generic body; profiles/r06_indirect_persistent_ab.txt)
      const char* pe = getenv("RESTIR_IND_PERSIST");   // (read per launch: the parity tests switch it inside one process)
      const int persistEnv = pe ? std::max(0, std::min(16, atoi(pe))) : RT_IND_PERSIST_DEFAULT;
      const int persist = sbK > 0 ? persistEnv : 0;
      hipLaunchKernelGGL(k_indirect_stage, dim3(genericBlocks + sbBlocks), block, lds + poolBytes, stream, S, F, st, cam, rowBegin, rowEnd, tilesX, tilesY, cap,
                         (const uint32_t*)F.tileOrder, F.qcount + 192, subShift, sbK, int(genericBlocks), persist);
#endif
      break;
    }
#if !RT_LAT
    case RT_STAGE_DIRECT_REUSE: hipLaunchKernelGGL(k_direct_reuse, grid, block, 0, stream, S, F, st, cam, rowBegin, rowEnd, tilesX, tilesY); break;
    case RT_STAGE_DENOISE_DIRECT: {
      // DirectResult -> DirA -> DirB -> DirA -> DirectResult (denoise_direct.comp:152-172)
      const float4* src[4] = {F.thisDirectResult, F.denoiseDirA, F.denoiseDirB, F.denoiseDirA};
      float4* dst[4] = {F.denoiseDirA, F.denoiseDirB, F.denoiseDirA, F.thisDirectResult};
      if(level < 0 || level > 3) return hipErrorInvalidValue;
      if(level == 0) {  // decode the band's geometry plus the halo rows the widest level (2*2^3) will tap
        const int g0 = std::max(0, rowBegin - 16) & ~7, g1 = std::min(gh, rowEnd + 16);
        const int gty = (g1 - g0 + 7) / 8;
        hipLaunchKernelGGL(k_denoise_geom<false>, dim3(tileGrid(tilesX, gty)), block, 0, stream, F, st, cam, g0, g1, tilesX, gty);
      }
      launchDenoiseLevel<false>(stream, F, st, src[level], dst[level], level, rowBegin, rowEnd, gw);
      break;
    }
    case RT_STAGE_DENOISE_INDIRECT: {
      // IndA -> IndB -> IndA -> thisIndirectResult (scratch) -> IndA -> IndB (denoise_indirect.comp:146-171)
      if(st.denoise == 0) return hipSuccess;
      const float4* src[5] = {F.denoiseIndA, F.denoiseIndB, F.denoiseIndA, F.thisIndirectResult, F.denoiseIndA};
      float4* dst[5] = {F.denoiseIndB, F.denoiseIndA, F.thisIndirectResult, F.denoiseIndA, F.denoiseIndB};
      if(level < 0 || level > 4) return hipErrorInvalidValue;
      if(level == 0) {  // halo = 2*2^4 half-res rows
        const int g0 = std::max(0, rowBegin - 32) & ~7, g1 = std::min(gh, rowEnd + 32);
        const int gty = (g1 - g0 + 7) / 8;
        hipLaunchKernelGGL(k_denoise_geom<true>, dim3(tileGrid(tilesX, gty)), block, 0, stream, F, st, cam, g0, g1, tilesX, gty);
      }
      launchDenoiseLevel<true>(stream, F, st, src[level], dst[level], level, rowBegin, rowEnd, gw);
      break;
    }
    case RT_STAGE_COMPOSE: hipLaunchKernelGGL(k_compose, grid, block, 0, stream, F, st, rowBegin, rowEnd, tilesX, tilesY); break;
#endif
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace RT_VARIANT
}  // namespace rt
