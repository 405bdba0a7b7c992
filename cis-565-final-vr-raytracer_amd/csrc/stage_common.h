// stage_common.h — helpers of the stage kernels (stages.hip).
#pragma once
#include "shading.h"
#include "stages.h"

namespace rt {

// ---- tile / lane bookkeeping ------------------------------------------------------------------------------------
struct TileCoord { int x, y; bool valid; };
// XCD-aware tile order.  Workgroup L runs on XCD L % 8 (observed placement; used for speed only).  The screen is cut into
// stripes of TILE_STRIPE tile rows; stripe s belongs to XCD s % 8, so every XCD's private L2 works on a few compact screen
// regions (and the BVH subtrees under them) while expensive image regions (foliage rows vs sky rows) are spread over all XCDs.
#ifndef RT_TILE_STRIPE
#define RT_TILE_STRIPE 1
#endif
#ifndef RT_TILE_CHUNK
#define RT_TILE_CHUNK 0      // > 0: chunks of this many tiles (row-major tile order) instead of stripes of whole tile rows
#endif
constexpr int TILE_STRIPE = RT_TILE_STRIPE;
// tiles per chunk; chunk c (row-major tile order) belongs to XCD c % 8.  Measured on the 1080p bench scene (frames in flight, ms/frame):
// stripes of 8 / 4 / 2 / 1 tile rows: 3.70 / 3.39 / 3.23 / 3.15 — the finer the interleave, the better the XCDs are balanced
// (sky rows vs street rows); profiles/r02_tile_order_ab.txt has the chunked variants.
// SMALL launches (row bands of a multi-GPU frame: fewer than RT_TILE_SMALL_ROWS tile rows): whole tile rows per XCD would leave XCDs without work — a
// 32-row band has 4 tile rows for 8 XCDs — so they are dealt in chunks of RT_TILE_SMALL_CHUNK tiles (round 3, profiles/r03_band_chunk_ab.txt).
#ifndef RT_TILE_SMALL_ROWS
#define RT_TILE_SMALL_ROWS 24
#endif
#ifndef RT_TILE_SMALL_CHUNK
#define RT_TILE_SMALL_CHUNK 8
#endif
__host__ __device__ inline int tileChunk(int tilesX, int tilesY)
{
  if(RT_TILE_CHUNK > 0) return RT_TILE_CHUNK;
  return tilesY < RT_TILE_SMALL_ROWS ? RT_TILE_SMALL_CHUNK : TILE_STRIPE * tilesX;
}
RT_DEV TileCoord tileOfBlock(int L, int tilesX, int tilesY)
{
  const int xcd = L & 7, k = L >> 3;                 // k-th workgroup of this XCD
  const int G = tileChunk(tilesX, tilesY);
  const int j = k / G, off = k - j * G;
  const int t = (j * 8 + xcd) * G + off;             // row-major tile index
  TileCoord tc;
  tc.y = t / tilesX;
  tc.x = t - tc.y * tilesX;
  tc.valid = t < tilesX * tilesY;
  return tc;
}
RT_DEV TileCoord tileOf(int tilesX, int tilesY) { return tileOfBlock(int(blockIdx.x), tilesX, tilesY); }
// the same mapping with the tile rows taken in the order of `rowOrder` (whole tile rows per XCD: launches with tileChunk == tilesX)
RT_DEV TileCoord tileOfOrdered(const uint16_t* rowOrder, int tilesX, int tilesY)
{
  if(!rowOrder) return tileOf(tilesX, tilesY);
  const int L = int(blockIdx.x), xcd = L & 7, k = L >> 3;
  const int j = k / tilesX, off = k - j * tilesX, s = j * 8 + xcd;
  TileCoord tc;
  tc.valid = s < tilesY;
  tc.y = tc.valid ? int(rowOrder[s]) : 0;
  tc.x = off;
  return tc;
}
// grid size (in workgroups) that covers tilesX x tilesY tiles with the mapping above
inline unsigned tileGrid(int tilesX, int tilesY)
{
  const int G = tileChunk(tilesX, tilesY);
  const int chunks = (tilesX * tilesY + G - 1) / G;
  const int perXcd = (chunks + 7) / 8;
  return unsigned(8 * perXcd * G);
}
// (A column mapping for row bands — XCD x owns a vertical stripe of the band, so that its L2 holds an eighth of the scene under the band — was measured
// in round 3: the trees of the benchmark street then all belong to two XCDs and the band takes 1.1-1.9x longer, profiles/r03_lat_wide_ab.txt.)

#ifndef RT_COUNT
#define RT_COUNT 1
#endif
RT_DEV void flushCounters(const DevFrame& F, const Ctx& c)
{
#if RT_COUNT
  if(!F.counters) return;
  atomicAdd(&F.counters[0], (unsigned long long)c.nClosest);
  atomicAdd(&F.counters[1], (unsigned long long)c.nAny);
  atomicAdd(&F.counters[2], (unsigned long long)c.tc.nodes);
  atomicAdd(&F.counters[3], (unsigned long long)c.tc.tris);
  atomicAdd(&F.counters[4], (unsigned long long)c.nShaded);
  atomicAdd(&F.counters[5], (unsigned long long)c.nRis);
  atomicAdd(&F.counters[6], (unsigned long long)c.tc.rounds);
  atomicAdd(&F.counters[7], (unsigned long long)c.tc.live);
#else
  (void)F; (void)c;
#endif
}

#if RT_WAVEPROF
// per-wave profile record (measurement builds): 16 x u32 per workgroup in DevFrame::waveProf, merged with atomicMax / atomicAdd.
//  0 tile x | 1 tile y | 2 wave cycles | 3 cycles in closest-hit traces | 4 in any-hit traces | 5-7 rounds node / tri / coop | 8-10 their cycles |
//  11 max nodes per lane | 12 max tris per lane | 13 alpha candidates resolved by the texture (sum) | 14 by the micro-map (sum) | 15 100 MHz ticks
RT_DEV void waveProfFlush(const DevFrame& F, const Ctx& c, int tx, int ty, uint64_t c0, uint64_t w0)
{
  uint32_t* rec = F.waveProf + size_t(blockIdx.x) * 16;
  rec[0] = uint32_t(tx); rec[1] = uint32_t(ty);
  atomicMax(&rec[2], uint32_t(clock64() - c0)); atomicMax(&rec[3], c.cycClosest); atomicMax(&rec[4], c.cycAny);
  atomicMax(&rec[5], c.tc.rN); atomicMax(&rec[6], c.tc.rT); atomicMax(&rec[7], c.tc.rC);
  atomicMax(&rec[8], c.tc.cN); atomicMax(&rec[9], c.tc.cT); atomicMax(&rec[10], c.tc.cC);
  atomicMax(&rec[11], c.tc.nodes); atomicMax(&rec[12], c.tc.tris);
  atomicAdd(&rec[13], c.tc.aTex); atomicAdd(&rec[14], c.tc.aOmm);
  atomicMax(&rec[15], uint32_t(wall_clock64() - w0));
  // launch-wide histograms of the traversal rounds (record 65534): [0..7] rounds by live lanes, [8..15] rounds by lanes that execute the round (lane 0's counters:
  // the counters are wave-uniform)
  if((threadIdx.x & 63) == 0) {
    uint32_t* h = F.waveProf + size_t(65534) * 16;
    for(int k = 0; k < 8; k++) { atomicAdd(&h[k], c.tc.hl[k]); atomicAdd(&h[8 + k], c.tc.he[k]); }
  }
}
#endif

// ---- image helpers: Vulkan storage-image semantics (out-of-bounds loads return 0) ------------------------------
RT_DEV uint4 loadG(const uint4* g, const DevFrame& F, i2 c)
{
  if(c.x < 0 || c.y < 0 || c.x >= F.W || c.y >= F.H) return make_uint4(0, 0, 0, 0);
  return g[size_t(c.y) * F.W + c.x];
}
RT_DEV f4 loadImg(const float4* img, const DevFrame& F, i2 c)
{
  if(c.x < 0 || c.y < 0 || c.x >= F.W || c.y >= F.H) return mk4(0, 0, 0, 0);
  const float4 v = img[size_t(c.y) * F.W + c.x];
  return mk4(v.x, v.y, v.z, v.w);
}
RT_DEV void storeImg(float4* img, const DevFrame& F, i2 c, f4 v) { img[size_t(c.y) * F.W + c.x] = make_float4(v.x, v.y, v.z, v.w); }
RT_DEV short sat16(int v) { return short(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
RT_DEV void storeMotion(const DevFrame& F, i2 c, i2 v) { F.motion[size_t(c.y) * F.W + c.x] = make_short2(sat16(v.x), sat16(v.y)); }  // RG16_SINT
RT_DEV i2 loadMotion(const DevFrame& F, i2 c)
{
  if(c.x < 0 || c.y < 0 || c.x >= F.W || c.y >= F.H) return i2{0, 0};
  const short2 m = F.motion[size_t(c.y) * F.W + c.x];
  return i2{int(m.x), int(m.y)};
}

// ---- shared by the direct stages ---------------------------------------------------------------------------------
RT_DEV uint4 encodeGeometryInfo(const State& state, float depth)  // direct_stage.comp:37-45
{
  uint4 g;
  g.x = rt_f2u(depth);
  g.y = compress_unit_vec(state.normal);
  g.z = packUnorm4x8(mk4(state.mat.metallic, state.mat.roughness, (state.mat.ior - 1.0f) / RT_MAX_IOR_MINUS_ONE, state.mat.transmission));
  g.w = packUnorm4x8(mk4(state.mat.albedo, 1.0f)) & 0xFFFFFFu;
  g.w += hash8bit(state.matID);
  return g;
}
RT_DEV void updateGeometryAlbedo(uint4& g, f3 albedo)  // direct_gen.comp:63-66
{
  uint32_t matId = g.w & 0xff000000u;
  g.w = (packUnorm4x8(mk4(albedo, 1.0f)) & 0x00ffffffu) | matId;
}
RT_DEV i2 createMotionIndex(const Ctx& c, f3 wpos)  // direct_stage.comp:125-139
{
  f4 proj = mul(c.cam.lastProjView, mk4(wpos, 1.0f));
  f3 ndc = xyz(proj) / proj.w;
  f2 mv = mk2(ndc.x, ndc.y) * 0.5f + 0.5f;
  f2 s = mv * mk2(float(c.rtx.size.x), float(c.rtx.size.y));
  return i2{rt_ftoi(s.x), rt_ftoi(s.y)};
}
// direct_stage.comp:47-84 == direct_reuse.comp:52-89
RT_DEV bool findTemporalNeighborDirect(const DevFrame& F, const rt_state& st, f3 norm, float reprojDepth, uint32_t matId, i2 lastCoord,
                                       rt_direct_reservoir& resv, uint32_t& lid)
{
  const i2 size{st.size.x, st.size.y};
  if(!inBound(lastCoord, i2{2, 0}, size)) return false;
  if(lastCoord.y < F.histRow0 || lastCoord.y >= F.histRow1) *F.histMiss = 1u;
  const uint4 g = loadG(F.lastG, F, lastCoord);
  const f3 pnorm = decompress_unit_vec(g.y);
  const float pdepth = rt_u2f(g.x);
  const uint32_t matHash = g.w & 0xFF000000u;
  if(inBound(lastCoord, size)) {
    if(hash8bit(matId) == matHash) {
      if(dot(norm, pnorm) > 0.9f && reprojDepth < pdepth * 1.05f) {
        const size_t li = size_t(lastCoord.y) * st.size.x + lastCoord.x;
        resv = F.lastDirectResv[li];
        lid = F.lastLightId[li];
        return true;
      }
    }
  }
  return false;
}
// resvUpdate / resvMerge / resvClamp / resvCheckValidity on the AoS reservoir (reservoir.glsl:34-82, 116-128)
RT_DEV bool resvUpdate(rt_direct_reservoir& r, const rt_light_sample& s, float w, float rr)
{
  r.weight += w; r.num += 1;
  if(rr * r.weight < w) { r.lightSample = s; return true; }
  return false;
}
RT_DEV bool resvMerge(rt_direct_reservoir& r, const rt_direct_reservoir& rhs, float rr)
{
  r.weight += rhs.weight; r.num += rhs.num;
  if(rr * r.weight < rhs.weight) { r.lightSample = rhs.lightSample; return true; }
  return false;
}
template <class R> RT_DEV void resvClamp(R& r, int clamp)
{
  if(r.num > uint32_t(clamp)) { r.weight *= float(clamp) / float(r.num); r.num = uint32_t(clamp); }
}
RT_DEV rt_direct_reservoir zeroDirectResv()
{
  rt_direct_reservoir r;
  r.lightSample.Li = rt_vec3{0, 0, 0}; r.lightSample.wi = rt_vec3{0, 0, 0}; r.lightSample.dist = 0.f; r.num = 0; r.weight = 0.f;
  return r;
}

// the M-candidate RIS loop (direct_stage.comp:189-200 == direct_gen.comp:110-122); the visibility of the winner (:202-210) is the caller's:
// returns whether the shadow ray has to be traced (a zero-weight reservoir cannot change — the only effect of the ray is weight := 0 — so its ray is skipped)
RT_DEV bool risCandidatesNoVisibility(Ctx& c, const State& state, f3 wo, rt_direct_reservoir& resv, uint32_t& lid, Ray& shadowRay, float& shadowDist)
{
  for(int i = 0; i < c.rtx.RISSampleNum; i++) {
    rt_light_sample ls;
    float p = c.SampleDirectLightNoVisibility(state.position, ls);
    f3 pHat = mk3(ls.Li) * metallicWorkflowBSDF(state.mat, state.ffnormal, wo, mk3(ls.wi)) * rt_abs(dot(state.ffnormal, mk3(ls.wi)));
    float weight = resvToScalar(pHat / p);
    if(Ctx::IsPdfInvalid(p) || rt_isnan(weight)) weight = 0.0f;
    if(resvUpdate(resv, ls, weight, rnd(c.seed))) lid = c.lastLightId;
  }
  const rt_light_sample ls = resv.lightSample;
  shadowRay = Ray{OffsetRay(state.position, state.ffnormal), mk3(ls.wi)};
  shadowDist = ((ls.dist - rt_abs(shadowRay.origin.x - state.position.x)) - rt_abs(shadowRay.origin.y - state.position.y)) - rt_abs(shadowRay.origin.z - state.position.z);  // Occlusion, pathtrace.glsl:18-22
  return resv.weight != 0.0f;
}
RT_DEV void risCandidates(Ctx& c, const State& state, f3 wo, rt_direct_reservoir& resv, uint32_t& lid)
{
  Ray shadowRay; float shadowDist;
  if(risCandidatesNoVisibility(c, state, wo, resv, lid, shadowRay, shadowDist) && c.AnyHit(shadowRay, shadowDist)) resv.weight = 0.0f;
}


struct GState { f3 position, normal, ffnormal; Material mat; uint32_t matID; };
// getDirectStateFromGBuffer / getIndirectStateFromGBuffer, pathtrace.glsl:277-313
RT_DEV bool stateFromGBuffer(uint4 g, const Ray& ray, GState& s, float& depth)
{
  depth = rt_u2f(g.x);
  if(depth >= RT_INFINITY * 0.8f) return false;
  s.position = ray.origin + ray.direction * depth;
  s.normal = decompress_unit_vec(g.y);
  s.ffnormal = dot(s.normal, ray.direction) <= 0.0f ? s.normal : -s.normal;
  s.mat.albedo = xyz(unpackUnorm4x8(g.w));
  s.mat.emission = mk3(0.f);
  const f4 matInfo = unpackUnorm4x8(g.z);
  s.mat.metallic = matInfo.x;
  s.mat.roughness = matInfo.y;
  s.mat.ior = matInfo.z * RT_MAX_IOR_MINUS_ONE + 1.f;
  s.mat.transmission = matInfo.w;
  s.matID = g.w >> 24;
  return true;
}


RT_DEV rt_gi_sample newGISample()  // :110-115
{
  rt_gi_sample s;
  s.L = rt_vec3{0, 0, 0}; s.xv = rt_vec3{0, 0, 0}; s.nv = rt_vec3{100.0f, 100.0f, 100.0f}; s.xs = rt_vec3{0, 0, 0}; s.ns = rt_vec3{0, 0, 0}; s.pHat = 0.f;
  return s;
}
RT_DEV bool GISampleValid(const rt_gi_sample& s) { return s.nv.x < 1.1f && !hasNan(mk3(s.L)); }  // :117-119
RT_DEV float MISw(const rt_state& st, float f, float g) { return (st.MIS > 0) ? powerHeuristic(f, g) : 1.0f; }  // :57-59


}  // namespace rt
