// wavefront.hip — the direct and indirect stages as a wavefront pipeline (the default path).
//
// The fused kernels of stages.hip keep a whole 8x8 tile's lanes together from the primary ray to the last bounce:
// measured on MI355X, 65 % (direct) and 89 % (indirect) of the VALU lanes idle because rays of one wave finish their
// traversal / their path at different times, and 150-210 VGPRs of shading state cap occupancy at 2-3 waves per SIMD while
// the traversal loop waits on HBM/L2.  Here each stage is cut at its ray queries:
//
//   direct   :  k_primary (trace)  ->  k_direct_shade (G-buffer, RIS, enqueue shadow ray)  ->  k_trace<any>  ->  k_direct_resolve
//   indirect :  k_ind_init -> { k_trace<closest> + k_trace<any> -> k_ind_bounce } x maxDepth -> k_ind_finish
//
// Trace kernels carry only traversal state (high occupancy, nothing else in registers); shading kernels have no
// traversal loop (all lanes shade at once); rays that survive a bounce are compacted into dense queues with one
// wave-aggregated atomic per wave (ballot + mbcnt).  Per-pixel arithmetic and RNG draw order are exactly those of the
// fused stages, so every output stays bit-identical (tests/test_gpu_parity.py runs both pipelines against the oracle).
#include "stage_common.h"
#include "direct_phases.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

// This file is compiled twice: as is (HDR environment only) and through *_sky.hip with RT_SKY = 1 (procedural sun & sky code
// paths compiled in).  Keeping sun_and_sky() out of the default kernels saves 13 VGPRs in k_direct_stage (one wave per SIMD of
// occupancy) — the procedural sky is the rarely used mode (default in_use = 0, sample_example.hpp:202).
#ifndef RT_SKY
#define RT_SKY 0
#endif
#if RT_SKY
#define RT_VARIANT sky
#else
#define RT_VARIANT base
#endif
namespace rt {
namespace RT_VARIANT {

// ---- queue append: one atomic per wave ------------------------------------------------------------------------------
RT_DEV uint32_t queueSlot(uint32_t* counter, bool want)
{
  const unsigned long long mask = __ballot(want ? 1 : 0);
  if(!want) return 0u;
  const uint32_t n = uint32_t(__popcll(mask));
  const uint32_t lane = uint32_t(threadIdx.x) & 63u;
  const uint32_t rank = uint32_t(__popcll(mask & ((1ull << lane) - 1ull)));
  uint32_t base = 0;
  if(rank == 0) base = atomicAdd(counter, n);
  base = uint32_t(__shfl(int(base), __ffsll((long long)mask) - 1));
  return base + rank;
}

RT_DEV Ctx makeCtx(const DevScene& S, const rt_state& st, const rt_scene_camera& cam, uint2* stack) { return Ctx(S, st, cam, stack); }

constexpr int CNT_SHADOW = 0;
RT_DEV int cntC(int depth) { return 1 + depth; }   // closest-ray queue consumed at `depth`
RT_DEV int cntA(int depth) { return 32 + depth; }  // shadow-ray queue whose result is consumed at `depth`

// ================================================================================================================
// generic trace kernels
// ================================================================================================================
// rayO = (origin.xyz, tmax), rayD = (dir.xyz, seed bits).  queue == nullptr: dense (index = thread id).
template <bool ANY>
__global__ __launch_bounds__(64) void k_trace(DevScene S, DevFrame F, const uint32_t* queue, const uint32_t* count, const float4* rayO, const float4* rayD,
                                              float4* hitOut, uint32_t* occOut)
{
  extern __shared__ uint2 s_stack[];
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if(i >= *count) return;
  const uint32_t px = queue ? queue[i] : i;
  const float4 o = rayO[px], d = rayD[px];
  RayHit hit; TravCounters tc{0, 0};
  const bool found = traceRay<ANY>(S, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), ANY ? o.w : RT_INFINITY, rt_f2u(d.w), s_stack + threadIdx.x, hit, tc);
  if(ANY) occOut[px] = found ? 1u : 0u;
  else hitOut[px] = make_float4(hit.t, rt_u2f(hit.gid), hit.u, hit.v);
  if(F.counters) {
    atomicAdd(&F.counters[ANY ? 1 : 0], 1ull);
    atomicAdd(&F.counters[2], (unsigned long long)tc.nodes);
    atomicAdd(&F.counters[3], (unsigned long long)tc.tris);
    atomicAdd(&F.counters[6], (unsigned long long)tc.rounds);
    atomicAdd(&F.counters[7], (unsigned long long)tc.live);
  }
}

// ================================================================================================================
// persistent trace kernel: a fixed number of waves pulls rays from the queue; a lane whose ray terminates is refilled
// with the next ray instead of idling until the slowest ray of its wave is done (wave lifetimes of the one-ray-per-lane
// kernels above spread over 100x on foliage: p50 112 us, max 2.7 ms for the primary rays of the benchmark frame).
// ================================================================================================================
constexpr int REFILL_MIN = 20;  // refill when at least this many lanes are idle (amortises the queue atomic)

template <bool ANY, bool PRIMARY>
__global__ __launch_bounds__(64) void k_trace_p(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, const uint32_t* queue, const uint32_t* countPtr,
                                                uint32_t fixedCount, uint32_t* head, const float4* rayO, const float4* rayD, float4* hitOut, uint32_t* occOut,
                                                int rowBegin, int rowEnd, int tilesX)
{
  extern __shared__ uint2 s_stack[];
  const uint32_t lane = threadIdx.x;
  uint2* stack = s_stack + lane;
  const uint32_t count = PRIMARY ? fixedCount : *countPtr;
  Trav T;
  T.found = false; T.hit.t = 0.f; T.hit.gid = 0xffffffffu; T.hit.u = T.hit.v = 0.f;
  bool live = false;
  uint32_t outIdx = 0;
  TravCounters tc{0, 0};
  uint32_t nRays = 0;
  bool exhausted = false;
  for(;;) {
    const unsigned long long idleMask = __ballot(live ? 0 : 1);
    const int nIdle = __popcll(idleMask);
    if(!exhausted && (nIdle >= REFILL_MIN)) {
      uint32_t base = 0;
      const int leader = __ffsll((long long)idleMask) - 1;
      if(int(lane) == leader) base = atomicAdd(head, uint32_t(nIdle));
      base = uint32_t(__shfl(int(base), leader));
      if(base + uint32_t(nIdle) >= count) exhausted = true;
      if(!live) {
        const uint32_t i = base + uint32_t(__popcll(idleMask & ((1ull << lane) - 1ull)));
        if(i < count) {
          f3 o, d; float tmax = RT_INFINITY; uint32_t seed; bool valid = true;
          if(PRIMARY) {
            const uint32_t tile = i >> 6, l = i & 63u;
            const i2 px{int(tile % uint32_t(tilesX)) * 8 + int(l & 7u), rowBegin + int(tile / uint32_t(tilesX)) * 8 + int(l >> 3)};
            valid = px.x < st.size.x && px.y < rowEnd;
            Ctx c(S, st, cam, nullptr);
            const Ray r = c.raySpawn(px, i2{st.size.x, st.size.y});
            o = r.origin; d = r.direction;
            seed = tea(uint32_t(st.size.x) * uint32_t(px.y) + uint32_t(px.x), st.time);
            outIdx = uint32_t(px.y) * uint32_t(st.size.x) + uint32_t(px.x);
          } else {
            outIdx = queue ? queue[i] : i;
            const float4 ro = rayO[outIdx], rd = rayD[outIdx];
            o = mk3(ro.x, ro.y, ro.z); d = mk3(rd.x, rd.y, rd.z); tmax = ro.w; seed = rt_f2u(rd.w);
          }
          if(valid) {
            nRays++;
            live = travInit<ANY>(T, o, d, ANY ? tmax : RT_INFINITY, seed);
            if(!live) { if(ANY) occOut[outIdx] = 0u; else hitOut[outIdx] = make_float4(T.hit.t, rt_u2f(T.hit.gid), T.hit.u, T.hit.v); }
          }
        }
      }
    }
    if(__ballot(live ? 1 : 0) == 0ull) { if(exhausted) break; else continue; }
    {
      const bool was = live;
      live = travRound<ANY>(S, T, live, stack, tc);
      if(was && !live) { if(ANY) occOut[outIdx] = T.found ? 1u : 0u; else hitOut[outIdx] = make_float4(T.hit.t, rt_u2f(T.hit.gid), T.hit.u, T.hit.v); }
    }
  }
  if(F.counters) {
    atomicAdd(&F.counters[ANY ? 1 : 0], (unsigned long long)nRays);
    atomicAdd(&F.counters[2], (unsigned long long)tc.nodes);
    atomicAdd(&F.counters[3], (unsigned long long)tc.tris);
    atomicAdd(&F.counters[6], (unsigned long long)tc.rounds);
    atomicAdd(&F.counters[7], (unsigned long long)tc.live);
  }
}

// ================================================================================================================
// direct stage
// ================================================================================================================
// primary rays: raySpawn + ClosestHit (direct_stage.comp:152, 279-280); dense, tile ordered (coherent)
__global__ __launch_bounds__(64) void k_primary(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY)
{
  extern __shared__ uint2 s_stack[];
  const TileCoord tile = tileOf(tilesX, tilesY);
  if(!tile.valid) return;
  const int lane = int(threadIdx.x);
  const i2 px{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + (lane >> 3)};
  if(px.x >= st.size.x || px.y >= rowEnd) return;
  Ctx c(S, st, cam, s_stack + lane);
  const uint32_t seed = tea(uint32_t(st.size.x) * uint32_t(px.y) + uint32_t(px.x), st.time);
  const Ray r = c.raySpawn(px, i2{st.size.x, st.size.y});
  RayHit hit; TravCounters tc{0, 0};
  const unsigned long long t0 = F.counters ? wall_clock64() : 0ull;
  traceRay<false>(S, r.origin, r.direction, RT_INFINITY, seed, s_stack + lane, hit, tc);
  F.hitRec[size_t(px.y) * st.size.x + px.x] = make_float4(hit.t, rt_u2f(hit.gid), hit.u, hit.v);
  if(F.counters) {
    // instrumented mode only: per-lane traversal statistics for profiling scripts (DirB is free during the direct stage)
    const unsigned long long t1 = wall_clock64();
    F.denoiseDirB[size_t(px.y) * st.size.x + px.x] = make_float4(float(t1 - t0), float(tc.nodes), float(tc.tris), float(t0 & 0xffffffull));
    atomicAdd(&F.counters[0], 1ull);
    atomicAdd(&F.counters[2], (unsigned long long)tc.nodes);
    atomicAdd(&F.counters[3], (unsigned long long)tc.tris);
    atomicAdd(&F.counters[6], (unsigned long long)tc.rounds);
    atomicAdd(&F.counters[7], (unsigned long long)tc.live);
  }
}

// direct_stage.comp:155-215: everything between the primary hit and the shadow ray (direct_phases.h)
__global__ __launch_bounds__(64) void k_direct_shade(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY, int genOnly)
{
  const TileCoord tile = tileOf(tilesX, tilesY);
  if(!tile.valid) return;
  const int lane = int(threadIdx.x);
  const i2 px{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + (lane >> 3)};
  const bool inside = px.x < st.size.x && px.y < rowEnd;
  bool wantShadow = false;
  size_t index = 0;
  if(inside) {
    Ctx c(S, st, cam, nullptr);
    index = size_t(px.y) * st.size.x + px.x;
    c.seed = tea(uint32_t(st.size.x) * uint32_t(px.y) + uint32_t(px.x), st.time);
    const float4 h = F.hitRec[index];
    c.hit.t = h.x; c.hit.gid = rt_f2u(h.y); c.hit.u = h.z; c.hit.v = h.w;
    float4 so = make_float4(0, 0, 0, 0), sd = make_float4(0, 0, 0, 0);
    wantShadow = directShadePixel(c, F, st, px, genOnly, so, sd);
    if(F.status[index] != ST_DONE) { F.shadowO[index] = so; F.shadowD[index] = sd; F.occ[index] = 0u; }
    if(F.counters) { atomicAdd(&F.counters[4], (unsigned long long)c.nShaded); atomicAdd(&F.counters[5], (unsigned long long)c.nRis); }
  }
  const uint32_t slot = queueSlot(&F.qcount[CNT_SHADOW], wantShadow);
  if(wantShadow) F.shadowQ[slot] = uint32_t(index);
}

// direct_stage.comp:208-270: visibility of the winner, temporal reuse, store, shade (direct_phases.h)
__global__ __launch_bounds__(64) void k_direct_resolve(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY, int genOnly)
{
  const TileCoord tile = tileOf(tilesX, tilesY);
  if(!tile.valid) return;
  const int lane = int(threadIdx.x);
  const i2 px{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + (lane >> 3)};
  if(px.x >= st.size.x || px.y >= rowEnd) return;
  const size_t index = size_t(px.y) * st.size.x + px.x;
  if(F.status[index] == ST_DONE) return;
  Ctx c(S, st, cam, nullptr);
  directResolvePixel(c, F, st, cam, px, F.occ[index] != 0u, genOnly);
}

// ================================================================================================================
// indirect stage
// ================================================================================================================
enum : uint32_t { PF_MULTIBOUNCE = 1u, PF_PENDING = 2u, PF_ACTIVE = 4u };

RT_DEV void pushClosest(const DevFrame& F, uint32_t pxi, f3 o, f3 d, uint32_t seed)
{
  F.rayCO[pxi] = make_float4(o.x, o.y, o.z, RT_INFINITY);
  F.rayCD[pxi] = make_float4(d.x, d.y, d.z, rt_u2f(seed));
}

// indirect_stage.comp:270-299 + the depth-1 head of pathTraceIndirect (:129-186)
__global__ __launch_bounds__(64) void k_ind_init(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY)
{
  const TileCoord tile = tileOf(tilesX, tilesY);
  if(!tile.valid) return;
  const int lane = int(threadIdx.x);
  const i2 indSize{st.size.x / 2, st.size.y / 2};
  const i2 px{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + (lane >> 3)};
  Ctx c(S, st, cam, nullptr);
  c.imageCoords = px;
  c.seed = tea(uint32_t(indSize.x) * uint32_t(px.y) + uint32_t(px.x), st.time);
  int mb = 0;
  if(lane == 0) mb = rnd(c.seed) < 0.25f ? 1 : 0;  // TILED_MULTIBOUNCE :283-288 (wave-uniform: one readfirstlane)
  const bool multiBounce = __builtin_amdgcn_readfirstlane(mb) != 0;
  const bool inside = px.x < indSize.x && px.y < indSize.y && px.y < rowEnd;
  bool want = false;
  uint32_t pxi = 0;
  if(inside) {
    pxi = uint32_t(px.y) * uint32_t(indSize.x) + uint32_t(px.x);
    const Ray ray = c.raySpawn(px, indSize);
    GState g0; float depth;
    PathRec P;
    P.flags = multiBounce ? PF_MULTIBOUNCE : 0u;
    P.gi = newGISample();
    P.primSamplePdf = 0.f; P.samplePdf = 0.f;
    P.pending = rt_vec3{0, 0, 0};
    if(!stateFromGBuffer(loadG(F.thisG, F, i2{px.x * 2, px.y * 2}), ray, g0, depth)) {
      storeImg(F.denoiseIndA, F, px, mk4(0, 0, 0, 0));
      P.flags |= 0x80000000u;  // G-buffer miss: nothing to finish
    } else {
      g0.position += g0.ffnormal * 2e-2f;
      const f3 throughput = mk3(multiBounce ? 4.0f : 1.0f);
      Material mat = g0.mat; mat.albedo = mk3(1.0f);
      const f3 wo = -ray.direction;
      f3 sampleWi = mk3(0.0f); float samplePdf = 0.0f;
      c.Sample(mat, wo, g0.ffnormal, sampleWi, samplePdf);
      if(!Ctx::IsPdfInvalid(samplePdf)) {
        P.primSamplePdf = samplePdf; P.samplePdf = samplePdf;
        P.gi.xv = toR(g0.position); P.gi.nv = toR(g0.ffnormal);
        pushClosest(F, pxi, OffsetRay(g0.position, g0.ffnormal), sampleWi, c.seed);
        P.flags |= PF_ACTIVE;
        want = true;
      }
      P.throughput = toR(throughput);
      P.position = toR(g0.position); P.ffnormal = toR(g0.ffnormal);
      P.roughness = mat.roughness; P.metallic = mat.metallic; P.albedo = rt_vec3{1.f, 1.f, 1.f};
    }
    P.seed = c.seed;
    F.path[pxi] = P;
  }
  const uint32_t slot = queueSlot(&F.qcount[cntC(1)], want);
  if(want) F.qC[1][slot] = pxi;
}

// One iteration of the bounce loop for the paths whose ray of `depth` was just traced (indirect_stage.comp:186-217),
// followed by the head of iteration depth+1 (:141-184): NEE sampling (shadow ray enqueued) and BSDF sampling (next ray enqueued).
__global__ __launch_bounds__(64) void k_ind_bounce(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int depth)
{
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  const bool live = i < F.qcount[cntC(depth)];
  bool wantC = false, wantA = false;
  uint32_t pxi = 0;
  if(live) {
    pxi = F.qC[depth & 1][i];
    PathRec P = F.path[pxi];
    Ctx c(S, st, cam, nullptr);
    c.seed = P.seed;
    // NEE of this depth was sampled by the previous kernel; its shadow ray has been traced since (:143-151)
    if(P.flags & PF_PENDING) {
      if(F.occH[pxi] == 0u) P.gi.L = toR(mk3(P.gi.L) + mk3(P.pending));
      P.flags &= ~PF_PENDING;
    }
    P.flags &= ~PF_ACTIVE;
    const float4 h = F.hitC[pxi];
    c.hit.t = h.x; c.hit.gid = rt_f2u(h.y); c.hit.u = h.z; c.hit.v = h.w;
    const float4 rd = F.rayCD[pxi];
    Ray ray{mk3(0.f), mk3(rd.x, rd.y, rd.z)};
    const f3 sampleWi = ray.direction;
    const f3 throughput = mk3(P.throughput);
    bool done = false;
    State state = zeroState();
    if(c.hit.t >= RT_INFINITY - 1e-4f) {  // :188-202
      if(depth > 1) {
        float lightPdf;
        const f3 Li = c.EnvEval(sampleWi, lightPdf);
        const float weight = MISw(st, P.samplePdf, lightPdf);
        P.gi.L = toR(mk3(P.gi.L) + Li * throughput * weight);
      } else {
        P.gi.xs = toR(mk3(P.position) + sampleWi * RT_INFINITY * 0.8f);
        P.gi.ns = toR(-sampleWi);
      }
      done = true;
    } else {
      state = c.GetState(ray.direction);
      c.GetMaterials(state, ray);
      if(state.isEmitter) {  // :207-219
        if(depth > 1) {
          float lightPdf;
          const f3 Li = c.LightEval(state, c.hit.t, sampleWi, lightPdf);
          const float weight = MISw(st, P.samplePdf, lightPdf);
          P.gi.L = toR(mk3(P.gi.L) + Li * throughput * weight);
        } else {
          P.gi.xs = toR(state.position);
          P.gi.ns = toR(state.ffnormal);
        }
        done = true;
      } else if(depth == 1) {
        P.gi.xs = toR(state.position);
        P.gi.ns = toR(state.ffnormal);
      }
    }
    if(!done && depth < st.maxDepth) {
      // ---- head of iteration depth+1 ----
      const f3 wo = -ray.direction;
      if(st.MIS > 0) {  // SampleDirectLight, pathtrace.glsl:185-203, cut at its Occlusion()
        rt_light_sample ls;
        const float lightPdf = c.SampleDirectLightNoVisibility(state.position, ls);
        if(!Ctx::IsPdfInvalid(lightPdf)) {
          const f3 wi = mk3(ls.wi);
          const f3 org = OffsetRay(state.position, state.ffnormal);
          const float tmax = ((ls.dist - rt_abs(org.x - state.position.x)) - rt_abs(org.y - state.position.y)) - rt_abs(org.z - state.position.z);
          const float BSDFPdf = metallicWorkflowPdf(state.mat, state.ffnormal, wo, wi);
          const float weight = MISw(st, lightPdf, BSDFPdf);
          P.pending = toR(mk3(ls.Li) * metallicWorkflowBSDF(state.mat, state.ffnormal, wo, wi) * absDot(state.ffnormal, wi) * throughput / lightPdf * weight);
          P.flags |= PF_PENDING;
          F.rayAO[pxi] = make_float4(org.x, org.y, org.z, tmax);
          F.rayAD[pxi] = make_float4(wi.x, wi.y, wi.z, rt_u2f(c.seed));
          wantA = true;
        }
      }
      f3 nextWi = mk3(0.0f); float samplePdf = 0.0f;
      const f3 sampleBSDF = c.Sample(state.mat, wo, state.ffnormal, nextWi, samplePdf);
      if(!Ctx::IsPdfInvalid(samplePdf) && (P.flags & PF_MULTIBOUNCE)) {  // `if(!multiBounce) return;` :163-166
        P.throughput = toR(throughput * (sampleBSDF / samplePdf * absDot(state.ffnormal, nextWi)));
        P.samplePdf = samplePdf;
        P.position = toR(state.position);
        pushClosest(F, pxi, OffsetRay(state.position, state.ffnormal), nextWi, c.seed);
        P.flags |= PF_ACTIVE;
        wantC = true;
      }
    }
    P.seed = c.seed;
    F.path[pxi] = P;
    if(F.counters) { atomicAdd(&F.counters[4], (unsigned long long)c.nShaded); atomicAdd(&F.counters[5], (unsigned long long)c.nRis); }
  }
  const uint32_t sa = queueSlot(&F.qcount[cntA(depth + 1)], wantA);
  if(wantA) F.qA[sa] = pxi;
  const uint32_t sc = queueSlot(&F.qcount[cntC(depth + 1)], wantC);
  if(wantC) F.qC[(depth + 1) & 1][sc] = pxi;
}

// ReSTIRIndirect + main's store, indirect_stage.comp:228-268, 301-308
__global__ __launch_bounds__(64) void k_ind_finish(DevScene S, DevFrame F, rt_state st, rt_scene_camera cam, int rowBegin, int rowEnd, int tilesX, int tilesY)
{
  const TileCoord tile = tileOf(tilesX, tilesY);
  if(!tile.valid) return;
  const int lane = int(threadIdx.x);
  const i2 indSize{st.size.x / 2, st.size.y / 2};
  const i2 px{tile.x * 8 + (lane & 7), rowBegin + tile.y * 8 + (lane >> 3)};
  if(px.x >= indSize.x || px.y >= indSize.y || px.y >= rowEnd) return;
  const uint32_t pxi = uint32_t(px.y) * uint32_t(indSize.x) + uint32_t(px.x);
  PathRec P = F.path[pxi];
  if(P.flags & 0x80000000u) return;
  Ctx c(S, st, cam, nullptr);
  c.imageCoords = px;
  c.seed = P.seed;
  if(P.flags & PF_PENDING) {  // a path that ended right after its NEE sample (invalid BSDF sample / single-bounce tile / last depth)
    if(F.occH[pxi] == 0u) P.gi.L = toR(mk3(P.gi.L) + mk3(P.pending));
  }
  const Ray ray = c.raySpawn(px, indSize);
  GState primState; float depth;
  stateFromGBuffer(loadG(F.thisG, F, i2{px.x * 2, px.y * 2}), ray, primState, depth);
  primState.position += primState.ffnormal * 2e-2f;
  const f3 primWo = -ray.direction;
  rt_gi_sample gi = P.gi;

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
    gi.pHat = resvToScalar(mk3(gi.L));
    sampleWeight = gi.pHat / P.primSamplePdf;
    if(rt_isnan(sampleWeight) || sampleWeight < 0.0f) sampleWeight = 0.0f;
  }
  {
    const float rr = rnd(c.seed);
    resv.weight += sampleWeight; resv.num += 1;
    if(rr * resv.weight < sampleWeight) resv.giSample = gi;
  }
  if(resvInvalidW(resv.weight)) { resv.num = 0; resv.weight = 0.f; resv.bigW = 0.f; }
  resvClamp(resv, st.reservoirClamp * 2);
  F.thisIndirectResv[pxi] = resv;
  gi = resv.giSample;
  if(!resvInvalidW(resv.weight) && GISampleValid(gi)) {
    const f3 primWi = normalize(mk3(gi.xs) - mk3(gi.xv));
    Material pm = primState.mat;
    pm.albedo = mk3(1.0f);
    const float bigW = resv.weight / (resvToScalar(mk3(resv.giSample.L)) * float(resv.num));
    indirect = mk3(gi.L) * metallicWorkflowBSDF(pm, mk3(gi.nv), primWo, primWi) * satDot(mk3(gi.nv), primWi) * bigW;
  }
  f3 pixelColor = HDRToLDR(c.clampRadiance(indirect));
  pixelColor = c.clampRadiance(pixelColor);
  storeImg(F.denoiseIndA, F, px, mk4(pixelColor, 1.0f));
}

// ================================================================================================================
// host-side sequencing
// ================================================================================================================
hipError_t launchStageWavefront(hipStream_t stream, const DevScene& Sin, const DevFrame& F, const rt_state& st, const rt_scene_camera& cam, int stage, int level,
                                int rowBegin, int rowEnd)
{
  const bool isDirect = stage == RT_STAGE_DIRECT || stage == RT_STAGE_DIRECT_GEN;
  if(!isDirect && stage != RT_STAGE_INDIRECT) return launchStage(stream, Sin, F, st, cam, stage, level, rowBegin, rowEnd);
  DevScene S = Sin;
  S.stackEntries = F.stackLds > 0 ? std::min(F.stackLds, S.stackTotal) : S.stackTotal;   // LDS part of the traversal stack for this launch
  if(stage == RT_STAGE_INDIRECT) S.stackOvf = Sin.stackOvfInd;
  // the spatial reuse modes exist in the fused organisation only (k_direct_stage + k_direct_spatial)
  if(stage == RT_STAGE_DIRECT && (st.ReSTIRState == RT_RESTIR_SPATIAL || st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL))
    return launchStage(stream, S, F, st, cam, stage, level, rowBegin, rowEnd);
  const bool half = !isDirect;
  const int gw = half ? st.size.x / 2 : st.size.x, gh = half ? st.size.y / 2 : st.size.y;
  if(rowEnd <= 0 || rowEnd > gh) rowEnd = gh;
  if(rowBegin < 0) rowBegin = 0;
  if(rowBegin >= rowEnd || gw <= 0) return hipSuccess;
  const int tilesX = (gw + 7) / 8, tilesY = (rowEnd - rowBegin + 7) / 8;
  const int nTiles = tilesX * tilesY;
  const dim3 grid(tileGrid(tilesX, tilesY)), block(64);
  const size_t lds = size_t(S.stackEntries) * 64 * sizeof(uint2);
  const unsigned cap = unsigned(nTiles);  // queue capacity in waves: at most one ray per pixel of the band
  static const int persistent = getenv("RESTIR_PERSISTENT") ? atoi(getenv("RESTIR_PERSISTENT")) : 1;
  static const int wavesPerCU = getenv("RESTIR_WAVES_PER_CU") ? atoi(getenv("RESTIR_WAVES_PER_CU")) : 16;
  const dim3 pgrid(unsigned(std::min<long long>(256ll * wavesPerCU, (long long)cap)));
  if(S.stackTotal > S.stackEntries && std::max(grid.x, pgrid.x) * 64u > S.stackOvfThreads) return hipErrorInvalidValue;
  uint32_t* heads = F.qcount + 128;
  hipError_t e = hipMemsetAsync(F.qcount, 0, 192 * sizeof(uint32_t), stream);  // slots 192.. belong to the tile lists / history-miss flag
  if(e != hipSuccess) return e;
  if(isDirect) {
    const int genOnly = stage == RT_STAGE_DIRECT_GEN ? 1 : 0;
    if(persistent)
      hipLaunchKernelGGL((k_trace_p<false, true>), pgrid, block, lds, stream, S, F, st, cam, (const uint32_t*)nullptr, (const uint32_t*)nullptr, uint32_t(nTiles) * 64u,
                         heads + 1, (const float4*)nullptr, (const float4*)nullptr, F.hitRec, (uint32_t*)nullptr, rowBegin, rowEnd, tilesX);
    else
      hipLaunchKernelGGL(k_primary, grid, block, lds, stream, S, F, st, cam, rowBegin, rowEnd, tilesX, tilesY);
    hipLaunchKernelGGL(k_direct_shade, grid, block, 0, stream, S, F, st, cam, rowBegin, rowEnd, tilesX, tilesY, genOnly);
    if(persistent)
      hipLaunchKernelGGL((k_trace_p<true, false>), pgrid, block, lds, stream, S, F, st, cam, (const uint32_t*)F.shadowQ, (const uint32_t*)(F.qcount + CNT_SHADOW), 0u,
                         heads + 0, (const float4*)F.shadowO, (const float4*)F.shadowD, (float4*)nullptr, F.occ, 0, 0, 0);
    else
      hipLaunchKernelGGL(k_trace<true>, dim3(cap), block, lds, stream, S, F, (const uint32_t*)F.shadowQ, (const uint32_t*)(F.qcount + CNT_SHADOW),
                         (const float4*)F.shadowO, (const float4*)F.shadowD, (float4*)nullptr, F.occ);
    hipLaunchKernelGGL(k_direct_resolve, grid, block, 0, stream, S, F, st, cam, rowBegin, rowEnd, tilesX, tilesY, genOnly);
  } else {
    if(st.maxDepth > 24) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_ind_init, grid, block, 0, stream, S, F, st, cam, rowBegin, rowEnd, tilesX, tilesY);
    for(int d = 1; d <= st.maxDepth; d++) {
      if(persistent) {
        if(d > 1)
          hipLaunchKernelGGL((k_trace_p<true, false>), pgrid, block, lds, stream, S, F, st, cam, (const uint32_t*)F.qA, (const uint32_t*)(F.qcount + 32 + d), 0u,
                             heads + 32 + d, (const float4*)F.rayAO, (const float4*)F.rayAD, (float4*)nullptr, F.occH, 0, 0, 0);
        hipLaunchKernelGGL((k_trace_p<false, false>), pgrid, block, lds, stream, S, F, st, cam, (const uint32_t*)F.qC[d & 1], (const uint32_t*)(F.qcount + 1 + d), 0u,
                           heads + 1 + d, (const float4*)F.rayCO, (const float4*)F.rayCD, F.hitC, (uint32_t*)nullptr, 0, 0, 0);
      } else {
        if(d > 1)
          hipLaunchKernelGGL(k_trace<true>, dim3(cap), block, lds, stream, S, F, (const uint32_t*)F.qA, (const uint32_t*)(F.qcount + 32 + d),
                             (const float4*)F.rayAO, (const float4*)F.rayAD, (float4*)nullptr, F.occH);
        hipLaunchKernelGGL(k_trace<false>, dim3(cap), block, lds, stream, S, F, (const uint32_t*)F.qC[d & 1], (const uint32_t*)(F.qcount + 1 + d),
                           (const float4*)F.rayCO, (const float4*)F.rayCD, F.hitC, (uint32_t*)nullptr);
      }
      hipLaunchKernelGGL(k_ind_bounce, dim3(cap), block, 0, stream, S, F, st, cam, d);
    }
    hipLaunchKernelGGL(k_ind_finish, grid, block, 0, stream, S, F, st, cam, rowBegin, rowEnd, tilesX, tilesY);
  }
  return hipGetLastError();
}

}  // namespace RT_VARIANT
}  // namespace rt
