// rt_api.cpp — the C-ABI of include/rt_abi.h over HIP: context, HBM residency, stage launches, readback.
// Stands where the reference has Renderer / Scene / AccelStructure / HdrSampling talking to Vulkan
// (src/renderer.cpp:62-302, src/scene.cpp:453-508, src/accelstruct.cpp:55-65, src/hdr_sampling.cpp:79-95).
#include <hip/hip_runtime.h>
#include <array>
#include <atomic>
#include <memory>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <map>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../include/rt_abi.h"
#include "../../include/rt_cpus.h"
#include "bvh8_builder.h"
#include "stages.h"

using namespace rt;

struct rt_ctx {
  int device = 0;
  hipStream_t ownStream = nullptr, stream = nullptr;
  hipStream_t sideStream = nullptr;           // direct A-Trous runs here, concurrently with the indirect stage
  hipStream_t indStream = nullptr;            // overlap 2: the indirect stage of frame f runs here, next to direct(f+1) on `stream`
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  // overlap 2 (frames in flight): ring of per-frame dependency events — direct done / indirect done / frame done
  hipEvent_t evD[4] = {}, evI[4] = {}, evDone[4] = {};
  uint64_t seq = 0;          // frames submitted through the pipelined path since the last join
  bool inFlight = false;     // work may be pending on indStream / sideStream
  // One stream per (role, level), created on first use and kept until rt_destroy: a stream takes a hardware queue of its priority class round-robin at creation, so a
  // context that destroyed and re-created its streams while trying settings ended with its filter stream on the main stream's queue (the 5th normal-priority stream of
  // the process lands on the 1st one's queue): 3.38 -> 3.71 ms per frame, found by two back-to-back bench lines in round 5.
  hipStream_t indStreams[3] = {nullptr, nullptr, nullptr}, sideStreams[3] = {nullptr, nullptr, nullptr};   // index = level + 1
  std::vector<hipStream_t> padStreams;   // idle streams of the creation-order probe (RESTIR_STREAM_PAD)
  int prio[3] = {0, 1, 0};   // priority level of the main / indirect / filter stream (-1 low, 0 normal, +1 high): prioSpec() at rt_create, rt_set_stream_priorities, the rule
  bool prioFromEnv = false;  // RESTIR_PRIO was set (A/B scripts stay in control)
  bool prioExplicit = false; // rt_set_stream_priorities was called: the rule stays out of it, also after rt_resize / a new scene
  bool prioDecided = false;  // the levels are final (environment, rt_set_stream_priorities, or the rule applied to the probe frames' stage times)
  float filterShare = -1.f;  // filters / (direct + indirect) of the LAST probe frame, stages run alone (the rule's input); -1: not measured
  int probeFrames = 0;       // probe frames rendered so far for the pending decision (PRIO_PROBE_FRAMES of them: warm caches, warm history)
  int denoiseSeen = -1;      // RtxState.denoise of the frames the decision was taken on: a toggle re-opens it (without the filters the share is compose alone)
  int mainIdx = -1, indIdx[3] = {-1, -1, -1}, sideIdx[3] = {-1, -1, -1};   // creation index (process-wide, g_streamsCreated) of every stream above: rt_get_stream_layout
  void* dSky = nullptr;      // SkyPre (csrc/sky.h), valid while sunAndSky.in_use == 1
  void* dPick = nullptr;     // rt_pick_result written by k_pick
  rt_sun_and_sky sunAndSky{};
  void* spareG = nullptr; void* spareMotion = nullptr;   // third G-buffer / second motion buffer (rotated per pipelined frame)
  // overlap 3 (three frames in flight): one more of each, and a third noisy / filtered direct image — direct(f) then only waits for frame f-3 (rt_render_frame)
  void* spareG2 = nullptr; void* spareMotion2 = nullptr; void* spareDirRes = nullptr;
  // The noisy indirect colour (RT_BUF_DENOISE_IND_A: written by the indirect stage, read and rewritten by its five filter levels) exists twice, by frame parity:
  // indirect(f+1) must not wait for the filters of frame f (round 3: that wait made a rank's period indirect + filters instead of max(direct, indirect)).
  // The boundary id names the buffer of the frame most recently passed to rt_render_frame / rt_run_stage / rt_select_frame.
  void* indA[2] = {nullptr, nullptr};
  int overlap = 2;           // 0 = one stream; 1 = direct A-Trous beside the indirect stage; 2 = 1 + consecutive frames overlap; 3 = 2 with a third frame in flight
  std::string err;
  // host copy of the scene (rt_build_accel runs after rt_upload_scene returns; the caller keeps ownership of its arrays)
  std::vector<rt_prim_mesh> primMeshes; std::vector<rt_vertex> vertices; std::vector<uint32_t> indices; std::vector<rt_instance> instances;
  std::vector<rt_material> materials; std::vector<DevTexture> devTextures;
  std::vector<std::vector<uint8_t>> hostAlpha;  // alpha channel of every texture (opacity micro-map build)
  std::vector<std::array<uint64_t, 2>> alphaHash;   // ... and its hash (rt_build_accel's cache key)
  // device allocations of the scene
  std::vector<void*> sceneAllocs, accelAllocs, ovfAllocs;
  DevScene ds{};
  bool haveScene = false, haveAccel = false;
  int maxDepth = 0; size_t numNodes = 0, numTris = 0, numRefs = 0, spatialSplits = 0;
  double sahNodeSteps = 0, sahTriSteps = 0;
  // screen-space buffers
  int W = 0, H = 0;
  void* bufs[RT_BUF_COUNT] = {};
  size_t bufBytes[RT_BUF_COUNT] = {};   // logical bytes (W x H elements): readback / upload size
  size_t bufAlloc[RT_BUF_COUNT] = {};   // allocated bytes: + RT_PAD_ROWS rows so equal-height row bands can be all-gathered in place
  rt_scene_camera cam{};
  // internal per-frame scratch (DevFrame)
  std::vector<void*> scratchAllocs;
  DevFrame scratch{};
  int histRow0 = 0, histRow1 = 1 << 30;  // rt_set_history_rows
  int traversal = RT_TRAVERSAL_AUTO;     // rt_set_traversal: which build of the traced kernels a launch gets
  bool counting = false;
  unsigned long long* dCounters = nullptr;
  // timing: per frame one event set; event 0 = frame start, event k = end of launch k.  Sets are harvested lazily.
  static constexpr int MAX_EV = 24, MAX_SETS = 1024;
  struct EvSet { hipEvent_t ev[MAX_EV]; int stage[MAX_EV]; int prev[MAX_EV]; int count; int last; };
  std::vector<EvSet> evSets;
  size_t evUsed = 0;
  double accStage[RT_STAGE_COUNT] = {}; double accFrame = 0; uint32_t accFrames = 0;
};

static void harvestTimings(rt_ctx* c)
{
  for(size_t s = 0; s < c->evUsed; s++) {
    rt_ctx::EvSet& E = c->evSets[s];
    for(int k = 1; k < E.count; k++) {
      float ms = 0.f;
      if(E.stage[k] >= 0 && hipEventElapsedTime(&ms, E.ev[E.prev[k]], E.ev[k]) == hipSuccess) c->accStage[E.stage[k]] += ms;
    }
    float ms = 0.f;
    if(E.count > 1 && hipEventElapsedTime(&ms, E.ev[0], E.ev[E.last]) == hipSuccess) { c->accFrame += ms; c->accFrames++; }
  }
  c->evUsed = 0;
}

// Wait (host) for everything this context has submitted.
static hipError_t syncAll(rt_ctx* c)
{
  hipError_t e = hipStreamSynchronize(c->stream);
  if(e == hipSuccess && c->indStream) e = hipStreamSynchronize(c->indStream);
  if(e == hipSuccess && c->sideStream) e = hipStreamSynchronize(c->sideStream);
  c->inFlight = false; c->seq = 0;
  return e;
}
// Make `stream` (device side) wait for the frames in flight on the internal streams: needed before work that is submitted
// to `stream` alone (rt_run_stage, the non-pipelined rt_render_frame) may touch the frame buffers.
static hipError_t joinInFlight(rt_ctx* c)
{
  if(!c->inFlight) return hipSuccess;
  const int r = int((c->seq - 1) & 3);
  hipError_t e = hipStreamWaitEvent(c->stream, c->evI[r], 0);
  if(e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->evDone[r], 0);
  c->inFlight = false; c->seq = 0;
  return e;
}

typedef hipError_t (*StageLauncher)(hipStream_t, const DevScene&, const DevFrame&, const rt_state&, const rt_scene_camera&, int, int, int, int);
// Which build of stages.hip runs a launch.  The traced kernels exist twice: the throughput build (one ray per lane, majority-vote rounds, 4-5 waves per
// SIMD: full frames are bound by instruction issue) and the latency build (csrc/stages_lat.hip: eight lanes per ray, a workgroup of eight waves per tile:
// a row band of a multi-GPU frame or a small image is bound by the dependent steps of its slowest rays, and the latency build shortens the step).
// RT_TRAVERSAL_AUTO decides by the number of 8x8 tiles of the launch; the thresholds are where the two builds measured equal on the benchmark scene
// (profiles/r03_band_chunk_ab.txt, profiles/r03_lat_wide_ab.txt):
//   launches that run alone (rt_set_overlap 0/1; the barrier schedule of rt_mgpu): direct stage up to 2000 tiles (64 rows of 1080p), indirect stage up to
//     2000 half-res tiles (256 rows of 1080p) — round 4, with gang mode in the latency build (round 3: 1440 / 1536; profiles/r04_band_thresholds.txt, profiles/r04_gang_ab.txt);
//   launches that share the CUs with the kernels of another frame (rt_set_overlap 2; frames in flight in rt_mgpu): 512 / 640 — eight lanes per ray buy a
//     short chain with 2-4x the lane-cycles, which a co-running kernel would have used.
// The counting build is a throughput build.  Bit-identical either way.  RESTIR_LAT_TILES / RESTIR_LAT_TILES_IND override both cases (experiments).
static int envInt(const char* name) { const char* e = getenv(name); return e ? atoi(e) : -1; }
static int latTilesDirect(bool shared)
{
  static const int alone = envInt("RESTIR_LAT_TILES"), sh = envInt("RESTIR_LAT_TILES_SHARED");
  return shared ? (sh >= 0 ? sh : (alone >= 0 ? alone : 512)) : (alone >= 0 ? alone : 2000);
}
static int latTilesIndirect(bool shared)
{
  static const int alone = envInt("RESTIR_LAT_TILES_IND"), sh = envInt("RESTIR_LAT_TILES_IND_SHARED");
  return shared ? (sh >= 0 ? sh : (alone >= 0 ? alone : 640)) : (alone >= 0 ? alone : 2000);
}
static StageLauncher stageLauncher(const rt_ctx* c, const rt_state& st, int stage, int rowBegin, int rowEnd)
{
  bool lat = false;
  if(!c->counting && (stage == RT_STAGE_DIRECT || stage == RT_STAGE_INDIRECT)) {
    if(c->traversal == RT_TRAVERSAL_LATENCY) lat = true;
    else if(c->traversal == RT_TRAVERSAL_AUTO) {
      const bool half = stage == RT_STAGE_INDIRECT;
      const int gw = half ? st.size.x / 2 : st.size.x, gh = half ? st.size.y / 2 : st.size.y;
      const int r1 = (rowEnd <= 0 || rowEnd > gh) ? gh : rowEnd, r0 = rowBegin < 0 ? 0 : rowBegin;
      const long tiles = long((gw + 7) / 8) * long((std::max(0, r1 - r0) + 7) / 8);
      lat = tiles <= (half ? latTilesIndirect(c->overlap >= 2) : latTilesDirect(c->overlap >= 2));
    }
  }
  if(c->ds.sky) return lat ? rt::sky_lat::launchStage : (c->counting ? rt::sky_cnt::launchStage : rt::sky::launchStage);
  return lat ? rt::base_lat::launchStage : (c->counting ? rt::base_cnt::launchStage : rt::base::launchStage);
}

// ---- host side of rt_build_accel, cached by scene content (buildHostAccel below) ---------------------------------------------------------------------
struct HostAccel {
  uint64_t key0 = 0, key1 = 0;      // content hash of everything the build reads
  BuildOutput bo;
  std::vector<AlphaRec> alpha;      // bgra left null: patched per device from alphaTex
  std::vector<int32_t> alphaTex;    // texture index of every alpha record, -1 = none
};
static std::mutex g_accelMutex;
static std::shared_ptr<const HostAccel> g_accelCache;   // the most recent build; dropped with the last context
static std::atomic<int> g_liveCtx{0};

static thread_local std::string g_createErr;

#define RT_HIP(ctx, call)                                                                                                   \
  do {                                                                                                                      \
    hipError_t e_ = (call);                                                                                                 \
    if(e_ != hipSuccess) {                                                                                                  \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                                       \
      return (e_ == hipErrorOutOfMemory) ? RT_ERR_OOM : RT_ERR_HIP;                                                         \
    }                                                                                                                       \
  } while(0)

// ---- opacity micro-map ---------------------------------------------------------------------------------------------
// Cell (i,j) covers barycentrics u in [i/8,(i+1)/8], v in [j/8,(j+1)/8].  Its texture footprint is the bounding box of the
// four corners' texcoords, widened by the bilinear footprint and one texel of slack for float rounding; the cell is
// classified only when min/max alpha over that footprint decide HitTest (traceray_rq.glsl:86-101) with a safety margin.
static int wrapTexel(int i, int n, int mode)
{
  if(mode == RT_WRAP_CLAMP) return i < 0 ? 0 : (i >= n ? n - 1 : i);
  if(mode == RT_WRAP_MIRROR) { int p = 2 * n; int m = i % p; if(m < 0) m += p; return m < n ? m : p - 1 - m; }
  int m = i % n; if(m < 0) m += n;
  return m;
}
static void buildOpacityMap(const AlphaRec& a, const std::vector<uint8_t>* alpha, uint32_t omm[4])
{
  omm[0] = omm[1] = omm[2] = omm[3] = 0;
  const bool mask = a.alphaMode == RT_ALPHA_MASK;
  for(int j = 0; j < 8; j++)
    for(int i = 0; i + j < 8; i++) {
      double amin = 1.0, amax = 1.0;
      if(alpha && a.w > 0 && a.h > 0) {
        double fx0 = 1e300, fx1 = -1e300, fy0 = 1e300, fy1 = -1e300;
        for(int c = 0; c < 4; c++) {
          const double u = (i + (c & 1)) / 8.0, v = (j + (c >> 1)) / 8.0, w0 = 1.0 - u - v;
          const double tx = a.uv0x * w0 + a.uv1x * u + a.uv2x * v, ty = a.uv0y * w0 + a.uv1y * u + a.uv2y * v;
          fx0 = std::min(fx0, tx * a.w); fx1 = std::max(fx1, tx * a.w); fy0 = std::min(fy0, ty * a.h); fy1 = std::max(fy1, ty * a.h);
        }
        if(!(fx1 - fx0 < 4096.0 && fy1 - fy0 < 4096.0) || !std::isfinite(fx0 + fx1 + fy0 + fy1)) continue;  // unknown
        const long long x0 = (long long)std::floor(fx0 - 0.5) - 1, x1 = (long long)std::floor(fx1 - 0.5) + 2;
        const long long y0 = (long long)std::floor(fy0 - 0.5) - 1, y1 = (long long)std::floor(fy1 - 0.5) + 2;
        if((x1 - x0 + 1) * (y1 - y0 + 1) > 65536) continue;  // footprint too large to scan: leave unknown
        int lo = 255, hi = 0;
        for(long long y = y0; y <= y1; y++) {
          const int yy = wrapTexel(int(y), a.h, a.wrapT);
          for(long long x = x0; x <= x1; x++) {
            const int t = (*alpha)[size_t(yy) * a.w + wrapTexel(int(x), a.w, a.wrapS)];
            lo = std::min(lo, t); hi = std::max(hi, t);
          }
        }
        amin = lo / 255.0; amax = hi / 255.0;
      }
      const double omin = double(a.baseAlpha) * amin, omax = double(a.baseAlpha) * amax;
      uint32_t state = 0;
      if(a.baseAlpha >= 0.f) {
        if(mask) {
          if(omin > double(a.cutoff) + 1e-4) state = 1;        // opacity 1 everywhere: rand > 1 never
          else if(omax < double(a.cutoff) - 1e-4) state = 2;   // opacity 0 everywhere
        } else {
          if(omin >= 1.0 + 1e-4) state = 1;                    // blend: rand in [0,1) never exceeds opacity >= 1
          else if(omax <= 0.0) state = 2;                      // exactly zero alpha
        }
      }
      const int cell = j * 8 + i;
      omm[cell >> 4] |= state << ((cell & 15) * 2);
    }
}

static int fail(rt_ctx* c, int code, const char* msg) { c->err = msg; return code; }

// CU-partitioned streams (experiment, round 4: profiles/r04_cu_mask_ab.txt).  RESTIR_CU_SPLIT="a-b,c-d,e-f": the main (direct stage), indirect and filter
// streams of rt_render_frame's overlapped schedules are created with hipExtStreamCreateWithCUMask and may use compute units [a,b) / [c,d) / [e,f) of EVERY XCD
// (32 per XCD on MI355X; mask bit i selects CU i / 8 of XCD i % 8 — scripts/probe/cu_mask_probe.hip).  Unset: ordinary streams, the whole chip.
static bool cuSplitRange(int which, int& lo, int& hi)
{
  static int r[3][2]; static int have = -1;
  if(have < 0) {
    have = 0;
    if(const char* e = getenv("RESTIR_CU_SPLIT"))
      if(sscanf(e, "%d-%d,%d-%d,%d-%d", &r[0][0], &r[0][1], &r[1][0], &r[1][1], &r[2][0], &r[2][1]) == 6) have = 1;
  }
  if(!have) return false;
  lo = std::max(0, std::min(32, r[which][0])); hi = std::max(lo + 1, std::min(32, r[which][1]));
  return !(lo == 0 && hi == 32);
}
// Stream priorities of the frames-in-flight schedule.  RESTIR_PRIO = 0..3 (rounds 2-4: 0 none, 1 indirect + filter streams high, 2 indirect stream high — the
// default —, 3 filter stream high) or three characters over {-, 0, +} for the main (direct stage) / indirect / filter stream: "+00" = main stream high, "0+-" =
// indirect high and filters low.  level: -1 low, 0 normal, +1 high.
static void prioSpec(int level[3])   // (read at every rt_create: a host may change RESTIR_PRIO between contexts)
{
  level[0] = 0; level[1] = 1; level[2] = 0;
  const char* e = getenv("RESTIR_PRIO");
  if(e && strlen(e) == 3 && strspn(e, "-0+") == 3) { for(int i = 0; i < 3; i++) level[i] = e[i] == '+' ? 1 : (e[i] == '-' ? -1 : 0); }
  else if(e) { const int m = atoi(e); level[0] = 0; level[1] = (m == 1 || m == 2) ? 1 : 0; level[2] = (m == 1 || m == 3) ? 1 : 0; }
}
static hipError_t createStreamLevel(hipStream_t* s, int which, int level);
static hipError_t createStream(hipStream_t* s, int which, bool high) { return createStreamLevel(s, which, high ? 1 : 0); }
static std::atomic<int> g_streamsCreated{0};   // HIP streams this library has created in the process so far (a stream's worth depends on its place in that order)
static hipError_t createStreamLevelRaw(hipStream_t* s, int which, int level);
static hipError_t createStreamLevel(hipStream_t* s, int which, int level)
{
  const hipError_t e = createStreamLevelRaw(s, which, level);
  if(e == hipSuccess) g_streamsCreated.fetch_add(1);
  return e;
}
static hipError_t createStreamLevelRaw(hipStream_t* s, int which, int level)
{
  const bool high = level > 0;
  int lo, hi;
  if(cuSplitRange(which, lo, hi)) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for(int b = 8 * lo; b < 8 * hi; b++) mask[b >> 5] |= 1u << (b & 31);
    return hipExtStreamCreateWithCUMask(s, 8, mask);
  }
  int plo = 0, phi = 0;
  (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
  if(level < 0 && phi < plo) return hipStreamCreateWithPriority(s, hipStreamNonBlocking, plo);   // (numerically larger = lower priority)
  return (high && phi < plo) ? hipStreamCreateWithPriority(s, hipStreamNonBlocking, phi) : hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

template <class T> static int upload(rt_ctx* c, std::vector<void*>& pool, const T* src, size_t count, const T** out)
{
  void* d = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  RT_HIP(c, hipMalloc(&d, bytes));
  pool.push_back(d);
  if(count && src) RT_HIP(c, hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice));
  else RT_HIP(c, hipMemset(d, 0, bytes));
  *out = static_cast<const T*>(d);
  return RT_OK;
}
static void freePool(std::vector<void*>& pool) { for(void* p : pool) (void)hipFree(p); pool.clear(); }
// RESTIR_BVH_TIMING=1: where the seconds of rt_upload_scene / rt_build_accel go (stderr)
struct LoadTimer {
  const bool on = getenv("RESTIR_BVH_TIMING") && atoi(getenv("RESTIR_BVH_TIMING")) != 0;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char* what) { if(!on) return; const auto n = std::chrono::steady_clock::now(); fprintf(stderr, "[scene load] %-34s %.3f s\n", what, std::chrono::duration<double>(n - t).count()); t = n; }
};
static int ensureStackOverflow(rt_ctx* c);
static void reopenPriorityDecision(rt_ctx* c);
static void hashBytes(const void* p, size_t n, uint64_t& h0, uint64_t& h1);
static int stackLdsEnv() { static const int v = getenv("RESTIR_STACK_LDS") ? std::max(2, atoi(getenv("RESTIR_STACK_LDS"))) : 0; return v; }
static int stackLdsMin() { return stackLdsEnv() ? stackLdsEnv() : 6; }   // the shortest LDS stack any schedule uses: sizes the overflow areas

static size_t elemBytes(int id)
{
  switch(id) {
    case RT_BUF_GBUFFER0: case RT_BUF_GBUFFER1: return 16;
    case RT_BUF_MOTION: return 4;
    case RT_BUF_DIRECT_RESV0: case RT_BUF_DIRECT_RESV1: case RT_BUF_DIRECT_RESV_TEMP: return sizeof(rt_direct_reservoir);
    case RT_BUF_INDIRECT_RESV0: case RT_BUF_INDIRECT_RESV1: case RT_BUF_INDIRECT_RESV_TEMP: return sizeof(rt_indirect_reservoir);
    case RT_BUF_LIGHT_ID0: case RT_BUF_LIGHT_ID1: case RT_BUF_LDR: return 4;
    default: return 16;
  }
}
#ifndef RT_WAVEPROF
#define RT_WAVEPROF 0
#endif
static constexpr size_t WAVEPROF_RECORDS = 1 << 17;   // measurement builds: one record per workgroup of a traced launch
static constexpr int RT_PAD_ROWS = 128;  // full-res rows of slack behind every buffer (64 for half-res buffers)
static bool halfRes(int id) { return id == RT_BUF_INDIRECT_RESV0 || id == RT_BUF_INDIRECT_RESV1 || id == RT_BUF_INDIRECT_RESV_TEMP; }

extern "C" {

uint32_t rt_abi_version(void) { return (RT_ABI_VERSION_MAJOR << 16) | RT_ABI_VERSION_MINOR; }

const char* rt_last_error(rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_createErr.c_str(); }

}  // extern "C"
// rt_create with explicit stream levels (main / indirect / filter; nullptr = RESTIR_PRIO or the defaults): csrc/mgpu.cpp gives every rank's context the levels of its
// schedule so that the rank's streams are the CONTEXT's streams, created in the order the single-GPU schedule was tuned for (round 6) — not part of the C ABI.
int rtCreateWithLevels(rt_ctx** out, int device, const int* levels);
extern "C" {
int rt_create(rt_ctx** out, int device) { return rtCreateWithLevels(out, device, nullptr); }
}
int rtCreateWithLevels(rt_ctx** out, int device, const int* levels)
{
  if(!out) { g_createErr = "rt_create: out is NULL"; return RT_ERR_INVALID_ARG; }
  *out = nullptr;
  int n = 0;
  if(hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_createErr = "rt_create: no HIP device (this library has no CPU path)"; return RT_ERR_NO_DEVICE; }
  if(device < 0 || device >= n) { g_createErr = "rt_create: device index out of range"; return RT_ERR_INVALID_ARG; }
  if(hipSetDevice(device) != hipSuccess) { g_createErr = "rt_create: hipSetDevice failed"; return RT_ERR_HIP; }
  rt_ctx* c = new(std::nothrow) rt_ctx();
  if(!c) { g_createErr = "rt_create: out of host memory"; return RT_ERR_OOM; }
  g_liveCtx.fetch_add(1);   // counted from here on: every exit below goes through rt_destroy, which un-counts it
  c->device = device;
  {
    prioSpec(c->prio); c->prioFromEnv = getenv("RESTIR_PRIO") != nullptr; c->prioDecided = c->prioFromEnv;
    if(levels) { for(int i = 0; i < 3; i++) c->prio[i] = std::max(-1, std::min(1, levels[i])); c->prioDecided = c->prioExplicit = true; }
    bool ok = createStreamLevel(&c->ownStream, 0, c->prio[0]) == hipSuccess;
    if(!ok) { g_createErr = "rt_create: hipStreamCreate failed"; rt_destroy(c); return RT_ERR_HIP; }
    c->mainIdx = g_streamsCreated.load() - 1;
    c->stream = c->ownStream;
    for(int i = 0; i < 4; i++) {
      ok = ok && hipEventCreateWithFlags(&c->evD[i], hipEventDisableTiming) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&c->evI[i], hipEventDisableTiming) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&c->evDone[i], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming) == hipSuccess;
    if(!ok) { g_createErr = "rt_create: creating the internal events failed"; rt_destroy(c); return RT_ERR_HIP; }
  }
  if(const char* e = getenv("RESTIR_OVERLAP")) c->overlap = atoi(e);
  if(hipMalloc(reinterpret_cast<void**>(&c->dCounters), 8 * sizeof(unsigned long long)) != hipSuccess) { g_createErr = "rt_create: hipMalloc failed"; rt_destroy(c); return RT_ERR_OOM; }
  (void)hipMemset(c->dCounters, 0, 8 * sizeof(unsigned long long));
  *out = c;
  return RT_OK;
}

extern "C" {

int rt_destroy(rt_ctx* c)
{
  if(!c) return RT_ERR_INVALID_ARG;
  (void)hipSetDevice(c->device);
  (void)syncAll(c);
  freePool(c->sceneAllocs); freePool(c->accelAllocs); freePool(c->scratchAllocs); freePool(c->ovfAllocs);
  for(int i = 0; i < RT_BUF_COUNT; i++) if(c->bufs[i]) (void)hipFree(c->bufs[i]);
  for(void* p : {c->spareG, c->spareMotion, c->spareG2, c->spareMotion2, c->spareDirRes}) if(p) (void)hipFree(p);
  for(void* p : c->indA) if(p && p != c->bufs[RT_BUF_DENOISE_IND_A]) (void)hipFree(p);
  for(hipStream_t q : c->padStreams) if(q) (void)hipStreamDestroy(q);
  for(hipStream_t& q : c->indStreams) { if(q) (void)hipStreamDestroy(q); q = nullptr; }
  for(hipStream_t& q : c->sideStreams) { if(q) (void)hipStreamDestroy(q); q = nullptr; }
  c->indStream = c->sideStream = nullptr;
  if(c->indStream) (void)hipStreamDestroy(c->indStream);
  for(int i = 0; i < 4; i++) { if(c->evD[i]) (void)hipEventDestroy(c->evD[i]); if(c->evI[i]) (void)hipEventDestroy(c->evI[i]); if(c->evDone[i]) (void)hipEventDestroy(c->evDone[i]); }
  if(c->dCounters) (void)hipFree(c->dCounters);
  if(c->dSky) (void)hipFree(c->dSky);
  if(c->dPick) (void)hipFree(c->dPick);
  for(auto& E : c->evSets) for(int i = 0; i < rt_ctx::MAX_EV; i++) (void)hipEventDestroy(E.ev[i]);
  if(c->ownStream) (void)hipStreamDestroy(c->ownStream);
  if(c->sideStream) (void)hipStreamDestroy(c->sideStream);
  if(c->evFork) (void)hipEventDestroy(c->evFork);
  if(c->evJoin) (void)hipEventDestroy(c->evJoin);
  delete c;
  if(g_liveCtx.fetch_sub(1) == 1) { std::lock_guard<std::mutex> one(g_accelMutex); g_accelCache.reset(); }   // the last context takes the cached host build with it
  return RT_OK;
}

int rt_set_stream(rt_ctx* c, void* s)
{
  if(!c) return RT_ERR_INVALID_ARG;
  c->stream = s ? static_cast<hipStream_t>(s) : c->ownStream;
  return RT_OK;
}

int rt_sync(rt_ctx* c)
{
  if(!c) return RT_ERR_INVALID_ARG;
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  return RT_OK;
}

int rt_upload_scene(rt_ctx* c, const rt_scene_desc* d)
{
  if(!c) return RT_ERR_INVALID_ARG;
  if(!d || !d->primMeshes || !d->vertices || !d->indices || !d->instances || !d->materials || d->numMaterials == 0)
    return fail(c, RT_ERR_INVALID_ARG, "rt_upload_scene: missing geometry/material arrays");
  for(uint32_t i = 0; i < d->numInstances; i++)
    if(d->instances[i].primMesh >= d->numPrimMeshes) return fail(c, RT_ERR_INVALID_ARG, "rt_upload_scene: instance references a missing prim mesh");
  for(uint32_t i = 0; i < d->numPrimMeshes; i++) {
    const rt_prim_mesh& pm = d->primMeshes[i];
    if(uint64_t(pm.firstIndex) + pm.indexCount > d->numIndices || uint64_t(pm.vertexOffset) + pm.vertexCount > d->numVertices || pm.indexCount % 3 != 0)
      return fail(c, RT_ERR_INVALID_ARG, "rt_upload_scene: prim mesh range out of bounds");
    if(pm.materialIndex >= int32_t(d->numMaterials)) return fail(c, RT_ERR_INVALID_ARG, "rt_upload_scene: material index out of range");
    for(uint32_t k = 0; k < pm.indexCount; k++)
      if(d->indices[pm.firstIndex + k] >= pm.vertexCount) return fail(c, RT_ERR_INVALID_ARG, "rt_upload_scene: vertex index beyond the prim mesh's vertex range");
  }
  // everything the kernels use as an array index is checked here: a bad id must come back as an error, not as a GPU fault
  const int32_t nTex = d->textures ? int32_t(d->numTextures) : 0;
  for(uint32_t i = 0; i < d->numMaterials; i++) {
    const rt_material& m = d->materials[i];
    const int32_t ids[5] = {m.pbrBaseColorTexture, m.pbrMetallicRoughnessTexture, m.emissiveTexture, m.normalTexture, m.transmissionTexture};
    for(int32_t id : ids)
      if(id < -1 || id >= nTex) return fail(c, RT_ERR_INVALID_ARG, "rt_upload_scene: material references a missing texture");
  }
  if(d->trigLights)
    for(uint32_t i = 0; i < d->lightInfo.trigLightSize; i++)
      if(d->trigLights[i].matIndex >= d->numMaterials || d->trigLights[i].impSamp.alias < 0 || uint32_t(d->trigLights[i].impSamp.alias) >= d->lightInfo.trigLightSize)
        return fail(c, RT_ERR_INVALID_ARG, "rt_upload_scene: triangle light with a bad material or alias index");
  if(d->puncLights)
    for(uint32_t i = 0; i < d->lightInfo.puncLightSize; i++)
      if(d->puncLights[i].impSamp.alias < 0 || uint32_t(d->puncLights[i].impSamp.alias) >= d->lightInfo.puncLightSize)
        return fail(c, RT_ERR_INVALID_ARG, "rt_upload_scene: punctual light with a bad alias index");
  if(d->envRgba32f && d->envAccel && d->envWidth > 0 && d->envHeight > 0) {
    const int64_t n = int64_t(d->envWidth) * d->envHeight;
    for(int64_t i = 0; i < n; i++)
      if(d->envAccel[i].alias < 0 || d->envAccel[i].alias >= n) return fail(c, RT_ERR_INVALID_ARG, "rt_upload_scene: environment alias table entry out of range");
  }
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  LoadTimer lt;
  freePool(c->sceneAllocs); freePool(c->accelAllocs);
  c->haveScene = c->haveAccel = false;
  c->ds = DevScene{};
  c->ds.sky = (c->sunAndSky.in_use == 1) ? static_cast<const SkyPre*>(c->dSky) : nullptr;
  c->primMeshes.assign(d->primMeshes, d->primMeshes + d->numPrimMeshes);
  c->vertices.assign(d->vertices, d->vertices + d->numVertices);
  c->indices.assign(d->indices, d->indices + d->numIndices);
  c->instances.assign(d->instances, d->instances + d->numInstances);
  c->materials.assign(d->materials, d->materials + d->numMaterials);

  int rc;
  if((rc = upload(c, c->sceneAllocs, d->primMeshes, d->numPrimMeshes, &c->ds.primMeshes))) return rc;
  if((rc = upload(c, c->sceneAllocs, d->vertices, size_t(d->numVertices), &c->ds.vertices))) return rc;
  if((rc = upload(c, c->sceneAllocs, d->indices, size_t(d->numIndices), &c->ds.indices))) return rc;
  if((rc = upload(c, c->sceneAllocs, d->materials, d->numMaterials, &c->ds.materials))) return rc;
  if((rc = upload(c, c->sceneAllocs, d->puncLights, d->puncLights ? d->lightInfo.puncLightSize : 0, &c->ds.puncLights))) return rc;
  if((rc = upload(c, c->sceneAllocs, d->trigLights, d->trigLights ? d->lightInfo.trigLightSize : 0, &c->ds.trigLights))) return rc;
  c->ds.lightInfo = d->lightInfo;
  if(!d->puncLights) c->ds.lightInfo.puncLightSize = 0;
  if(!d->trigLights) c->ds.lightInfo.trigLightSize = 0;
  // textures (BGRA8, LOD 0)
  lt.lap("geometry copies + uploads");
  c->hostAlpha.clear();
  std::vector<const uint8_t*> alphaSrc;
  std::vector<DevTexture> texs(std::max<uint32_t>(d->numTextures, 1));
  static const uint8_t white[4] = {255, 255, 255, 255};
  for(uint32_t i = 0; i < uint32_t(texs.size()); i++) {
    rt_texture t{white, 1, 1, RT_WRAP_REPEAT, RT_WRAP_REPEAT, RT_FILTER_LINEAR, 0};
    if(i < d->numTextures && d->textures && d->textures[i].bgra8 && d->textures[i].width > 0 && d->textures[i].height > 0) t = d->textures[i];
    const uint8_t* dp = nullptr;
    if((rc = upload(c, c->sceneAllocs, t.bgra8, size_t(t.width) * t.height * 4, &dp))) return rc;
    texs[i] = DevTexture{dp, t.width, t.height, t.wrapS, t.wrapT, t.magFilter, 0};
    c->hostAlpha.emplace_back(size_t(t.width) * t.height);
    alphaSrc.push_back(t.bgra8);
  }
  lt.lap("texture uploads");
  {  // the alpha planes (opacity micro-map build) and their hashes (rt_build_accel's cache key), one task per texture on the host's cores: 0.87 G byte-strided
     // copies + a hash of them took 1.1 s of the headline scene's load on one core (round 6)
    c->alphaHash.assign(c->hostAlpha.size(), {0ull, 0ull});
    std::atomic<size_t> next{0};
    auto work = [&] {
      for(;;) {
        const size_t i = next.fetch_add(1);
        if(i >= c->hostAlpha.size()) return;
        std::vector<uint8_t>& a = c->hostAlpha[i];
        const uint8_t* src = alphaSrc[i];
        for(size_t k = 0; k < a.size(); k++) a[k] = src[k * 4 + 3];
        uint64_t h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull;
        hashBytes(a.data(), a.size(), h0, h1);
        c->alphaHash[i] = {h0, h1};
      }
    };
    const size_t nt = std::min<size_t>(c->hostAlpha.size(), size_t(rt_cpu_budget()));
    std::vector<std::thread> pool;
    for(size_t k = 1; k < nt; k++) pool.emplace_back(work);
    work();
    for(auto& th : pool) th.join();
  }
  lt.lap("alpha planes + hashes");
  if((rc = upload(c, c->sceneAllocs, texs.data(), texs.size(), &c->ds.textures))) return rc;
  c->devTextures = texs;
  // environment
  if(d->envRgba32f && d->envWidth > 0 && d->envHeight > 0) {
    const size_t n = size_t(d->envWidth) * d->envHeight;
    if((rc = upload(c, c->sceneAllocs, d->envRgba32f, n * 4, &c->ds.env))) return rc;
    if(d->envAccel) { if((rc = upload(c, c->sceneAllocs, d->envAccel, n, &c->ds.envAccel))) return rc; }
    else {
      std::vector<rt_impt_samp> a(n, rt_impt_samp{0, 1.0f, 0.0f, 0.0f});
      if((rc = upload(c, c->sceneAllocs, a.data(), n, &c->ds.envAccel))) return rc;
    }
    c->ds.envW = d->envWidth; c->ds.envH = d->envHeight;
  } else {
    const float black[4] = {0, 0, 0, 0};
    const rt_impt_samp one{0, 1.0f, 0.0f, 0.0f};
    if((rc = upload(c, c->sceneAllocs, black, 4, &c->ds.env))) return rc;
    if((rc = upload(c, c->sceneAllocs, &one, 1, &c->ds.envAccel))) return rc;
    c->ds.envW = c->ds.envH = 1;
  }
  RT_HIP(c, hipDeviceSynchronize());
  c->haveScene = true;
  reopenPriorityDecision(c);   // (the context is drained: syncAll above / hipDeviceSynchronize below)
  return RT_OK;
}


static void hashBytes(const void* p, size_t n, uint64_t& h0, uint64_t& h1)
{
  const unsigned char* b = static_cast<const unsigned char*>(p);
  size_t i = 0;
  for(; i + 8 <= n; i += 8) {
    uint64_t w; memcpy(&w, b + i, 8);
    h0 = (h0 ^ w) * 0x9E3779B97F4A7C15ull; h0 ^= h0 >> 29;
    h1 = (h1 + w) * 0xC2B2AE3D27D4EB4Full; h1 ^= h1 >> 31;
  }
  uint64_t w = 0; memcpy(&w, b + i, n - i); w |= uint64_t(n) << 56;
  h0 = (h0 ^ w) * 0x9E3779B97F4A7C15ull; h0 ^= h0 >> 29;
  h1 = (h1 + w) * 0xC2B2AE3D27D4EB4Full; h1 ^= h1 >> 31;
}
#define RT_HASH_VEC(v) do { const uint64_t n_ = (v).size(); hashBytes(&n_, 8, h0, h1); if(n_) hashBytes((v).data(), n_ * sizeof((v)[0]), h0, h1); } while(0)
static void sceneKey(const rt_ctx* c, uint64_t& h0, uint64_t& h1)
{
  h0 = 0x243F6A8885A308D3ull; h1 = 0x13198A2E03707344ull;
  RT_HASH_VEC(c->primMeshes); RT_HASH_VEC(c->vertices); RT_HASH_VEC(c->indices); RT_HASH_VEC(c->instances); RT_HASH_VEC(c->materials);
  for(size_t i = 0; i < c->devTextures.size(); i++) {
    const DevTexture& t = c->devTextures[i];
    const int meta[5] = {t.w, t.h, t.wrapS, t.wrapT, t.filter};
    hashBytes(meta, sizeof(meta), h0, h1);
    if(i < c->alphaHash.size()) { const uint64_t n_ = c->hostAlpha[i].size(); hashBytes(&n_, 8, h0, h1); hashBytes(c->alphaHash[i].data(), 16, h0, h1); }   // (hashed per texture at upload, in parallel)
  }
  // the builder's settings are part of what is cached: a host that changes RESTIR_BVH_* between two contexts of one process gets the tree it asked for, not the first
  // one's (advisor finding of round 5)
  for(const char* name : {"RESTIR_BVH_SPLIT", "RESTIR_BVH_SPLIT_BUDGET", "RESTIR_BVH_SPLIT_ALPHA", "RESTIR_BVH_SPLIT_WORK", "RESTIR_BVH_BUDGET_RULE", "RESTIR_BVH_ROTATE", "RESTIR_BVH_ROTATE_GG",
                           "RESTIR_BVH_REINSERT", "RESTIR_BVH_BINS", "RESTIR_BVH_SBINS", "RESTIR_BVH_SLOTCOST", "RESTIR_BVH_COLLAPSE", "RESTIR_BVH_SLOTS", "RESTIR_BVH_PAR_MIN", "RESTIR_BVH_SEQ_MAX"}) {
    const char* v = getenv(name);
    hashBytes(name, strlen(name), h0, h1);
    if(v) hashBytes(v, strlen(v) + 1, h0, h1);
  }
}
#undef RT_HASH_VEC
static int buildHostAccel(rt_ctx* c, HostAccel& out)
{
  LoadTimer lt;
  rt_scene_desc d{};
  d.numPrimMeshes = uint32_t(c->primMeshes.size()); d.primMeshes = c->primMeshes.data();
  d.numVertices = c->vertices.size(); d.vertices = c->vertices.data();
  d.numIndices = c->indices.size(); d.indices = c->indices.data();
  d.numInstances = uint32_t(c->instances.size()); d.instances = c->instances.data();
  BuildOutput& bo = out.bo;
  int threads = rt_cpu_budget();   // (the cgroup quota, not the machine: include/rt_cpus.h)
  if(!buildBvh8(d, bo, threads > 0 ? threads : 1)) return fail(c, RT_ERR_INVALID_ARG, "rt_build_accel: BVH8 build failed");
  if(bo.maxDepth > STACK_MAX) {
    // rotations and spatial splits can deepen a tree: before giving up, the tree of rounds 1-4 (object splits only), which would have built (advisor finding of round 5)
    bo = BuildOutput{};
    if(!buildBvh8(d, bo, threads > 0 ? threads : 1, true)) return fail(c, RT_ERR_INVALID_ARG, "rt_build_accel: BVH8 build failed");
  }
  if(bo.maxDepth > STACK_MAX) return fail(c, RT_ERR_INVALID_ARG, "rt_build_accel: BVH8 deeper than the traversal stack");
  lt.lap("buildBvh8");
  // alpha records for the triangles that go through HitTest (instances without FORCE_OPAQUE)
  std::vector<AlphaRec>& alpha = out.alpha; alpha.assign(1, AlphaRec{});
  std::vector<int32_t>& alphaTex = out.alphaTex; alphaTex.assign(1, -1);
  // (a hash map since round 6: 2 M lookups with 48-byte keys in a std::map were a second of the headline scene's load)
  struct KeyHash { size_t operator()(const std::array<uint32_t, 12>& k) const { uint64_t h = 1469598103934665603ull; for(uint32_t w : k) { h ^= w; h *= 1099511628211ull; } return size_t(h ^ (h >> 29)); } };
  std::unordered_map<std::array<uint32_t, 12>, std::array<uint32_t, 4>, KeyHash> ommCache;
  ommCache.reserve(1 << 16);
  std::vector<uint32_t> alphaOf;   // by globalId: the record of a triangle that has several references (spatial splits) is made once
  if(bo.tris.size() > bo.triRef.size()) alphaOf.assign(bo.triRef.size(), 0u);
  std::vector<std::array<uint32_t, 4>> ommOf;
  for(Tri48& T : bo.tris) {
    if(T.flags & TRI_OPAQUE) continue;
    if(!alphaOf.empty() && alphaOf[T.globalId]) { T.alphaIdx = alphaOf[T.globalId]; memcpy(T.omm, ommOf[T.alphaIdx].data(), sizeof(T.omm)); continue; }
    const TriRef ref = bo.triRef[T.globalId];
    const rt_prim_mesh& pm = c->primMeshes[c->instances[ref.inst].primMesh];
    const rt_material& m = c->materials[size_t(pm.materialIndex > 0 ? pm.materialIndex : 0)];
    const uint32_t* ix = &c->indices[pm.firstIndex + 3 * ref.prim];
    AlphaRec a{};
    int32_t tex = -1;
    const rt_vec2 t0 = c->vertices[pm.vertexOffset + ix[0]].texcoord, t1 = c->vertices[pm.vertexOffset + ix[1]].texcoord, t2 = c->vertices[pm.vertexOffset + ix[2]].texcoord;
    a.uv0x = t0.x; a.uv0y = t0.y; a.uv1x = t1.x; a.uv1y = t1.y; a.uv2x = t2.x; a.uv2y = t2.y;
    a.baseAlpha = m.pbrBaseColorFactor.w; a.cutoff = m.alphaCutoff; a.alphaMode = m.alphaMode;
    if(m.pbrBaseColorTexture > -1 && size_t(m.pbrBaseColorTexture) < c->devTextures.size()) {
      const DevTexture& t = c->devTextures[size_t(m.pbrBaseColorTexture)];
      tex = m.pbrBaseColorTexture; a.w = t.w; a.h = t.h; a.wrapS = t.wrapS; a.wrapT = t.wrapT; a.filter = t.filter;
    }
    T.alphaIdx = uint32_t(alpha.size());
    alpha.push_back(a); alphaTex.push_back(tex);
    // opacity micro-map, cached per distinct (texture, uv triple, alpha parameters)
    std::array<uint32_t, 12> key{};
    memcpy(key.data(), &a.uv0x, 8 * sizeof(float));
    key[8] = uint32_t(m.pbrBaseColorTexture); key[9] = uint32_t(a.alphaMode); key[10] = uint32_t(a.wrapS) ^ (uint32_t(a.wrapT) << 16); key[11] = uint32_t(a.filter);
    auto it = ommCache.find(key);
    if(it == ommCache.end()) {
      std::array<uint32_t, 4> o{};
      const std::vector<uint8_t>* al = (tex >= 0 && size_t(tex) < c->hostAlpha.size()) ? &c->hostAlpha[size_t(m.pbrBaseColorTexture)] : nullptr;
      buildOpacityMap(a, al, o.data());
      it = ommCache.emplace(key, o).first;
    }
    memcpy(T.omm, it->second.data(), sizeof(T.omm));
    if(!alphaOf.empty()) { alphaOf[T.globalId] = T.alphaIdx; ommOf.resize(alpha.size()); ommOf[T.alphaIdx] = it->second; }
  }
  lt.lap("alpha records + micro-maps");
  return RT_OK;
}

int rt_build_accel(rt_ctx* c)
{
  if(!c) return RT_ERR_INVALID_ARG;
  if(!c->haveScene) return fail(c, RT_ERR_NO_SCENE, "rt_build_accel: no scene uploaded");
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  freePool(c->accelAllocs);
  c->haveAccel = false;
  // Host products (BVH8, alpha records, opacity micro-maps): built once per distinct scene in this process and shared by every context that uploads
  // the same scene — the N ranks of an rt_mgpu context, or an application's contexts on several devices (1.4-1.6 s per build at 2.8 M triangles).
  std::shared_ptr<const HostAccel> ha;
  LoadTimer lt;
  {
    std::lock_guard<std::mutex> one(g_accelMutex);   // also: one multi-threaded host build at a time
    uint64_t k0, k1;
    sceneKey(c, k0, k1);
    lt.lap("scene key");
    if(g_accelCache && g_accelCache->key0 == k0 && g_accelCache->key1 == k1) ha = g_accelCache;
    else {
      auto fresh = std::make_shared<HostAccel>();
      fresh->key0 = k0; fresh->key1 = k1;
      const int rc = buildHostAccel(c, *fresh);
      if(rc) return rc;
      ha = fresh; g_accelCache = ha;
    }
  }
  lt.lap("host products (build or cache)");
  const BuildOutput& bo = ha->bo;
  int rc;
  {
    std::vector<AlphaRec> alpha = ha->alpha;   // texture addresses are per device
    for(size_t i = 0; i < alpha.size(); i++) if(ha->alphaTex[i] >= 0) alpha[i].bgra = c->devTextures[size_t(ha->alphaTex[i])].bgra;
    if((rc = upload(c, c->accelAllocs, alpha.data(), alpha.size(), &c->ds.alphaRec))) return rc;
    // the latency build's copy, addressable without the triangle record (one dependent access less per alpha candidate); 64 B per triangle
    std::vector<AlphaRec> byTri(bo.tris.size(), AlphaRec{});
    for(size_t i = 0; i < bo.tris.size(); i++) if(!(bo.tris[i].flags & TRI_OPAQUE)) byTri[i] = alpha[bo.tris[i].alphaIdx];
    if((rc = upload(c, c->accelAllocs, byTri.data(), byTri.size(), &c->ds.alphaByTri))) return rc;
  }
  if((rc = upload(c, c->accelAllocs, bo.nodes.data(), bo.nodes.size(), &c->ds.nodes))) return rc;
  if((rc = upload(c, c->accelAllocs, bo.tris.data(), bo.tris.size(), &c->ds.tris))) return rc;
  if((rc = upload(c, c->accelAllocs, bo.triRef.data(), bo.triRef.size(), &c->ds.triRef))) return rc;
  if((rc = upload(c, c->accelAllocs, bo.instances.data(), bo.instances.size(), &c->ds.instances))) return rc;
  c->ds.numNodes = uint32_t(bo.nodes.size()); c->ds.numTris = uint32_t(bo.tris.size());
  // traversal stack: what a ray of this tree can need.  How much of it each lane keeps in LDS is a per-launch choice (stackLdsEntries below): with frames
  // in flight 6 entries = 3 KB per wave (measured, profiles/r02_short_stack_ab.txt), deeper entries in a per-thread area in HBM (ensureStackOverflow);
  // serial schedules have the LDS to themselves and keep the whole stack there.  RESTIR_STACK_LDS=<n> forces n entries in every schedule.
  c->ds.stackTotal = std::max(8, ((bo.maxDepth + 1 + 3) / 4) * 4);
  c->ds.stackEntries = c->ds.stackTotal;   // the launchers shorten it per launch (DevFrame::stackLds)
  c->ds.triPad = bo.pad;
  { const char* e = getenv("RESTIR_COOP"); c->ds.coopLive = e ? std::max(0, std::min(64, atoi(e))) : 4; }
  { const char* e = getenv("RESTIR_GANG"); c->ds.gangMax = e ? std::max(0, std::min(7, atoi(e))) : 4; }   // latency build: gang mode for the last rays of a wave (traverse.h)
  c->numNodes = bo.nodes.size(); c->numTris = bo.triRef.size(); c->maxDepth = bo.maxDepth;
  c->numRefs = bo.tris.size(); c->spatialSplits = size_t(bo.spatialSplits); c->sahNodeSteps = bo.sahNodeSteps; c->sahTriSteps = bo.sahTriSteps;
  RT_HIP(c, hipDeviceSynchronize());
  lt.lap("uploads (tree, alpha records)");
  c->haveAccel = true;
  reopenPriorityDecision(c);
  return ensureStackOverflow(c);
}

int rt_resize(rt_ctx* c, int w, int h)
{
  if(!c) return RT_ERR_INVALID_ARG;
  if(w <= 0 || h <= 0 || w > 32767 || h > 32767) return fail(c, RT_ERR_INVALID_ARG, "rt_resize: size must be in 1..32767 (RG16_SINT motion vectors)");
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  reopenPriorityDecision(c);
  for(void*& p : c->indA) { if(p && p != c->bufs[RT_BUF_DENOISE_IND_A]) (void)hipFree(p); p = nullptr; }
  for(int i = 0; i < RT_BUF_COUNT; i++) { if(c->bufs[i]) (void)hipFree(c->bufs[i]); c->bufs[i] = nullptr; c->bufBytes[i] = 0; }
  for(void** p : {&c->spareG, &c->spareMotion, &c->spareG2, &c->spareMotion2, &c->spareDirRes}) if(*p) { (void)hipFree(*p); *p = nullptr; }
  c->W = c->H = 0;
  const size_t n = size_t(w) * h, nh = size_t(w / 2) * (h / 2);
  for(int i = 0; i < RT_BUF_COUNT; i++) {
    const size_t bytes = (halfRes(i) ? nh : n) * elemBytes(i);
    const size_t alloc = bytes + (halfRes(i) ? size_t(w / 2) * (RT_PAD_ROWS / 2) : size_t(w) * RT_PAD_ROWS) * elemBytes(i) + 16;
    RT_HIP(c, hipMalloc(&c->bufs[i], alloc));
    RT_HIP(c, hipMemset(c->bufs[i], (i == RT_BUF_LIGHT_ID0 || i == RT_BUF_LIGHT_ID1) ? 0xff : 0, alloc));
    c->bufBytes[i] = bytes; c->bufAlloc[i] = alloc;
    if(i == RT_BUF_DENOISE_IND_A) {  // the second parity of the noisy indirect colour
      c->indA[0] = c->bufs[i];
      RT_HIP(c, hipMalloc(&c->indA[1], alloc));
      RT_HIP(c, hipMemset(c->indA[1], 0, alloc));
    }
    if(i == RT_BUF_GBUFFER0 || i == RT_BUF_MOTION) {  // rotation partners for frames in flight (rt_render_frame, overlap 2)
      for(void** spare : {(i == RT_BUF_MOTION) ? &c->spareMotion : &c->spareG, (i == RT_BUF_MOTION) ? &c->spareMotion2 : &c->spareG2}) {
        RT_HIP(c, hipMalloc(spare, alloc));
        RT_HIP(c, hipMemset(*spare, 0, alloc));
      }
    }
    if(i == RT_BUF_DIRECT_RESULT0) {   // overlap 3: the direct image of frame f-3 is the one direct(f) overwrites
      RT_HIP(c, hipMalloc(&c->spareDirRes, alloc));
      RT_HIP(c, hipMemset(c->spareDirRes, 0, alloc));
    }
  }
  // internal scratch
  freePool(c->scratchAllocs);
  c->scratch = DevFrame{};
  auto alloc = [&](size_t bytes, void** out) -> int {
    RT_HIP(c, hipMalloc(out, std::max<size_t>(bytes, 256)));
    c->scratchAllocs.push_back(*out);
    RT_HIP(c, hipMemset(*out, 0, std::max<size_t>(bytes, 256)));
    return RT_OK;
  };
  DevFrame& X = c->scratch;
  int rc;
#define RT_SCRATCH(field, count, T) if((rc = alloc(size_t(count) * sizeof(T), reinterpret_cast<void**>(&X.field)))) return rc
  RT_SCRATCH(surf, n, SurfRec); RT_SCRATCH(status, n, uint32_t); RT_SCRATCH(qcount, 256, uint32_t);
  RT_SCRATCH(geomN, n, float4); RT_SCRATCH(geomP, n, float4); RT_SCRATCH(geomNh, nh, float4); RT_SCRATCH(geomPh, nh, float4);
  RT_SCRATCH(postRowSums, size_t(h) * 6, double); RT_SCRATCH(postMean, 8, float);
  RT_SCRATCH(postMipD, n + 64, float4); RT_SCRATCH(postMipI, n + 64, float4);   // levels 1..7 (n/3 texels; up to n for one-pixel-wide images)
  RT_SCRATCH(rowCost, 4096 + 64, uint32_t); RT_SCRATCH(rowOrder, 4096 + 64, uint16_t);
  RT_SCRATCH(tileOrder, (size_t(w / 2 + 7) / 8) * (size_t(h / 2 + 7) / 8 + 16 * 2) + 64 + 4096, uint32_t);   // 8 per-XCD lists: tiles + one chunk of slack each
#if RT_WAVEPROF
  RT_SCRATCH(waveProf, WAVEPROF_RECORDS * 16, uint32_t);
#endif
#undef RT_SCRATCH
  RT_HIP(c, hipDeviceSynchronize());  // memsets above ran on the null stream; the ctx stream does not wait for it implicitly
  c->W = w; c->H = h;
  return ensureStackOverflow(c);
}

// Overflow part of the traversal stacks (DevScene::stackOvf): (stackTotal - stackEntries) entries for every thread of the largest traced launch of
// this frame size, one area per stage kind.  Only schedules that shorten the LDS stack touch them (frames in flight, RESTIR_STACK_LDS), so they are
// allocated when such a schedule is selected and (re)sized when the tree or the frame size changes; a serial ctx (every rt_mgpu rank of the barrier
// schedule) holds none.  Callers have drained the ctx.
static int ensureStackOverflow(rt_ctx* c)
{
  freePool(c->ovfAllocs);
  c->ds.stackOvf = c->ds.stackOvfInd = nullptr; c->ds.stackOvfThreads = 0;
  if(!(stackLdsEnv() || c->overlap >= 2)) return RT_OK;
  if(!c->haveAccel || c->W <= 0 || c->ds.stackTotal <= stackLdsMin()) return RT_OK;
  const size_t tilesX = size_t(c->W + 7) / 8, tilesY = size_t(c->H + 7) / 8;
  const size_t blocks = std::max<size_t>(16384, tilesX * tilesY + 8 * tilesX + 1024);   // >= any traced grid: tileGrid() of the full frame; 4 waves per half-res tile; persistent launches
  const size_t bytes = blocks * 64 * size_t(c->ds.stackTotal - stackLdsMin()) * sizeof(uint2);
  void* p[2] = {nullptr, nullptr};
  for(int i = 0; i < 2; i++) { RT_HIP(c, hipMalloc(&p[i], bytes)); c->ovfAllocs.push_back(p[i]); }
  c->ds.stackOvf = static_cast<uint2*>(p[0]); c->ds.stackOvfInd = static_cast<uint2*>(p[1]);
  c->ds.stackOvfThreads = uint32_t(blocks * 64);
  return RT_OK;
}

int rt_set_camera(rt_ctx* c, const rt_scene_camera* cam)
{
  if(!c || !cam) return RT_ERR_INVALID_ARG;
  c->cam = *cam;
  return RT_OK;
}

static void selectFrame(rt_ctx* c, int frames) { if(c->indA[0]) c->bufs[RT_BUF_DENOISE_IND_A] = c->indA[frames & 1]; }

// The two extra streams of rt_render_frame's overlapped schedules, created on first use: a host that drives the stages itself (rt_run_stage on its own streams:
// rt_mgpu, tiled.py) never needs them, and every stream of a process takes one of the device's few hardware queues (4 by default) out of the rotation.
// Stream priorities of the frames-in-flight schedule: prioSpec() above (RESTIR_PRIO).  Default: indirect stream high (measured 2 % faster than indirect + filter
// streams high on the lite config 4, round 2).
static hipError_t ensureOverlapStreams(rt_ctx* c)
{
  if(c->sideStream && c->indStream) return hipSuccess;
  hipError_t e = hipSuccess;
  hipStream_t& side = c->sideStreams[c->prio[2] + 1];
  hipStream_t& ind = c->indStreams[c->prio[1] + 1];
  // probe (round 5, scripts/r05_stream_pad_probe.sh): RESTIR_STREAM_PAD="a,b[,level]" creates a / b idle streams before the filter / the indirect stream — how a stream's
  // position in the process's creation order maps to the hardware queue it shares
  static int padA = -1, padB = 0, padL = 0;
  if(padA < 0) { padA = 0; if(const char* pe = getenv("RESTIR_STREAM_PAD")) (void)sscanf(pe, "%d,%d,%d", &padA, &padB, &padL); }
  auto pad = [&](int n) { for(int i = 0; i < n; i++) { hipStream_t d = nullptr; (void)createStreamLevel(&d, 2, padL); c->padStreams.push_back(d); } };
  if(!side) { pad(padA); e = createStreamLevel(&side, 2, c->prio[2]); if(e == hipSuccess) c->sideIdx[c->prio[2] + 1] = g_streamsCreated.load() - 1; }
  if(e == hipSuccess && !ind) { pad(padB); e = createStreamLevel(&ind, 1, c->prio[1]); if(e == hipSuccess) c->indIdx[c->prio[1] + 1] = g_streamsCreated.load() - 1; }
  c->sideStream = side; c->indStream = ind;
  return e;
}

// The rule of rt_render_frame's probe frames (see there).  RESTIR_PRIO_PROBE = number of probe frames (default 3; 1 = round 5's first-frame decision).
static constexpr float PRIO_FILTER_SHARE = 0.30f;   // (restir_amd/renderer.py mirrors it for reports)
static int prioProbeFrames() { static const int n = getenv("RESTIR_PRIO_PROBE") ? std::max(1, atoi(getenv("RESTIR_PRIO_PROBE"))) : 3; return n; }
// what the decision was taken on has changed (target size, scene, tree, denoise toggle): the next frames probe again.  The streams of the old levels stay alive, idle
// (one per role and level, never destroyed: see rt_ctx::indStreams); the caller has drained the context.
static void reopenPriorityDecision(rt_ctx* c)
{
  if(c->prioExplicit || c->prioFromEnv) return;
  c->prioDecided = false; c->filterShare = -1.f; c->probeFrames = 0; c->denoiseSeen = -1;
  c->prio[1] = 1; c->prio[2] = 0;
  c->indStream = c->sideStream = nullptr;
}

static DevFrame makeFrame(rt_ctx* c, int frames)
{
  selectFrame(c, frames);
  const int cur = frames & 1, last = (frames + 1) & 1;  // m_descSet[(frames+1)%2]: this = [!i] (renderer.cpp:157, 346-356)
  DevFrame F{};
  F.stackLds = stackLdsEnv() ? stackLdsEnv() : (c->overlap >= 2 ? 6 : 0);
  F.thisG = static_cast<uint4*>(c->bufs[RT_BUF_GBUFFER0 + cur]); F.lastG = static_cast<const uint4*>(c->bufs[RT_BUF_GBUFFER0 + last]);
  F.motion = static_cast<short2*>(c->bufs[RT_BUF_MOTION]);
  F.thisDirectResv = static_cast<rt_direct_reservoir*>(c->bufs[RT_BUF_DIRECT_RESV0 + cur]);
  F.lastDirectResv = static_cast<const rt_direct_reservoir*>(c->bufs[RT_BUF_DIRECT_RESV0 + last]);
  F.thisIndirectResv = static_cast<rt_indirect_reservoir*>(c->bufs[RT_BUF_INDIRECT_RESV0 + cur]);
  F.lastIndirectResv = static_cast<const rt_indirect_reservoir*>(c->bufs[RT_BUF_INDIRECT_RESV0 + last]);
  F.tempDirectResv = static_cast<rt_direct_reservoir*>(c->bufs[RT_BUF_DIRECT_RESV_TEMP]);
  F.thisLightId = static_cast<uint32_t*>(c->bufs[RT_BUF_LIGHT_ID0 + cur]); F.lastLightId = static_cast<const uint32_t*>(c->bufs[RT_BUF_LIGHT_ID0 + last]);
  F.thisDirectResult = static_cast<float4*>(c->bufs[RT_BUF_DIRECT_RESULT0 + cur]);
  F.thisIndirectResult = static_cast<float4*>(c->bufs[RT_BUF_INDIRECT_RESULT0 + cur]);
  F.denoiseDirA = static_cast<float4*>(c->bufs[RT_BUF_DENOISE_DIR_A]); F.denoiseDirB = static_cast<float4*>(c->bufs[RT_BUF_DENOISE_DIR_B]);
  F.denoiseIndA = static_cast<float4*>(c->bufs[RT_BUF_DENOISE_IND_A]); F.denoiseIndB = static_cast<float4*>(c->bufs[RT_BUF_DENOISE_IND_B]);
  F.counters = c->counting ? c->dCounters : nullptr;
  F.W = c->W; F.H = c->H;
  const DevFrame& X = c->scratch;
  F.surf = X.surf; F.status = X.status; F.qcount = X.qcount; F.waveProf = X.waveProf;
  F.histRow0 = c->histRow0; F.histRow1 = c->histRow1; F.histMiss = X.qcount + 250;
  // (measured, profiles/r05_row_order_ab.txt: -1.1 % in flight on the real scene and on config 3, +0.9 % on lite and under a moving camera, the stage ALONE 2-12 % slower —
  //  the heavy rows then all run at once instead of between cheap ones.  Off unless RESTIR_ROW_ORDER=1.)
  { static const bool rowOrderOn = getenv("RESTIR_ROW_ORDER") && atoi(getenv("RESTIR_ROW_ORDER")) != 0;
    F.rowCost = rowOrderOn ? X.rowCost : nullptr; F.rowOrder = rowOrderOn ? X.rowOrder : nullptr; }
  F.geomN = X.geomN; F.geomP = X.geomP; F.geomNh = X.geomNh; F.geomPh = X.geomPh; F.tileOrder = X.tileOrder; F.postRowSums = X.postRowSums; F.postMean = X.postMean;
  return F;
}

static int checkReady(rt_ctx* c, const rt_state* st)
{
  if(!c || !st) return RT_ERR_INVALID_ARG;
  if(!c->haveScene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
  if(!c->haveAccel) return fail(c, RT_ERR_NO_ACCEL, "rt_build_accel has not been called");
  if(c->W == 0 || st->size.x != c->W || st->size.y != c->H) return fail(c, RT_ERR_NO_TARGET, "RtxState.size does not match rt_resize");
  return RT_OK;
}

int rt_run_stage(rt_ctx* c, const rt_state* st, int frames, int stage, int level, int rowBegin, int rowEnd)
{
  int rc = checkReady(c, st);
  if(rc) return rc;
  if(stage < 0 || stage >= RT_STAGE_COUNT) return fail(c, RT_ERR_INVALID_ARG, "rt_run_stage: unknown stage");
  if(rowBegin < 0 || (rowBegin & 7)) return fail(c, RT_ERR_INVALID_ARG, "rt_run_stage: rowBegin must be a non-negative multiple of 8");
  {  // levels: the filter chains have 4 / 5, the direct stage its two halves in the spatial modes, every other stage only level 0
    const bool spatial = st->ReSTIRState == RT_RESTIR_SPATIAL || st->ReSTIRState == RT_RESTIR_SPATIOTEMPORAL;
    const int maxLevel = stage == RT_STAGE_DENOISE_DIRECT ? 3 : (stage == RT_STAGE_DENOISE_INDIRECT ? 4 : ((stage == RT_STAGE_DIRECT && spatial) ? 2 : 0));
    if(level < 0 || level > maxLevel)
      return fail(c, RT_ERR_INVALID_ARG, "rt_run_stage: level out of range for this stage (RT_STAGE_DIRECT levels 1 / 2 exist in the spatial modes only)");
  }
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, joinInFlight(c));
  DevFrame F = makeFrame(c, frames);
  if(stage == RT_STAGE_INDIRECT) F.histMiss = c->scratch.qcount + 251;  // per-stage-kind flag (rt_history_miss_stage)
  // (stage kinds share no scratch: a caller may spread them over several streams, tiled.PipelinedTiledFrame does)
  const hipError_t e = stageLauncher(c, *st, stage, rowBegin, rowEnd)(c->stream, c->ds, F, *st, c->cam, stage, level, rowBegin, rowEnd);
  if(e == hipErrorInvalidValue) return fail(c, RT_ERR_INVALID_ARG, "rt_run_stage: level out of range for this stage (RT_STAGE_DIRECT levels 1 / 2 exist in the spatial modes only)");
  if(e == hipErrorInvalidConfiguration) return fail(c, RT_ERR_HIP, "rt_run_stage: the traversal-stack overflow area is missing or too small for this launch (internal sizing error)");
  RT_HIP(c, e);
  return RT_OK;
}

int rt_render_frame(rt_ctx* c, const rt_state* st, int frames)
{
  int rc = checkReady(c, st);
  if(rc) return rc;
  RT_HIP(c, hipSetDevice(c->device));
  if(c->evUsed == rt_ctx::MAX_SETS) { RT_HIP(c, syncAll(c)); harvestTimings(c); }
  if(c->evUsed == c->evSets.size()) {
    rt_ctx::EvSet E{};
    for(int i = 0; i < rt_ctx::MAX_EV; i++) RT_HIP(c, hipEventCreate(&E.ev[i]));
    c->evSets.push_back(E);
  }
  // The first PRIO_PROBE_FRAMES frames of a frames-in-flight context ("probe frames") run every stage alone on the main stream and are timed; the LAST of them — warm
  // caches, warm history, the steady state's ray lengths — decides the priorities of the two other streams BEFORE they are created (below): filter stream high as
  // well when the filter chain is a sizeable part of the frame's work.  Round 5 decided on frame 0 alone (cold history; threshold 0.14 with the real scene 10 % from
  // the edge).  Warm shares and the setting that is fastest (profiles/r06_prio_rule.txt, every cell a fresh process): real exterior scene 0.16 -> (1,0); the same scene
  // seen from above across the street 0.26 -> (1,0) by 5 %; lite 0.24 -> (1,1) by 1.6 %; interior 4K 0.41, 15 degrees off axis 0.43, config 3 0.53 -> (1,1) by 6-9 %.
  // No threshold gets lite and the elevated view both right; 0.30 errs towards the indirect stream alone, which costs 1.6 % where it is wrong (0.20 cost 4.9 %).  A camera
  // that starts at the expensive pose and moves off it (bench.py --moving-camera) is judged on its first three frames and keeps (1,0): 2.9 % slower than (1,1) once the
  // view has become cheap — a host that knows its workload says so (rt_set_stream_priorities).  The decision has to precede the streams: a stream's place in the process's
  // creation order changes what the schedule gets out of it (a setting introduced after another one's streams exist ran 15-45 % slower than in a fresh process,
  // profiles/r05_prio_by_config_ab.txt), so settings cannot be compared in place — and cannot be switched later for free either.  A new target size, a new scene / tree
  // or a denoise toggle re-opens the decision (reopenPriorityDecision); an explicit rt_set_stream_priorities / RESTIR_PRIO closes it for good.
  if(c->prioDecided && !c->prioExplicit && !c->prioFromEnv && c->denoiseSeen >= 0 && (st->denoise > 0) != (c->denoiseSeen > 0)) { RT_HIP(c, syncAll(c)); reopenPriorityDecision(c); }
  const bool decide = c->overlap >= 2 && !c->prioDecided && c->spareG && c->spareMotion;
  if(decide) { RT_HIP(c, syncAll(c)); harvestTimings(c); }
  const double tracedBefore = c->accStage[RT_STAGE_DIRECT] + c->accStage[RT_STAGE_INDIRECT];
  const double filterBefore = c->accStage[RT_STAGE_DENOISE_DIRECT] + c->accStage[RT_STAGE_DENOISE_INDIRECT] + c->accStage[RT_STAGE_COMPOSE];
  rt_ctx::EvSet& E = c->evSets[c->evUsed];
  if(c->overlap >= 1 && !decide) RT_HIP(c, ensureOverlapStreams(c));
  const bool pipelined = !decide && c->overlap >= 2 && c->sideStream && c->indStream && c->spareG && c->spareMotion;
  if(pipelined) {
    // Rotate the G-buffer (3 physical buffers) and the motion buffer (2): direct(f+1) must not overwrite what indirect(f)
    // still reads (its own G-buffer + motion, and G(f-1) for temporal reprojection).  The boundary ids keep their meaning:
    // RT_BUF_GBUFFER0 + (frames & 1) / RT_BUF_MOTION are this frame's buffers once the call returns.
    // Mode 3 (three frames in flight): one more buffer of each kind, taken oldest first, and the direct image (noisy, filtered in place, composed) three deep —
    // direct(f) then overwrites G(f-4), motion(f-3) and the direct image of f-3 and has nothing to wait for in frame f-2 (below).
    if(c->overlap >= 3 && c->spareG2 && c->spareMotion2 && c->spareDirRes) {
      void*& g = c->bufs[RT_BUF_GBUFFER0 + (frames & 1)]; void* t = g; g = c->spareG; c->spareG = c->spareG2; c->spareG2 = t;
      void*& m = c->bufs[RT_BUF_MOTION]; t = m; m = c->spareMotion; c->spareMotion = c->spareMotion2; c->spareMotion2 = t;
      std::swap(c->bufs[RT_BUF_DIRECT_RESULT0 + (frames & 1)], c->spareDirRes);
    } else {
      std::swap(c->bufs[RT_BUF_GBUFFER0 + (frames & 1)], c->spareG);
      std::swap(c->bufs[RT_BUF_MOTION], c->spareMotion);
    }
  } else {
    RT_HIP(c, joinInFlight(c));
  }
  const DevFrame F = makeFrame(c, frames);
  int k = 0, lastMain = 0, lastSide = 0, lastInd = 0;
  auto run = [&](hipStream_t strm, int stage, int level) -> int {
    hipError_t e = stageLauncher(c, *st, stage, 0, 0)(strm, c->ds, F, *st, c->cam, stage, level, 0, 0);
    if(e != hipSuccess) { c->err = std::string("launchStage: ") + hipGetErrorString(e); return RT_ERR_HIP; }
    e = hipEventRecord(E.ev[k], strm);
    if(e != hipSuccess) { c->err = std::string("hipEventRecord: ") + hipGetErrorString(e); return RT_ERR_HIP; }
    int& last = (strm == c->stream) ? lastMain : (strm == c->indStream ? lastInd : lastSide);
    E.stage[k] = stage; E.prev[k] = last; last = k; k++;
    return RT_OK;
  };
  auto mark = [&](hipStream_t strm, int& last) -> hipError_t {  // start-of-chain timestamp on a stream (after its waits)
    hipError_t e = hipEventRecord(E.ev[k], strm);
    E.stage[k] = -1; E.prev[k] = k; last = k; k++;
    return e;
  };

  if(pipelined) {
    // Frames in flight.  Three in-order streams:
    //   stream    : direct(f)                                      -> evD
    //   indStream : wait evD; indirect(f)                          -> evI
    //   sideStream: wait evD; direct A-Trous x4; wait evI; indirect A-Trous x5; compose -> evDone
    // so that direct(f+1), submitted by the next call, runs beside indirect(f) and the filters of frame f: the long
    // tail of multi-bounce tiles no longer leaves the chip idle, and the bandwidth-bound filters hide behind traversal.
    // Buffer reuse across frames is ordered explicitly:
    //   direct(f) overwrites G(f-3) [read by indirect(f-2)], motion(f-2) [indirect(f-2)] and the result image of f-2 [compose(f-2)];
    //   indirect(f) overwrites the noisy-indirect buffer of its parity, which the indirect A-Trous of f-2 read (two buffers: no wait for f-1's filters).
    // Mode 3: with the deeper rotation above the same three hazards are those of frame f-3.  Mode 2's wait closes a loop direct(f-2) -> filters(f-2) -> direct(f):
    // two periods cannot be shorter than a direct stage plus the whole filter chain of one frame, and in flight that chain is stretched to the length of a frame.
    const uint64_t s = c->seq;
    const int r = int(s & 3);
    const uint64_t depth = (c->overlap >= 3 && c->spareG2 && c->spareMotion2 && c->spareDirRes) ? 3u : 2u;
    if(s >= depth) {
      RT_HIP(c, hipStreamWaitEvent(c->stream, c->evI[(s - depth) & 3], 0));
      RT_HIP(c, hipStreamWaitEvent(c->stream, c->evDone[(s - depth) & 3], 0));
    }
    RT_HIP(c, mark(c->stream, lastMain));
    if((rc = run(c->stream, RT_STAGE_DIRECT, 0))) return rc;
    RT_HIP(c, hipEventRecord(c->evD[r], c->stream));

    RT_HIP(c, hipStreamWaitEvent(c->indStream, c->evD[r], 0));
    if(s >= 2) RT_HIP(c, hipStreamWaitEvent(c->indStream, c->evDone[(s - 2) & 3], 0));   // the noisy-indirect buffer of this parity: filtered for f-2
    RT_HIP(c, mark(c->indStream, lastInd));
    if((rc = run(c->indStream, RT_STAGE_INDIRECT, 0))) return rc;
    RT_HIP(c, hipEventRecord(c->evI[r], c->indStream));

    RT_HIP(c, hipStreamWaitEvent(c->sideStream, c->evD[r], 0));
    RT_HIP(c, mark(c->sideStream, lastSide));
    if(st->denoise > 0) for(int i = 0; i < 4; i++) if((rc = run(c->sideStream, RT_STAGE_DENOISE_DIRECT, i))) return rc;
    RT_HIP(c, hipStreamWaitEvent(c->sideStream, c->evI[r], 0));
    RT_HIP(c, mark(c->sideStream, lastSide));
    if(st->denoise > 0) for(int i = 0; i < 5; i++) if((rc = run(c->sideStream, RT_STAGE_DENOISE_INDIRECT, i))) return rc;
    if((rc = run(c->sideStream, RT_STAGE_COMPOSE, 0))) return rc;
    RT_HIP(c, hipEventRecord(c->evDone[r], c->sideStream));
    c->seq = s + 1; c->inFlight = true;
    E.count = k; E.last = lastSide;
    c->evUsed++;
    return RT_OK;
  }

  RT_HIP(c, mark(c->stream, lastMain));
  // Renderer::run, renderer.cpp:163-205.  The direct A-Trous chain only depends on the direct stage and the indirect stage
  // + its A-Trous chain only on the G-buffer, so the two chains run on two streams and join before compose: the direct
  // filter (ALU bound, full occupancy) fills the CUs that the indirect stage's long tail of multi-bounce tiles leaves idle.
  const bool fork = !decide && c->overlap && st->denoise > 0 && c->sideStream;
  if((rc = run(c->stream, RT_STAGE_DIRECT, 0))) return rc;
  if(fork) {
    RT_HIP(c, hipEventRecord(c->evFork, c->stream));
    RT_HIP(c, hipStreamWaitEvent(c->sideStream, c->evFork, 0));
    RT_HIP(c, mark(c->sideStream, lastSide));
    for(int i = 0; i < 4; i++) if((rc = run(c->sideStream, RT_STAGE_DENOISE_DIRECT, i))) return rc;
    RT_HIP(c, hipEventRecord(c->evJoin, c->sideStream));
  }
  if((rc = run(c->stream, RT_STAGE_INDIRECT, 0))) return rc;
  if(st->denoise > 0) {
    if(!fork) for(int i = 0; i < 4; i++) if((rc = run(c->stream, RT_STAGE_DENOISE_DIRECT, i))) return rc;
    for(int i = 0; i < 5; i++) if((rc = run(c->stream, RT_STAGE_DENOISE_INDIRECT, i))) return rc;
  }
  if(fork) RT_HIP(c, hipStreamWaitEvent(c->stream, c->evJoin, 0));
  if((rc = run(c->stream, RT_STAGE_COMPOSE, 0))) return rc;
  E.count = k; E.last = lastMain;
  c->evUsed++;
  if(decide) {
    RT_HIP(c, syncAll(c));
    harvestTimings(c);
    const double traced = (c->accStage[RT_STAGE_DIRECT] + c->accStage[RT_STAGE_INDIRECT]) - tracedBefore;
    const double filt = (c->accStage[RT_STAGE_DENOISE_DIRECT] + c->accStage[RT_STAGE_DENOISE_INDIRECT] + c->accStage[RT_STAGE_COMPOSE]) - filterBefore;
    c->filterShare = traced > 0.0 ? float(filt / traced) : -1.f;
    c->denoiseSeen = st->denoise > 0 ? 1 : 0;
    if(++c->probeFrames >= prioProbeFrames()) {
      c->prio[1] = 1; c->prio[2] = (c->filterShare >= PRIO_FILTER_SHARE) ? 1 : 0;
      c->prioDecided = true;
      c->indStream = c->sideStream = nullptr;   // (created with these levels by the next frame: filter stream first, then the indirect stream)
    }
  }
  return RT_OK;
}

size_t rt_buffer_bytes(rt_ctx* c, int buffer) { return (c && buffer >= 0 && buffer < RT_BUF_COUNT) ? c->bufBytes[buffer] : 0; }

int rt_readback(rt_ctx* c, int buffer, void* dst, size_t bytes)
{
  if(!c || !dst || buffer < 0 || buffer >= RT_BUF_COUNT) return RT_ERR_INVALID_ARG;
  if(c->W == 0) return fail(c, RT_ERR_NO_TARGET, "rt_readback: rt_resize has not been called");
  if(bytes != c->bufBytes[buffer]) return fail(c, RT_ERR_INVALID_ARG, "rt_readback: size mismatch (see rt_buffer_bytes)");
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  RT_HIP(c, hipMemcpy(dst, c->bufs[buffer], bytes, hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_upload_history(rt_ctx* c, int buffer, const void* src, size_t bytes)
{
  if(!c || !src || buffer < 0 || buffer >= RT_BUF_COUNT) return RT_ERR_INVALID_ARG;
  if(c->W == 0) return fail(c, RT_ERR_NO_TARGET, "rt_upload_history: rt_resize has not been called");
  if(bytes != c->bufBytes[buffer]) return fail(c, RT_ERR_INVALID_ARG, "rt_upload_history: size mismatch (see rt_buffer_bytes)");
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  RT_HIP(c, hipMemcpy(c->bufs[buffer], src, bytes, hipMemcpyHostToDevice));
  if(buffer == RT_BUF_DENOISE_IND_A)   // both parities: the caller's next stage may name either frame
    for(void* p : c->indA) if(p && p != c->bufs[buffer]) RT_HIP(c, hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
  RT_HIP(c, hipDeviceSynchronize());
  return RT_OK;
}

int rt_device_ptr(rt_ctx* c, int buffer, void** ptr, size_t* bytes, size_t* rowPitch)
{
  if(!c || buffer < 0 || buffer >= RT_BUF_COUNT || !ptr) return RT_ERR_INVALID_ARG;
  if(c->W == 0) return fail(c, RT_ERR_NO_TARGET, "rt_device_ptr: rt_resize has not been called");
  *ptr = c->bufs[buffer];
  if(bytes) *bytes = c->bufAlloc[buffer];
  if(rowPitch) *rowPitch = size_t(halfRes(buffer) ? c->W / 2 : c->W) * elemBytes(buffer);
  return RT_OK;
}

int rt_set_counting(rt_ctx* c, int enable)
{
  if(!c) return RT_ERR_INVALID_ARG;
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  RT_HIP(c, hipMemset(c->dCounters, 0, 8 * sizeof(unsigned long long)));
  RT_HIP(c, hipDeviceSynchronize());
  harvestTimings(c);
  for(double& v : c->accStage) v = 0;
  c->accFrame = 0; c->accFrames = 0;
  c->counting = enable != 0;
  return RT_OK;
}

int rt_get_counters(rt_ctx* c, rt_counters* out)
{
  if(!c || !out) return RT_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  unsigned long long h[8];
  RT_HIP(c, hipMemcpy(h, c->dCounters, sizeof(h), hipMemcpyDeviceToHost));
  out->closestHitRays = h[0]; out->anyHitRays = h[1]; out->nodesVisited = h[2]; out->trisTested = h[3]; out->hitsShaded = h[4]; out->risCandidates = h[5]; out->laneRounds = h[6]; out->laneLiveRounds = h[7];
  harvestTimings(c);
  for(int i = 0; i < RT_STAGE_COUNT; i++) out->stageMs[i] = float(c->accStage[i]);
  out->frameMs = float(c->accFrame);
  out->framesTimed = c->accFrames;
  return RT_OK;
}

int rt_set_history_rows(rt_ctx* c, int row0, int row1)
{
  if(!c || row0 < 0 || row1 < row0) return RT_ERR_INVALID_ARG;
  c->histRow0 = row0; c->histRow1 = row1;
  return RT_OK;
}

int rt_history_miss(rt_ctx* c, int* missed)
{
  if(!c || !missed) return RT_ERR_INVALID_ARG;
  if(c->W == 0) return fail(c, RT_ERR_NO_TARGET, "rt_history_miss: rt_resize has not been called");
  RT_HIP(c, hipSetDevice(c->device));
  uint32_t v[2] = {0, 0};
  RT_HIP(c, syncAll(c));
  RT_HIP(c, hipMemcpyAsync(v, c->scratch.qcount + 250, sizeof(v), hipMemcpyDeviceToHost, c->stream));
  RT_HIP(c, hipMemsetAsync(c->scratch.qcount + 250, 0, sizeof(v), c->stream));
  RT_HIP(c, hipStreamSynchronize(c->stream));
  *missed = (v[0] | v[1]) ? 1 : 0;
  return RT_OK;
}

int rt_history_miss_stage(rt_ctx* c, int stage, int* missed)
{
  if(!c || !missed || stage < 0 || stage >= RT_STAGE_COUNT) return RT_ERR_INVALID_ARG;
  if(c->W == 0) return fail(c, RT_ERR_NO_TARGET, "rt_history_miss_stage: rt_resize has not been called");
  RT_HIP(c, hipSetDevice(c->device));
  uint32_t* flag = c->scratch.qcount + 250 + (stage == RT_STAGE_INDIRECT ? 1 : 0);
  uint32_t v = 0;
  RT_HIP(c, hipMemcpyAsync(&v, flag, sizeof(v), hipMemcpyDeviceToHost, c->stream));
  RT_HIP(c, hipMemsetAsync(flag, 0, sizeof(v), c->stream));
  RT_HIP(c, hipStreamSynchronize(c->stream));  // the ctx stream only: other streams of the host keep running
  *missed = v ? 1 : 0;
  return RT_OK;
}

int rt_select_frame(rt_ctx* c, int frames)
{
  if(!c) return RT_ERR_INVALID_ARG;
  selectFrame(c, frames);
  return RT_OK;
}

int rt_rotate_buffers(rt_ctx* c, int frames)
{
  if(!c) return RT_ERR_INVALID_ARG;
  if(c->W == 0 || !c->spareG || !c->spareMotion) return fail(c, RT_ERR_NO_TARGET, "rt_rotate_buffers: rt_resize has not been called");
  std::swap(c->bufs[RT_BUF_GBUFFER0 + (frames & 1)], c->spareG);
  std::swap(c->bufs[RT_BUF_MOTION], c->spareMotion);
  return RT_OK;
}

int rt_set_traversal(rt_ctx* c, int mode)
{
  if(!c || mode < RT_TRAVERSAL_AUTO || mode > RT_TRAVERSAL_LATENCY) return RT_ERR_INVALID_ARG;
  c->traversal = mode;
  return RT_OK;
}

int rt_pick(rt_ctx* c, const rt_mat4* modelViewInv, const rt_mat4* perspectiveInv, float pickX, float pickY, rt_pick_result* out)
{
  if(!c || !modelViewInv || !perspectiveInv || !out) return RT_ERR_INVALID_ARG;
  if(!c->haveScene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
  if(!c->haveAccel) return fail(c, RT_ERR_NO_ACCEL, "rt_build_accel has not been called");
  RT_HIP(c, hipSetDevice(c->device));
  if(!c->dPick) RT_HIP(c, hipMalloc(&c->dPick, sizeof(rt_pick_result)));
  rt_pick_result* d = static_cast<rt_pick_result*>(c->dPick);
  RT_HIP(c, launchPick(c->stream, c->ds, *modelViewInv, *perspectiveInv, pickX, pickY, d));
  RT_HIP(c, hipMemcpyAsync(out, d, sizeof(*out), hipMemcpyDeviceToHost, c->stream));
  RT_HIP(c, hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_trace_rays(rt_ctx* c, int n, const float* rays, float* out, int anyHit)
{
  if(!c || n < 0 || (n > 0 && (!rays || !out))) return RT_ERR_INVALID_ARG;
  if(!c->haveScene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
  if(!c->haveAccel) return fail(c, RT_ERR_NO_ACCEL, "rt_build_accel has not been called");
  if(n == 0) return RT_OK;
  RT_HIP(c, hipSetDevice(c->device));
  float4* dr = nullptr; float4* dout = nullptr;
  RT_HIP(c, hipMalloc(reinterpret_cast<void**>(&dr), size_t(n) * 32));
  if(hipMalloc(reinterpret_cast<void**>(&dout), size_t(n) * 16) != hipSuccess) { (void)hipFree(dr); return fail(c, RT_ERR_OOM, "rt_trace_rays: hipMalloc failed"); }
  hipError_t e = hipMemcpyAsync(dr, rays, size_t(n) * 32, hipMemcpyHostToDevice, c->stream);
  if(e == hipSuccess) e = launchTraceRays(c->stream, c->ds, n, dr, dout, anyHit);
  if(e == hipSuccess) e = hipMemcpyAsync(out, dout, size_t(n) * 16, hipMemcpyDeviceToHost, c->stream);
  if(e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dr); (void)hipFree(dout);
  RT_HIP(c, e);
  return RT_OK;
}

int rt_set_sun_and_sky(rt_ctx* c, const rt_sun_and_sky* ss)
{
  if(!c || !ss) return RT_ERR_INVALID_ARG;
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  c->sunAndSky = *ss;
  if(ss->in_use == 1) {
    if(!c->dSky) RT_HIP(c, hipMalloc(&c->dSky, 512));
    RT_HIP(c, launchSkyPrepare(c->stream, *ss, static_cast<SkyPre*>(c->dSky)));
    RT_HIP(c, hipStreamSynchronize(c->stream));
  }
  c->ds.sky = (ss->in_use == 1) ? static_cast<const SkyPre*>(c->dSky) : nullptr;
  return RT_OK;
}

int rt_tonemap(rt_ctx* c, const rt_tonemapper* tm, int debugging_mode, int frames)
{
  if(!c || !tm) return RT_ERR_INVALID_ARG;
  if(c->W == 0) return fail(c, RT_ERR_NO_TARGET, "rt_tonemap: rt_resize has not been called");
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, joinInFlight(c));  // the result images of `frames` are complete once compose (side stream) is done
  const int cur = frames & 1;
  RT_HIP(c, launchTonemap(c->stream, static_cast<const float4*>(c->bufs[RT_BUF_DIRECT_RESULT0 + cur]), static_cast<const float4*>(c->bufs[RT_BUF_INDIRECT_RESULT0 + cur]),
                          c->scratch.postRowSums, c->scratch.postMean, *tm, debugging_mode, c->W, c->H, static_cast<uint32_t*>(c->bufs[RT_BUF_LDR]),
                          c->scratch.postMipD, c->scratch.postMipI));
  return RT_OK;
}

int rt_set_overlap(rt_ctx* c, int mode)
{
  if(!c || mode < 0 || mode > 3) return RT_ERR_INVALID_ARG;
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  const bool hadShort = c->overlap >= 2;
  c->overlap = mode;
  if((mode >= 2) != hadShort) return ensureStackOverflow(c);   // the short LDS stacks of frames in flight spill into an HBM area that serial schedules do not hold
  return RT_OK;
}

/* Priorities of the indirect and the filter stream of the frames-in-flight schedule (levels -1 low, 0 normal, +1 high; the main stream keeps the level it was
 * created with).  Which setting is fastest depends on the workload (profiles/r05_prio_by_config_ab.txt); unset, the context decides at its first frame
 * (rt_render_frame: the rule and its data).  Best called BEFORE the first frame: a stream created after others exist may share a hardware queue with them.
 * Results are identical under every setting. */
int rt_set_stream_priorities(rt_ctx* c, int indirectLevel, int filterLevel)
{
  if(!c || indirectLevel < -1 || indirectLevel > 1 || filterLevel < -1 || filterLevel > 1) return RT_ERR_INVALID_ARG;
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  c->prioDecided = c->prioExplicit = true;   // an explicit choice: the probe-frame rule stays out of it, also after rt_resize / a new scene
  if(c->prio[1] == indirectLevel && c->prio[2] == filterLevel) return RT_OK;
  c->prio[1] = indirectLevel; c->prio[2] = filterLevel;
  c->indStream = c->sideStream = nullptr;   // (selected — and created, once per level — on first use: ensureOverlapStreams; streams of other levels stay alive, idle)
  return RT_OK;
}

/* The levels in use and the measurement behind them: filterShare = filters / (direct + indirect) of the last probe frame, every stage alone (-1: no frame yet, or the
 * levels were set explicitly / by RESTIR_PRIO); decided = 0 while the probe frames have not all been rendered. */
int rt_get_stream_priorities(rt_ctx* c, int* indirectLevel, int* filterLevel, float* filterShare, int* decided)
{
  if(!c) return RT_ERR_INVALID_ARG;
  if(indirectLevel) *indirectLevel = c->prio[1];
  if(filterLevel) *filterLevel = c->prio[2];
  if(filterShare) *filterShare = c->filterShare;
  if(decided) *decided = c->prioDecided ? 1 : 0;
  return RT_OK;
}

/* The context's three streams of the frames-in-flight schedule, for a host that issues the stages itself (rt_run_stage + rt_set_stream: restir_amd/tiled.py,
 * csrc/mgpu.cpp).  Creates the filter and the indirect stream if they do not exist yet — with the current levels, in the order the schedule was tuned for (filter
 * stream first, then the indirect stream, nothing in between) — so that such a host runs on the SAME stream layout as rt_render_frame instead of on streams from
 * another pool created whenever (round-5 finding: a stream's place in the process's creation order is worth 15-75 % of a frame).  Call it before anything else in the
 * process creates streams (torch's pool, RCCL's communicator).  The handles stay owned by the context. */
int rt_get_streams(rt_ctx* c, void** mainStream, void** indirectStream, void** filterStream)
{
  if(!c) return RT_ERR_INVALID_ARG;
  RT_HIP(c, hipSetDevice(c->device));
  if(indirectStream || filterStream) {   // (asking for the main stream alone creates nothing: csrc/mgpu.cpp creates a rank's other two on its first frame in flight)
    RT_HIP(c, ensureOverlapStreams(c));
    c->prioDecided = true;   // the levels these streams were created with stand: a later rt_render_frame does not probe and switch
  }
  if(mainStream) *mainStream = c->ownStream;
  if(indirectStream) *indirectStream = c->indStream;
  if(filterStream) *filterStream = c->sideStream;
  return RT_OK;
}

/* Where this context's streams sit in the process's creation order: created = HIP streams this LIBRARY has created in the process so far (all contexts; streams made by
 * the host — torch's pool, RCCL — are not visible to it), index[0..2] = creation index of the main / indirect / filter stream in use (-1: not created yet).  bench.py
 * prints it with every N > 1 line (`stream_layout`). */
int rt_get_stream_layout(rt_ctx* c, int* created, int index[3])
{
  if(!c) return RT_ERR_INVALID_ARG;
  if(created) *created = g_streamsCreated.load();
  if(index) { index[0] = c->mainIdx; index[1] = c->indStream ? c->indIdx[c->prio[1] + 1] : -1; index[2] = c->sideStream ? c->sideIdx[c->prio[2] + 1] : -1; }
  return RT_OK;
}

/* extra introspection used by bench.py / DESIGN.md numbers (not part of the reference-facing surface) */
int rt_accel_stats(rt_ctx* c, uint64_t* numNodes, uint64_t* numTris, int* maxDepth)
{
  if(!c || !c->haveAccel) return RT_ERR_NO_ACCEL;
  if(numNodes) *numNodes = c->numNodes;
  if(numTris) *numTris = c->numTris;
  if(maxDepth) *maxDepth = c->maxDepth;
  return RT_OK;
}

/* tree quality: leaf records (references: > triangles when spatial splits duplicated some), spatial splits taken, and the SAH expectation of node / triangle
 * steps of a ray that hits the root box */
int rt_accel_quality(rt_ctx* c, uint64_t* references, uint64_t* spatialSplits, double* sahNodeSteps, double* sahTriSteps)
{
  if(!c || !c->haveAccel) return RT_ERR_NO_ACCEL;
  if(references) *references = c->numRefs;
  if(spatialSplits) *spatialSplits = c->spatialSplits;
  if(sahNodeSteps) *sahNodeSteps = c->sahNodeSteps;
  if(sahTriSteps) *sahTriSteps = c->sahTriSteps;
  return RT_OK;
}

#if RT_WAVEPROF
/* measurement builds only (scripts/wave_profile.py): read and clear the per-wave profile records kept in the hitRec scratch */
int rt_debug_wave_profile(rt_ctx* c, void* dst, size_t bytes)
{
  if(!c || !dst || c->W == 0) return RT_ERR_INVALID_ARG;
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  bytes = std::min(bytes, WAVEPROF_RECORDS * 64);
  RT_HIP(c, hipMemcpy(dst, c->scratch.waveProf, bytes, hipMemcpyDeviceToHost));
  RT_HIP(c, hipMemset(c->scratch.waveProf, 0, bytes));
  RT_HIP(c, hipDeviceSynchronize());
  return RT_OK;
}
#endif

int rt_measure_valu_peak(rt_ctx* c, int variant, int wavesPerSimd, double* waveInstPerSec)
{
  if(!c || !waveInstPerSec || variant < 0 || variant > 1 || wavesPerSimd < 1 || wavesPerSimd > 8) return RT_ERR_INVALID_ARG;
  RT_HIP(c, hipSetDevice(c->device));
  RT_HIP(c, syncAll(c));
  double seconds = 0.0;
  RT_HIP(c, measureValuIssue(c->stream, variant, wavesPerSimd, waveInstPerSec, &seconds));
  return RT_OK;
}

}  // extern "C"
