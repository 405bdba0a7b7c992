// mgpu.cpp — rt_mgpu_*: the row-tiled multi-GPU frame as ONE native context that drives N devices from one
// process: one host thread, one rt_ctx and one (barrier schedule) or three (frames in flight) HIP streams per device, halos moved with
// hipMemcpyPeerAsync over xGMI.
// It is the C-ABI form of what restir_amd/tiled.py does over torch.distributed: a C++ host (the reference's language,
// src/main.cpp:200-264) gets the multi-GPU frame behind the same five calls as the single-GPU one, and no Python runs per frame.
//
// Partition: contiguous row bands whose boundaries are multiples of 16 rows (8x8 tiles and the half-resolution grid stay aligned).
// Band heights are COST WEIGHTED: every rank times its ray-traced stages with HIP events, the times are spread over the rank's
// 16-row stripes into a smoothed per-stripe cost, and the next frame's boundaries equalise the summed cost (sky rows are cheap,
// street rows expensive: equal heights leave a 1.6x spread at 8 ranks).  A boundary moves at most two stripes per frame, which is
// what the history halo covers.  Scene, BVH8 and full-size screen buffers are replicated on every device; RNG seeds use global
// pixel indices, so every output is bit-identical to the single-GPU frame (tests/test_gpu_mgpu.py).
//
// BARRIER SCHEDULE per rank (rt_mgpu_set_pipeline(0); host barriers between the numbered steps; every copy is a PULL by the rank that needs the rows):
//   1. history: last frame's G-buffer / direct reservoirs / light ids / indirect reservoirs for the band +- 32 rows, from the ranks
//      that owned those rows LAST frame (this also moves state when a boundary moved)
//   2. direct stage on the band, indirect stage on the band's half-res rows; a temporal lookup outside band + halo raises a flag;
//      if any rank's flag is up, every rank pulls the full history and runs the two stages again (exact for any camera motion)
//   3. one exchange for all nine A-Trous passes: 144 G-buffer rows, 40 rows of noisy direct colour, 72 half-res rows of noisy
//      indirect colour from the owners; then the filters on regions that start wider than the band and shrink per level
//      (direct +32/+24/+16/+0, indirect +64/+56/+48/+32/+0 rows: the overlap is recomputed instead of exchanged 9 times); compose
//   4. rank 0 pulls the two result images' bands (the display rank of SURVEY §8e(3))
//
// FRAMES IN FLIGHT (rt_mgpu_set_pipeline, the default): the schedule of rt_render_frame's overlap mode 2 (src/renderer.cpp:154-206 is the reference
// order it preserves) per rank — three streams: main = direct(f), ind = indirect(f), side = filters + compose — with every cross-rank dependency a HIP
// event another rank's stream waits for (hipStreamWaitEvent), no host barrier between the steps and none between the ranks:
//   main(f): waits D(f-1) of the ranks it pulls history rows from, and I(f-2) / Done(f-2) / G(f-2) of everybody who still reads what it overwrites
//   ind(f) : waits its own D(f) and the neighbours' I(f-1) (indirect-reservoir history rows)
//   side   : second half of frame f-1 (indirect filter halo pull, 5 levels, compose -> Done(f-1); rank 0's copy stream gathers the result bands -> G(f-1)),
//            then the direct filters of frame f after the neighbours' D(f); a level that overwrites rows a neighbour pulls waits for that pull (X / XI events)
// rt_mgpu_render_frame only queues the frame for the rank threads (at most two frames ahead) and returns.  The host side of a rank blocks in two places per
// frame, on ITS OWN streams only: reading the history-miss flag of its direct stage and, one call later, of its indirect stage; a miss is handled by the rank
// alone (it pulls the full history from the owners — nobody overwrites it before this rank's D / I event — and re-runs its stage).  Host-side counters
// (`issued`) only make sure an event has been RECORDED before another thread makes a stream wait for it.  The spatial-reuse modes, whose direct stage
// exchanges rows in the middle, and the measurement mode rt_mgpu_set_serialize run the barrier schedule above.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/rt_abi.h"

// rt_create with explicit stream levels (csrc/rt_api.cpp; internal, not part of the C ABI)
int rtCreateWithLevels(rt_ctx** out, int device, const int* levels);

#ifndef RT_TEST_HOOKS
#define RT_TEST_HOOKS 0
#endif

namespace {

// default levels of a rank's main / indirect / filter stream (RESTIR_MGPU_PRIO overrides): rt_mgpu_create
constexpr int MGPU_PRIO_DEFAULT[3] = {0, 1, 0};

// full-res rows of last-frame history around the band: adaptive (rt_mgpu::histHalo) — HIST_HALO_MIN while no temporal lookup leaves band + halo, doubled
// (up to HIST_HALO_MAX) for the frames after one did, halved again after HIST_HALO_CALM frames without; a lookup outside is always caught (exact fallback)
constexpr int HIST_HALO_MIN = 16, HIST_HALO_MAX = 64, HIST_HALO_CALM = 16;
constexpr int DIRECT_GROW[4] = {32, 24, 16, 0};     // rows added to the band for A-Trous level l's output (multiples of 8)
constexpr int INDIRECT_GROW[5] = {64, 56, 48, 32, 0};
constexpr int HALO_DIRECT_COLOR = 40, HALO_INDIRECT_COLOR = 72;
// G-buffer rows the nine filter passes read beyond the band: the direct chain reaches DIRECT_GROW[0] + 2 rows (every row), the indirect chain
// INDIRECT_GROW[0] + 2 half-res rows = 132 full-res rows, of which it reads the even ones only (loadThisGeometry(2q), denoise_common.glsl:42-55)
constexpr int HALO_GBUFFER_FULL = 40, HALO_GBUFFER = 144;
constexpr int MAX_RANKS = RT_MGPU_MAX_RANKS;
constexpr int RING = 8;                             // per-frame events are reused every RING frames (a rank is never more than 3 frames from its peers)

struct Barrier {   // reusable, for a fixed number of threads
  std::mutex m; std::condition_variable cv; int n = 0, count = 0; uint64_t gen = 0;
  void init(int k) { n = k; count = 0; }
  void wait()
  {
    std::unique_lock<std::mutex> l(m);
    const uint64_t g = gen;
    if(++count == n) { count = 0; gen++; cv.notify_all(); }
    else cv.wait(l, [&] { return gen != g; });
  }
};

// what a rank thread needs to render one frame: a copy, because rt_mgpu_render_frame returns before the frame is issued (frames in flight)
struct FrameCmd {
  rt_state st{}; int frames = 0; rt_scene_camera cam{};
  int bands[MAX_RANKS + 1] = {}, prev[MAX_RANKS + 1] = {};
  bool haveHistory = false, gather = true;
  int histHalo = HIST_HALO_MIN;
  bool samePartition = false;   // bands == prev: rows the previous frame's filter halo brought need no second pull
  int64_t seq = 0;      // frames since the pipeline was (re)started
  int solo = -1;        // measurement: only this rank renders (the others publish their flags and skip)
};
struct Cmd {
  enum Kind { UPLOAD, RESIZE, FRAME, FRAMEP, DRAIN, QUIT } kind = QUIT;
  FrameCmd f;
};
enum HaloKind { HK_HISTORY = 0, HK_FILTER = 1, HK_SPATIAL = 2, HK_MOVED = 3, HK_GATHER = 4, HK_FALLBACK = 5, HK_COUNT = 6 };

struct Rank {
  int id = 0, dev = 0;
  rt_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;                                       // barrier schedule: everything; pipelined schedule: "main"
  hipStream_t sInd = nullptr, sSide = nullptr, sCopy = nullptr;
  hipEvent_t ev[4] = {};   // barrier schedule timing: start, traced, filters-begin, end
  // pipelined schedule: one event per frame slot; issued[k] = the latest frame sequence number whose event k has been RECORDED (host-side only)
  enum { E_D = 0, E_I, E_DONE, E_G, E_X, E_XI, E_KINDS };
  hipEvent_t evp[E_KINDS][RING] = {};
  std::atomic<int64_t> issued[E_KINDS];
  hipEvent_t tm[RING][8] = {};                                       // timing: direct begin/end, indirect begin/end, direct filters begin/end, second half begin/end
  void* gptr[RING] = {};                                              // this rank's physical G-buffer of frame seq (the boundary ids rotate)
  void* indA[2] = {};                                                 // its two noisy-indirect-colour buffers, by frame parity (rt_select_frame)
  bool havePrev = false; FrameCmd prevCmd;                           // frame whose second half is still to be issued
  int64_t aIssued = -1;                                               // frame sequence number whose direct stage has been issued (look-ahead)
  int rotatedFor = -1;                                                // `frames` value the rotating buffer ids currently name
  int mode = -1;                                                      // rt_set_overlap value of the ctx (0: barrier schedule, 2: frames in flight)
  std::thread th;
  std::mutex qm; std::condition_variable qcv; std::deque<Cmd> q;
  int rc = RT_OK;
  float tracedMs = 0, filterMs = 0;
  uint64_t pulled[HK_COUNT] = {};        // bytes pulled for the frame being issued (barrier schedule) / by frame parity (pulledBy, frames in flight)
  uint64_t pulledBy[2][HK_COUNT] = {};
  uint64_t lastPulled[HK_COUNT] = {};    // ... of the latest COMPLETE frame: what rt_mgpu_get_stats reports
  int accSlot = -1;                      // >= 0: pullRowsOn accounts to pulledBy[accSlot]
  // link timing (frames in flight): the four pull groups of a frame bracketed by events on the stream that carries them — history rows (main), indirect-reservoir
  // history (ind), direct filter halo (side), indirect filter halo (side) — and their bytes; rt_mgpu_get_link_stats reports the latest complete frame
  enum { LG_HISTORY = 0, LG_HISTORY_IND, LG_FILTER_DIRECT, LG_FILTER_INDIRECT, LG_COUNT };
  hipEvent_t tl[RING][2 * LG_COUNT] = {};
  uint64_t grpBytes[RING][LG_COUNT] = {};
  int grp = -1, grpSlot = 0;             // >= 0: pullRowsOn also accounts to grpBytes[grpSlot][grp]
  float linkMs[LG_COUNT] = {}; uint64_t linkBytes[LG_COUNT] = {};
  Rank() { for(auto& a : issued) a.store(-1); }
};

}  // namespace

extern "C" int rt_mgpu_plan_bands(int height, int numRanks, const float* stripeCost, const int* prevBands, int maxMoveStripes, int* outBands);

struct rt_mgpu {
  int n = 0;
  std::vector<Rank> ranks;
  int W = 0, H = 0;
  std::vector<int> bands, prevBands;   // n + 1 row boundaries (multiples of 16, last = H)
  std::vector<float> stripeCost;       // smoothed cost per 16-row stripe
  std::vector<float> rankMs;           // smoothed time per rank (diffusion phase of the balancer)
  std::vector<int> boundaryCool;       // per rank: frames until its measured time belongs to its new band
  int balanceFrames = 0;               // frames balanced since the partition was last reset
  bool haveHistory = false, balance = true, serialize = false, gatherResults = true, pipeline = true;
  int solo = -1;
  int levels[3] = {0, 0, 0};           // priority levels of every rank's main / indirect / filter stream (rt_mgpu_create)
  bool corruptHalo = false;            // test hook (RESTIR_TEST_CORRUPT_HALO=1): the first bytes of every pulled noisy-direct-colour halo are overwritten, so the
                                       // bench's tiled == untiled gate must report a mismatch (tests/test_gpu_bench_cli.py).  Never set outside tests.
  std::vector<uint8_t> peerOk;         // [puller rank * n + owner rank]: direct peer access from the puller's device to the owner's is enabled (rt_mgpu_create)
  int64_t seq = 0; int lastFrames = -2; bool pipeActive = false;
  int histHalo = HIST_HALO_MIN; uint32_t fallbacksSeen = 0; int calmFrames = 0; bool lastDenoise = false;
  rt_scene_camera cam{}; const rt_scene_desc* desc = nullptr;
  // completion of posted commands
  std::mutex pm; std::condition_variable pcv; int pending = 0;
  Barrier step;
  int missFlags[MAX_RANKS] = {};
  std::mutex turn;                     // serialize mode: one rank's kernels at a time
  std::mutex buildTurn;                // one multi-threaded host BVH8 build at a time
  std::atomic<bool> abort{false};
  std::atomic<uint32_t> fallbacks{0};
  std::string err;
  rt_mgpu_stats stats{};
  std::mutex errLock;
  void fail(int rank, int rc, const char* what, const char* detail)
  {
    std::lock_guard<std::mutex> l(errLock);
    if(err.empty()) err = std::string("rt_mgpu rank ") + std::to_string(rank) + ": " + what + " failed (" + std::to_string(rc) + "): " + (detail ? detail : "");
    abort = true;
  }
};

namespace {

bool halfRows(int buf)
{
  return buf == RT_BUF_INDIRECT_RESV0 || buf == RT_BUF_INDIRECT_RESV1 || buf == RT_BUF_INDIRECT_RESV_TEMP || buf == RT_BUF_DENOISE_IND_A || buf == RT_BUF_DENOISE_IND_B;
}

#define MG_CHECK(call, what)                                                             \
  do {                                                                                   \
    const int rc_ = (call);                                                              \
    if(rc_ != RT_OK) { M.fail(R.id, rc_, what, rt_last_error(R.ctx)); R.rc = rc_; }      \
  } while(0)
#define MG_HIP(call, what)                                                               \
  do {                                                                                   \
    const hipError_t e_ = (call);                                                        \
    if(e_ != hipSuccess) { M.fail(R.id, int(e_), what, hipGetErrorString(e_)); R.rc = RT_ERR_HIP; } \
  } while(0)

// copy rows [a, b) of `buf` into rank R's copy from the ranks that own them under `part` (full-res boundaries), on `strm`.  gslot >= 0: `buf` is a
// G-buffer and both sides address the physical buffer of that frame slot (the ids rotate with frames in flight).
// evenOnly: only the even rows of [a, b) (the half-resolution filters read G(2q)); widthBytes > 0: only the first widthBytes of every row (the half-res filter
// temporaries live in the left half of full-pitch images).
void pullRowsOn(rt_mgpu& M, Rank& R, hipStream_t strm, int buf, int a, int b, const int* part, int kind, int gslot = -1, bool evenOnly = false, size_t widthBytes = 0, int indParity = -1)
{
  const bool half = halfRows(buf);
  const int limit = half ? M.H / 2 : M.H;
  a = std::max(0, a); b = std::min(limit, b);
  if(b <= a) return;
  void* dst = nullptr; size_t bytes = 0, pitch = 0;
  MG_CHECK(rt_device_ptr(R.ctx, buf, &dst, &bytes, &pitch), "rt_device_ptr");
  if(gslot >= 0) dst = R.gptr[gslot];
  if(indParity >= 0) dst = R.indA[indParity];
  for(int q = 0; q < M.n; q++) {
    if(q == R.id) continue;
    int lo = half ? part[q] / 2 : part[q], hi = half ? part[q + 1] / 2 : part[q + 1];
    hi = std::min(hi, limit);
    lo = std::max(lo, a); hi = std::min(hi, b);
    if(hi <= lo) continue;
    void* src = nullptr; size_t sb = 0, sp = 0;
    Rank& Q = M.ranks[q];
    if(rt_device_ptr(Q.ctx, buf, &src, &sb, &sp) != RT_OK || sp != pitch) { M.fail(R.id, RT_ERR_INVALID_ARG, "peer rt_device_ptr", ""); R.rc = RT_ERR_INVALID_ARG; return; }
    if(gslot >= 0) src = Q.gptr[gslot];
    if(indParity >= 0) src = Q.indA[indParity];
    if(!dst || !src) continue;   // (frame slot never rendered on that rank: nothing to pull)
    if(evenOnly || (widthBytes > 0 && widthBytes < pitch)) {
      if(evenOnly) lo = (lo + 1) & ~1;
      const int rows = evenOnly ? (hi - lo + 1) / 2 : hi - lo;
      if(rows <= 0) continue;
      const size_t off = size_t(lo) * pitch, stride = evenOnly ? 2 * pitch : pitch, w = widthBytes > 0 ? std::min(widthBytes, pitch) : pitch;
      if(Q.dev == R.dev || M.peerOk[size_t(R.id) * size_t(M.n) + size_t(q)]) {
        // (unified addressing + peer access: a 2-D device-to-device copy may cross devices)
        MG_HIP(hipMemcpy2DAsync(static_cast<char*>(dst) + off, stride, static_cast<char*>(src) + off, stride, w, size_t(rows), Q.dev == R.dev ? hipMemcpyDeviceToDevice : hipMemcpyDefault, strm), "hipMemcpy2DAsync");
      } else {
        // no peer access between the two devices: hipMemcpyPeerAsync still works (staged by the runtime), a cross-device 2-D default copy may not — row by row
        for(int y = 0; y < rows; y++)
          MG_HIP(hipMemcpyPeerAsync(static_cast<char*>(dst) + off + size_t(y) * stride, R.dev, static_cast<char*>(src) + off + size_t(y) * stride, Q.dev, w, strm), "hipMemcpyPeerAsync (row)");
      }
      (R.accSlot >= 0 ? R.pulledBy[R.accSlot] : R.pulled)[kind] += w * size_t(rows);
      if(R.grp >= 0) R.grpBytes[R.grpSlot][R.grp] += w * size_t(rows);
      continue;
    }
    const size_t off = size_t(lo) * pitch, len = size_t(hi - lo) * pitch;
    if(Q.dev == R.dev) MG_HIP(hipMemcpyAsync(static_cast<char*>(dst) + off, static_cast<char*>(src) + off, len, hipMemcpyDeviceToDevice, strm), "hipMemcpyAsync");
    else MG_HIP(hipMemcpyPeerAsync(static_cast<char*>(dst) + off, R.dev, static_cast<char*>(src) + off, Q.dev, len, strm), "hipMemcpyPeerAsync");
    (R.accSlot >= 0 ? R.pulledBy[R.accSlot] : R.pulled)[kind] += len;
    if(R.grp >= 0) R.grpBytes[R.grpSlot][R.grp] += len;
#if RT_TEST_HOOKS
    if(M.corruptHalo && kind == HK_FILTER && (buf == RT_BUF_DIRECT_RESULT0 || buf == RT_BUF_DIRECT_RESULT1))
      MG_HIP(hipMemsetAsync(static_cast<char*>(dst) + off, 0x3f, std::min<size_t>(len, 1024), strm), "hipMemsetAsync (test hook)");
#endif
  }
}
void pullRows(rt_mgpu& M, Rank& R, int buf, int a, int b, const std::vector<int>& part, int kind) { pullRowsOn(M, R, R.stream, buf, a, b, part.data(), kind); }
// the filter halos of one frame: G-buffer (every row next to the band, even rows further out) and noisy direct colour
void pullFilterHaloDirect(rt_mgpu& M, Rank& R, hipStream_t strm, int cur, int y0, int y1, const int* part, int gslot)
{
  pullRowsOn(M, R, strm, RT_BUF_GBUFFER0 + cur, y0 - HALO_GBUFFER_FULL, y1 + HALO_GBUFFER_FULL, part, HK_FILTER, gslot);
  pullRowsOn(M, R, strm, RT_BUF_GBUFFER0 + cur, y0 - HALO_GBUFFER, y0 - HALO_GBUFFER_FULL, part, HK_FILTER, gslot, true);
  pullRowsOn(M, R, strm, RT_BUF_GBUFFER0 + cur, y1 + HALO_GBUFFER_FULL, y1 + HALO_GBUFFER, part, HK_FILTER, gslot, true);
  pullRowsOn(M, R, strm, RT_BUF_DIRECT_RESULT0 + cur, y0 - HALO_DIRECT_COLOR, y1 + HALO_DIRECT_COLOR, part, HK_FILTER);
}
void pullFilterHaloIndirect(rt_mgpu& M, Rank& R, hipStream_t strm, int frames, int h0, int h1, const int* part)
{
  pullRowsOn(M, R, strm, RT_BUF_DENOISE_IND_A, h0 - HALO_INDIRECT_COLOR, h1 + HALO_INDIRECT_COLOR, part, HK_FILTER, -1, false, size_t(M.W / 2) * 16, frames & 1);
}

void runStage(rt_mgpu& M, Rank& R, const rt_state& st, int frames, int stage, int level, int r0, int r1, int limit)
{
  r0 = std::max(0, r0); r1 = std::min(limit, r1);
  if(r1 > r0) MG_CHECK(rt_run_stage(R.ctx, &st, frames, stage, level, r0, r1), "rt_run_stage");
}

// ==================================================================================================================================================
// barrier schedule
// ==================================================================================================================================================
// direct stage on the band, indirect stage on its half-res rows.  With spatial reuse the direct stage runs in two halves around an
// exchange: every pixel caches its reservoir (level 1), the ranks pull the rows next to their band from RT_BUF_DIRECT_RESV_TEMP — the
// neighbour picks of direct_stage.comp:86-107 reach one pixel up / down — and then merge and shade (level 2).  Contains one barrier in
// the spatial modes (uniform over ranks: the mode comes from the shared RtxState).
void tracedStages(rt_mgpu& M, Rank& R, const FrameCmd& c, int y0, int y1, int h0, int h1)
{
  const std::vector<int> bands(c.bands, c.bands + M.n + 1);
  const bool spatial = M.n > 1 && (c.st.ReSTIRState == RT_RESTIR_SPATIAL || c.st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL);
  if(!spatial) runStage(M, R, c.st, c.frames, RT_STAGE_DIRECT, 0, y0, y1, M.H);
  else {
    runStage(M, R, c.st, c.frames, RT_STAGE_DIRECT, 1, y0, y1, M.H);
    MG_HIP(hipStreamSynchronize(R.stream), "sync");
    M.step.wait();
    pullRows(M, R, RT_BUF_DIRECT_RESV_TEMP, y0 - 2, y0, bands, HK_SPATIAL);
    pullRows(M, R, RT_BUF_DIRECT_RESV_TEMP, y1, y1 + 2, bands, HK_SPATIAL);
    runStage(M, R, c.st, c.frames, RT_STAGE_DIRECT, 2, y0, y1, M.H);
  }
  runStage(M, R, c.st, c.frames, RT_STAGE_INDIRECT, 0, h0, h1, M.H / 2);
}

void frameOnRank(rt_mgpu& M, Rank& R, const FrameCmd& c)
{
  const int f = c.frames, cur = f & 1, last = cur ^ 1, H = M.H, Hh = H / 2;
  const std::vector<int> bands(c.bands, c.bands + M.n + 1), prevBands(c.prev, c.prev + M.n + 1);
  const int y0 = bands[R.id], y1 = bands[R.id + 1];
  const int h0 = std::min(y0 / 2, Hh), h1 = std::min(y1 / 2, Hh);
  const bool multi = M.n > 1;
  for(auto& p : R.pulled) p = 0;
  R.accSlot = -1;
  if(R.mode != 0) { MG_CHECK(rt_set_overlap(R.ctx, 0), "rt_set_overlap"); R.mode = 0; }   // launches run alone: the whole traversal stack in LDS
  MG_CHECK(rt_set_stream(R.ctx, R.stream), "rt_set_stream");
  MG_CHECK(rt_set_camera(R.ctx, &c.cam), "rt_set_camera");
  // ---- 1. history rows for the band + halo, from last frame's owners ----
  if(multi && c.haveHistory) {
    const int HIST_HALO = c.histHalo;
    // (the G-buffer rows next to the band arrived with last frame's filter halo when the partition did not move)
    if(!(c.samePartition && HIST_HALO <= HALO_GBUFFER_FULL)) pullRows(M, R, RT_BUF_GBUFFER0 + last, y0 - HIST_HALO, y1 + HIST_HALO, prevBands, HK_HISTORY);
    for(int buf : {RT_BUF_DIRECT_RESV0 + last, RT_BUF_LIGHT_ID0 + last}) pullRows(M, R, buf, y0 - HIST_HALO, y1 + HIST_HALO, prevBands, HK_HISTORY);
    pullRows(M, R, RT_BUF_INDIRECT_RESV0 + last, h0 - HIST_HALO / 2, h1 + HIST_HALO / 2, prevBands, HK_HISTORY);
    // rows that changed owner also bring the OTHER parity of the reservoir buffers along: pixels that return early (miss, emitter,
    // debug view) leave their slot untouched (reference quirk), so a slot of this frame's buffer can keep the value of
    // two frames ago.  Nothing is copied while the partition stands still (pullRows only copies rows other ranks owned).
    for(int buf : {RT_BUF_DIRECT_RESV0 + cur, RT_BUF_LIGHT_ID0 + cur, int(RT_BUF_DIRECT_RESV_TEMP)}) pullRows(M, R, buf, y0, y1, prevBands, HK_MOVED);
    pullRows(M, R, RT_BUF_INDIRECT_RESV0 + cur, h0, h1, prevBands, HK_MOVED);
  }
  MG_CHECK(rt_set_history_rows(R.ctx, multi ? std::max(0, y0 - c.histHalo) : 0, multi ? std::min(H, y1 + c.histHalo) : H), "rt_set_history_rows");
  // ---- 2. ray-traced stages ----
  {
    const bool spatialSplit = multi && (c.st.ReSTIRState == RT_RESTIR_SPATIAL || c.st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL);
    std::unique_lock<std::mutex> turn(M.turn, std::defer_lock);
    if(M.serialize && !spatialSplit) { turn.lock(); MG_HIP(hipStreamSynchronize(R.stream), "sync"); }   // (the split stage has a barrier inside: ranks cannot take turns)
    MG_HIP(hipEventRecord(R.ev[0], R.stream), "hipEventRecord");
    tracedStages(M, R, c, y0, y1, h0, h1);
    MG_HIP(hipEventRecord(R.ev[1], R.stream), "hipEventRecord");
    int miss = 0;
    if(multi) MG_CHECK(rt_history_miss(R.ctx, &miss), "rt_history_miss");   // waits for the stream
    else MG_HIP(hipStreamSynchronize(R.stream), "sync");
    M.missFlags[R.id] = miss;
  }
  M.step.wait();
  bool any = false;
  for(int q = 0; q < M.n; q++) any = any || M.missFlags[q] != 0;
  if(any) {  // exact fallback: the whole history, then the two stages again
    for(int buf : {RT_BUF_GBUFFER0 + last, RT_BUF_DIRECT_RESV0 + last, RT_BUF_LIGHT_ID0 + last}) pullRows(M, R, buf, 0, H, prevBands, HK_FALLBACK);
    pullRows(M, R, RT_BUF_INDIRECT_RESV0 + last, 0, Hh, prevBands, HK_FALLBACK);
    MG_CHECK(rt_set_history_rows(R.ctx, 0, H), "rt_set_history_rows");
    const bool spatialSplit = c.st.ReSTIRState == RT_RESTIR_SPATIAL || c.st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL;
    std::unique_lock<std::mutex> turn(M.turn, std::defer_lock);
    if(M.serialize && !spatialSplit) turn.lock();
    tracedStages(M, R, c, y0, y1, h0, h1);
    int miss = 0;
    MG_CHECK(rt_history_miss(R.ctx, &miss), "rt_history_miss");   // clears the flag, waits for the stream
    if(R.id == 0) M.fallbacks++;
  }
  M.step.wait();   // every rank's G-buffer / reservoirs / noisy colours of this frame are complete
  // ---- 3. one exchange for the nine filter passes, filters, compose ----
  if(multi && c.st.denoise > 0) {
    pullFilterHaloDirect(M, R, R.stream, cur, y0, y1, bands.data(), -1);
    pullFilterHaloIndirect(M, R, R.stream, f, h0, h1, bands.data());
    MG_HIP(hipStreamSynchronize(R.stream), "sync");
  }
  M.step.wait();   // nobody overwrites a source row (level 3 / compose write the result image) before every pull has landed
  {
    std::unique_lock<std::mutex> turn(M.turn, std::defer_lock);
    if(M.serialize) turn.lock();
    MG_HIP(hipEventRecord(R.ev[2], R.stream), "hipEventRecord");
    if(c.st.denoise > 0) {
      for(int l = 0; l < 4; l++) { const int g = multi ? DIRECT_GROW[l] : 0; runStage(M, R, c.st, f, RT_STAGE_DENOISE_DIRECT, l, y0 - g, y1 + g, H); }
      for(int l = 0; l < 5; l++) { const int g = multi ? INDIRECT_GROW[l] : 0; runStage(M, R, c.st, f, RT_STAGE_DENOISE_INDIRECT, l, h0 - g, h1 + g, Hh); }
    }
    runStage(M, R, c.st, f, RT_STAGE_COMPOSE, 0, y0, y1, H);
    MG_HIP(hipEventRecord(R.ev[3], R.stream), "hipEventRecord");
    MG_HIP(hipStreamSynchronize(R.stream), "sync");
  }
  M.step.wait();
  // ---- 4. result bands to the display rank ----
  if(multi && R.id == 0 && c.gather) {
    pullRows(M, R, RT_BUF_DIRECT_RESULT0 + cur, 0, H, bands, HK_GATHER);
    pullRows(M, R, RT_BUF_INDIRECT_RESULT0 + cur, 0, H, bands, HK_GATHER);
    MG_HIP(hipStreamSynchronize(R.stream), "sync");
  }
  float ms = 0.f;
  if(hipEventElapsedTime(&ms, R.ev[0], R.ev[1]) == hipSuccess) R.tracedMs = ms;
  if(hipEventElapsedTime(&ms, R.ev[2], R.ev[3]) == hipSuccess) R.filterMs = ms;
  for(int k = 0; k < HK_COUNT; k++) R.lastPulled[k] = R.pulled[k];
}

// ==================================================================================================================================================
// frames in flight
// ==================================================================================================================================================
// host side: wait until rank q has RECORDED event `kind` of frame `want` (a stream may only be made to wait for a recorded event)
void hostWaitIssued(rt_mgpu& M, Rank& Q, int kind, int64_t want)
{
  if(want < 0) return;
  int spins = 0;
  while(Q.issued[kind].load(std::memory_order_acquire) < want && !M.abort.load()) {
    if(++spins < 200) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}
void recordIssued(rt_mgpu& M, Rank& R, int kind, int64_t seq, hipStream_t strm)
{
  MG_HIP(hipEventRecord(R.evp[kind][seq % RING], strm), "hipEventRecord");
  R.issued[kind].store(seq, std::memory_order_release);
}
// make `strm` wait for event `kind` of frame `seq` on every rank (self included if `self`)
void waitAll(rt_mgpu& M, Rank& R, hipStream_t strm, int kind, int64_t seq, bool self)
{
  if(seq < 0) return;
  for(int q = 0; q < M.n; q++) {
    if(q == R.id && !self) continue;
    Rank& Q = M.ranks[q];
    if(q != R.id) hostWaitIssued(M, Q, kind, seq);
    MG_HIP(hipStreamWaitEvent(strm, Q.evp[kind][seq % RING], 0), "hipStreamWaitEvent");
  }
}
// the rank's indirect and filter stream (the context's, created here in its order: rt_get_streams), the display rank's copy stream — on the rank's first frame in flight
void ensurePipeStreams(rt_mgpu& M, Rank& R)
{
  if(R.sInd && R.sSide && (R.id != 0 || R.sCopy)) return;
  void* ind = nullptr; void* side = nullptr;
  MG_CHECK(rt_get_streams(R.ctx, nullptr, &ind, &side), "rt_get_streams");
  R.sInd = static_cast<hipStream_t>(ind); R.sSide = static_cast<hipStream_t>(side);
  if(R.id == 0 && !R.sCopy) MG_HIP(hipStreamCreateWithFlags(&R.sCopy, hipStreamNonBlocking), "hipStreamCreate");   // the display rank's gather; the others have no use for a fourth hardware queue
}
void syncRank(rt_mgpu& M, Rank& R)
{
  MG_HIP(hipStreamSynchronize(R.stream), "sync"); if(R.sInd) MG_HIP(hipStreamSynchronize(R.sInd), "sync");
  if(R.sSide) MG_HIP(hipStreamSynchronize(R.sSide), "sync"); if(R.sCopy) MG_HIP(hipStreamSynchronize(R.sCopy), "sync");
}
void harvestTiming(Rank& R, int64_t seq)
{
  if(seq < 0) return;
  hipEvent_t* t = R.tm[seq % RING];
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  if(hipEventElapsedTime(&a, t[0], t[1]) == hipSuccess && hipEventElapsedTime(&b, t[2], t[3]) == hipSuccess) R.tracedMs = a + b;
  if(hipEventElapsedTime(&c, t[4], t[5]) == hipSuccess && hipEventElapsedTime(&d, t[6], t[7]) == hipSuccess) R.filterMs = c + d;
  for(int g = 0; g < Rank::LG_COUNT; g++) {
    float ms = 0.f;
    const uint64_t bytes = R.grpBytes[seq % RING][g];
    if(!bytes) { R.linkBytes[g] = 0; R.linkMs[g] = 0.f; continue; }
    // bytes and milliseconds are reported as a PAIR of one frame: when the events of this frame are not ready (or unreadable) the previous pair stands
    if(hipEventElapsedTime(&ms, R.tl[seq % RING][2 * g], R.tl[seq % RING][2 * g + 1]) == hipSuccess) { R.linkBytes[g] = bytes; R.linkMs[g] = ms; }
  }
}
// bracket a group of pulls on `strm` for the link statistics
struct LinkGroup {
  rt_mgpu& M; Rank& R; hipStream_t strm; int g, slot;
  LinkGroup(rt_mgpu& M_, Rank& R_, hipStream_t s, int g_, int slot_) : M(M_), R(R_), strm(s), g(g_), slot(slot_)
  {
    R.grpBytes[slot][g] = 0; R.grp = g; R.grpSlot = slot;
    (void)hipEventRecord(R.tl[slot][2 * g], strm);
  }
  ~LinkGroup() { (void)hipEventRecord(R.tl[slot][2 * g + 1], strm); R.grp = -1; }
};

// second half of frame p = prevCmd: validate its indirect stage, then (side stream) the indirect filter halo, 5 levels, compose; rank 0 gathers
void finishPrev(rt_mgpu& M, Rank& R, int rotatedFrames /* frames value the G-buffer ids were last rotated for, or -1 */)
{
  if(!R.havePrev) return;
  R.havePrev = false;
  const FrameCmd& c = R.prevCmd;
  const int64_t p = c.seq; const int slot = int(p % RING);
  const int f = c.frames, cur = f & 1, last = cur ^ 1, H = M.H, Hh = H / 2;
  const int y0 = c.bands[R.id], y1 = c.bands[R.id + 1], h0 = std::min(y0 / 2, Hh), h1 = std::min(y1 / 2, Hh);
  const bool multi = M.n > 1;
  const int accWas = R.accSlot;
  R.accSlot = int(p & 1);
  MG_CHECK(rt_set_camera(R.ctx, &c.cam), "rt_set_camera");   // the deferred launches belong to frame p: its camera, not the next frame's
  // ---- validate indirect(p) ----
  MG_CHECK(rt_set_stream(R.ctx, R.sInd), "rt_set_stream");
  int miss = 0;
  MG_CHECK(rt_history_miss_stage(R.ctx, RT_STAGE_INDIRECT, &miss), "rt_history_miss_stage");   // waits for the ind stream only
  if(miss && multi && c.haveHistory) {
    // exact fallback, by this rank alone: the owners keep G(p-1) and the indirect reservoirs of p-1 until they have seen this rank's I(p)
    syncRank(M, R);
    if(rotatedFrames >= 0 && rotatedFrames != f) MG_CHECK(rt_rotate_buffers(R.ctx, rotatedFrames), "rt_rotate_buffers");   // undo: the ids name frame p's buffers again
    pullRowsOn(M, R, R.sInd, RT_BUF_GBUFFER0 + last, 0, H, c.prev, HK_FALLBACK, p >= 1 ? int((p - 1) % RING) : -1);
    pullRowsOn(M, R, R.sInd, RT_BUF_INDIRECT_RESV0 + last, 0, Hh, c.prev, HK_FALLBACK);
    MG_CHECK(rt_set_history_rows(R.ctx, 0, H), "rt_set_history_rows");
    runStage(M, R, c.st, f, RT_STAGE_INDIRECT, 0, h0, h1, Hh);
    MG_CHECK(rt_history_miss_stage(R.ctx, RT_STAGE_INDIRECT, &miss), "rt_history_miss_stage");   // clears the flag, waits
    if(rotatedFrames >= 0 && rotatedFrames != f) MG_CHECK(rt_rotate_buffers(R.ctx, rotatedFrames), "rt_rotate_buffers");   // redo
    M.fallbacks++;
  }
  MG_HIP(hipEventRecord(R.tm[slot][3], R.sInd), "hipEventRecord");
  recordIssued(M, R, Rank::E_I, p, R.sInd);
  // ---- side stream: indirect filters + compose ----
  MG_CHECK(rt_set_stream(R.ctx, R.sSide), "rt_set_stream");
  waitAll(M, R, R.sSide, Rank::E_I, p, true);
  MG_HIP(hipEventRecord(R.tm[slot][6], R.sSide), "hipEventRecord");
  if(multi && c.st.denoise > 0) { LinkGroup lg(M, R, R.sSide, Rank::LG_FILTER_INDIRECT, slot); pullFilterHaloIndirect(M, R, R.sSide, f, h0, h1, c.bands); }
  recordIssued(M, R, Rank::E_XI, p, R.sSide);
  if(c.st.denoise > 0) {
    runStage(M, R, c.st, f, RT_STAGE_DENOISE_INDIRECT, 0, h0 - (multi ? INDIRECT_GROW[0] : 0), h1 + (multi ? INDIRECT_GROW[0] : 0), Hh);   // IndA -> IndB
    if(multi) waitAll(M, R, R.sSide, Rank::E_XI, p, false);   // level 1 writes IndA: every neighbour has its copy of this band's noisy rows
    for(int l = 1; l < 5; l++) { const int g = multi ? INDIRECT_GROW[l] : 0; runStage(M, R, c.st, f, RT_STAGE_DENOISE_INDIRECT, l, h0 - g, h1 + g, Hh); }
  }
  runStage(M, R, c.st, f, RT_STAGE_COMPOSE, 0, y0, y1, H);
  MG_HIP(hipEventRecord(R.tm[slot][7], R.sSide), "hipEventRecord");
  recordIssued(M, R, Rank::E_DONE, p, R.sSide);
  // ---- display rank: result bands on the copy stream, off every other stream's path ----
  if(R.id == 0) {
    if(multi && c.gather) {
      waitAll(M, R, R.sCopy, Rank::E_DONE, p, true);
      pullRowsOn(M, R, R.sCopy, RT_BUF_DIRECT_RESULT0 + cur, 0, H, c.bands, HK_GATHER);
      pullRowsOn(M, R, R.sCopy, RT_BUF_INDIRECT_RESULT0 + cur, 0, H, c.bands, HK_GATHER);
    }
    recordIssued(M, R, Rank::E_G, p, R.sCopy);
  }
  for(int k = 0; k < HK_COUNT; k++) R.lastPulled[k] = R.pulledBy[p & 1][k];
  R.accSlot = accWas;
}

// ---- A. direct(f) on the main stream ------------------------------------------------------------------------------------------------------------
void pipeDirect(rt_mgpu& M, Rank& R, const FrameCmd& c)
{
  const int64_t s = c.seq; const int slot = int(s % RING);
  const int f = c.frames, cur = f & 1, last = cur ^ 1, H = M.H;
  const int y0 = c.bands[R.id], y1 = c.bands[R.id + 1];
  const bool multi = M.n > 1;
  R.aIssued = s;
  if(R.mode != 2) { MG_CHECK(rt_set_overlap(R.ctx, 2), "rt_set_overlap"); R.mode = 2; }   // short LDS traversal stacks: kernels of two frames share the CUs
  R.accSlot = int(s & 1);
  for(auto& p : R.pulledBy[s & 1]) p = 0;
  harvestTiming(R, s - 3);
  for(auto& b : R.grpBytes[slot]) b = 0;
  MG_CHECK(rt_rotate_buffers(R.ctx, f), "rt_rotate_buffers");   // G-buffer x3, motion x2: direct(f) must not overwrite what indirect(f-1) still reads
  R.rotatedFor = f;
  { void* p = nullptr; size_t b = 0, pitch = 0; MG_CHECK(rt_device_ptr(R.ctx, RT_BUF_GBUFFER0 + cur, &p, &b, &pitch), "rt_device_ptr"); R.gptr[slot] = p; }
  waitAll(M, R, R.stream, Rank::E_D, s - 1, false);     // history rows come from the neighbours' direct(f-1); they have also pulled what direct(f) overwrites
  waitAll(M, R, R.stream, Rank::E_I, s - 2, true);      // G(f-3) / motion(f-2): read by indirect(f-2), here and (history rows) next door
  waitAll(M, R, R.stream, Rank::E_DONE, s - 2, true);   // result image / filter scratch of f-2
  if(multi && c.gather && s >= 2) { hostWaitIssued(M, M.ranks[0], Rank::E_G, s - 2); MG_HIP(hipStreamWaitEvent(R.stream, M.ranks[0].evp[Rank::E_G][(s - 2) % RING], 0), "hipStreamWaitEvent"); }
  MG_CHECK(rt_set_stream(R.ctx, R.stream), "rt_set_stream");
  MG_CHECK(rt_set_camera(R.ctx, &c.cam), "rt_set_camera");
  if(multi && c.haveHistory) {
    LinkGroup lg(M, R, R.stream, Rank::LG_HISTORY, slot);
    const int gs = s >= 1 ? int((s - 1) % RING) : -1;   // (first frame after a restart: everything is drained, the boundary ids name last frame's buffers on every rank)
    const int HIST_HALO = c.histHalo;
    // The G-buffer rows next to the band came with the filter halo of f-1 when the partition did not move — IF that pull (this rank's SIDE stream) has been
    // issued: a direct stage issued ahead of the previous frame's filters (look-ahead) pulls the rows itself.
    const bool haloHasThem = c.samePartition && HIST_HALO <= HALO_GBUFFER_FULL && s >= 1 && R.issued[Rank::E_X].load(std::memory_order_acquire) >= s - 1;
    if(haloHasThem) MG_HIP(hipStreamWaitEvent(R.stream, R.evp[Rank::E_X][(s - 1) % RING], 0), "hipStreamWaitEvent");
    else pullRowsOn(M, R, R.stream, RT_BUF_GBUFFER0 + last, y0 - HIST_HALO, y1 + HIST_HALO, c.prev, HK_HISTORY, gs);
    for(int buf : {RT_BUF_DIRECT_RESV0 + last, RT_BUF_LIGHT_ID0 + last}) pullRowsOn(M, R, R.stream, buf, y0 - HIST_HALO, y1 + HIST_HALO, c.prev, HK_HISTORY);
    for(int buf : {RT_BUF_DIRECT_RESV0 + cur, RT_BUF_LIGHT_ID0 + cur, int(RT_BUF_DIRECT_RESV_TEMP)}) pullRowsOn(M, R, R.stream, buf, y0, y1, c.prev, HK_MOVED);
  }
  MG_CHECK(rt_set_history_rows(R.ctx, multi ? std::max(0, y0 - c.histHalo) : 0, multi ? std::min(H, y1 + c.histHalo) : H), "rt_set_history_rows");
  MG_HIP(hipEventRecord(R.tm[slot][0], R.stream), "hipEventRecord");
  runStage(M, R, c.st, f, RT_STAGE_DIRECT, 0, y0, y1, H);
  MG_HIP(hipEventRecord(R.tm[slot][1], R.stream), "hipEventRecord");
}

// ---- C. validate direct(f): the host waits for ITS OWN main stream; a miss is repaired by this rank alone -------------------------------------------------
void pipeValidateDirect(rt_mgpu& M, Rank& R, const FrameCmd& c)
{
  const int64_t s = c.seq;
  const int f = c.frames, last = (f & 1) ^ 1, H = M.H;
  const int y0 = c.bands[R.id], y1 = c.bands[R.id + 1];
  const bool multi = M.n > 1;
  R.accSlot = int(s & 1);
  MG_CHECK(rt_set_stream(R.ctx, R.stream), "rt_set_stream");
  MG_CHECK(rt_set_camera(R.ctx, &c.cam), "rt_set_camera");
  int miss = 0;
  MG_CHECK(rt_history_miss_stage(R.ctx, RT_STAGE_DIRECT, &miss), "rt_history_miss_stage");
  if(miss && multi && c.haveHistory) {
    const int gs = s >= 1 ? int((s - 1) % RING) : -1;
    pullRowsOn(M, R, R.stream, RT_BUF_GBUFFER0 + last, 0, H, c.prev, HK_FALLBACK, gs);
    for(int buf : {RT_BUF_DIRECT_RESV0 + last, RT_BUF_LIGHT_ID0 + last}) pullRowsOn(M, R, R.stream, buf, 0, H, c.prev, HK_FALLBACK);
    MG_CHECK(rt_set_history_rows(R.ctx, 0, H), "rt_set_history_rows");
    runStage(M, R, c.st, f, RT_STAGE_DIRECT, 0, y0, y1, H);
    MG_CHECK(rt_history_miss_stage(R.ctx, RT_STAGE_DIRECT, &miss), "rt_history_miss_stage");   // clears, waits
    M.fallbacks++;
  }
  recordIssued(M, R, Rank::E_D, s, R.stream);
}

// ---- D. indirect(f) on the ind stream, E. direct filters of frame f on the side stream (after the second half of f-1) ------------------------------------
void pipeIndirectAndDirectFilters(rt_mgpu& M, Rank& R, const FrameCmd& c)
{
  const int64_t s = c.seq; const int slot = int(s % RING);
  const int f = c.frames, cur = f & 1, last = cur ^ 1, H = M.H, Hh = H / 2;
  const int y0 = c.bands[R.id], y1 = c.bands[R.id + 1], h0 = std::min(y0 / 2, Hh), h1 = std::min(y1 / 2, Hh);
  const bool multi = M.n > 1;
  R.accSlot = int(s & 1);
  MG_CHECK(rt_set_camera(R.ctx, &c.cam), "rt_set_camera");
  // if the next frame's direct stage has already been issued (look-ahead), the boundary ids of the rotating buffers name ITS buffers: name this frame's again
  const bool ahead = R.rotatedFor != f;
  if(ahead) MG_CHECK(rt_rotate_buffers(R.ctx, R.rotatedFor), "rt_rotate_buffers");
  MG_HIP(hipStreamWaitEvent(R.sInd, R.evp[Rank::E_D][slot], 0), "hipStreamWaitEvent");
  waitAll(M, R, R.sInd, Rank::E_I, s - 1, false);       // the neighbours' indirect reservoirs of f-1
  waitAll(M, R, R.sInd, Rank::E_XI, s - 2, false);      // the noisy-indirect buffer of this parity (frame f-2): pulled next door ...
  if(s >= 2) MG_HIP(hipStreamWaitEvent(R.sInd, R.evp[Rank::E_DONE][(s - 2) % RING], 0), "hipStreamWaitEvent");   // ... and filtered here
  MG_CHECK(rt_set_stream(R.ctx, R.sInd), "rt_set_stream");
  if(multi && c.haveHistory) {
    LinkGroup lg(M, R, R.sInd, Rank::LG_HISTORY_IND, slot);
    pullRowsOn(M, R, R.sInd, RT_BUF_INDIRECT_RESV0 + last, h0 - c.histHalo / 2, h1 + c.histHalo / 2, c.prev, HK_HISTORY);
    pullRowsOn(M, R, R.sInd, RT_BUF_INDIRECT_RESV0 + cur, h0, h1, c.prev, HK_MOVED);
  }
  MG_CHECK(rt_set_history_rows(R.ctx, multi ? std::max(0, y0 - c.histHalo) : 0, multi ? std::min(H, y1 + c.histHalo) : H), "rt_set_history_rows");
  MG_HIP(hipEventRecord(R.tm[slot][2], R.sInd), "hipEventRecord");
  runStage(M, R, c.st, f, RT_STAGE_INDIRECT, 0, h0, h1, Hh);
  MG_CHECK(rt_set_stream(R.ctx, R.sSide), "rt_set_stream");
  waitAll(M, R, R.sSide, Rank::E_D, s, true);
  MG_HIP(hipEventRecord(R.tm[slot][4], R.sSide), "hipEventRecord");
  if(multi && c.st.denoise > 0) {
    LinkGroup lg(M, R, R.sSide, Rank::LG_FILTER_DIRECT, slot);
    pullFilterHaloDirect(M, R, R.sSide, cur, y0, y1, c.bands, slot);
  }
  recordIssued(M, R, Rank::E_X, s, R.sSide);
  if(c.st.denoise > 0) {
    for(int l = 0; l < 3; l++) { const int g = multi ? DIRECT_GROW[l] : 0; runStage(M, R, c.st, f, RT_STAGE_DENOISE_DIRECT, l, y0 - g, y1 + g, H); }
    if(multi) waitAll(M, R, R.sSide, Rank::E_X, s, false);   // level 3 writes the result image: every neighbour has its copy of this band's noisy rows
    runStage(M, R, c.st, f, RT_STAGE_DENOISE_DIRECT, 3, y0, y1, H);
  } else if(multi) waitAll(M, R, R.sSide, Rank::E_X, s, false);
  MG_HIP(hipEventRecord(R.tm[slot][5], R.sSide), "hipEventRecord");
  if(ahead) MG_CHECK(rt_rotate_buffers(R.ctx, R.rotatedFor), "rt_rotate_buffers");
  R.prevCmd = c; R.havePrev = true;
}

// look-ahead only when it cannot block this rank's host thread: every event pipeDirect(next) waits for has been recorded
bool peersReadyForDirect(rt_mgpu& M, Rank& R, const FrameCmd& next)
{
  const int64_t s = next.seq;
  for(int q = 0; q < M.n; q++) {
    if(q == R.id) continue;
    Rank& Q = M.ranks[q];
    if(Q.issued[Rank::E_D].load(std::memory_order_acquire) < s - 1 || Q.issued[Rank::E_I].load(std::memory_order_acquire) < s - 2 ||
       Q.issued[Rank::E_DONE].load(std::memory_order_acquire) < s - 2) return false;
  }
  if(M.n > 1 && next.gather && M.ranks[0].issued[Rank::E_G].load(std::memory_order_acquire) < s - 2) return false;
  return true;
}

// One frame of the frames-in-flight schedule.  `next`: the frame after this one if the application has already queued it — its direct stage is then issued
// as soon as this frame's direct stage is validated, BEFORE this frame's indirect stage and filters, so the main stream never waits for the host.
void framePipelined(rt_mgpu& M, Rank& R, const FrameCmd& c)
{
  const int64_t s = c.seq;
  if(c.solo >= 0 && c.solo != R.id) {   // measurement: this rank sits the frame out; its flags must not hold the others back
    R.havePrev = false;
    for(auto& a : R.issued) a.store(s, std::memory_order_release);
    return;
  }
  ensurePipeStreams(M, R);
  if(R.rc != RT_OK) return;
  if(R.aIssued != s) pipeDirect(M, R, c);
  // second half of frame f-1, issued while direct(f) runs (the ids of the rotating buffers name frame f's: finishPrev undoes that for a re-run only)
  finishPrev(M, R, R.rotatedFor);

  pipeValidateDirect(M, R, c);

  // look-ahead: has the application queued the next frame by now?  (looked up here, a direct stage after this frame was taken from the queue)
  Cmd nextCmd; bool haveNext = false;
  {
    std::lock_guard<std::mutex> l(R.qm);
    if(R.q.size() >= 2 && R.q[1].kind == Cmd::FRAMEP) { nextCmd = R.q[1]; haveNext = true; }
  }
  const FrameCmd* next = haveNext ? &nextCmd.f : nullptr;
  if(next && next->seq == s + 1 && !(next->solo >= 0 && next->solo != R.id) && peersReadyForDirect(M, R, *next)) pipeDirect(M, R, *next);
  pipeIndirectAndDirectFilters(M, R, c);
}

void drainRank(rt_mgpu& M, Rank& R)
{
  finishPrev(M, R, -1);
  syncRank(M, R);
  if(R.ctx) { MG_CHECK(rt_set_stream(R.ctx, R.stream), "rt_set_stream"); MG_CHECK(rt_sync(R.ctx), "rt_sync"); }
}

void worker(rt_mgpu* Mp, int id)
{
  rt_mgpu& M = *Mp;
  Rank& R = M.ranks[id];
  (void)hipSetDevice(R.dev);
  for(;;) {
    Cmd cmd;
    {
      std::unique_lock<std::mutex> l(R.qm);
      R.qcv.wait(l, [&] { return !R.q.empty(); });
      cmd = R.q.front();
    }
    if(cmd.kind == Cmd::QUIT) break;
    switch(cmd.kind) {
      case Cmd::UPLOAD:
        MG_CHECK(rt_upload_scene(R.ctx, M.desc), "rt_upload_scene");
        if(R.rc == RT_OK) { std::lock_guard<std::mutex> one(M.buildTurn); MG_CHECK(rt_build_accel(R.ctx), "rt_build_accel"); }   // the host build is multi-threaded itself
        break;
      case Cmd::RESIZE:
        MG_CHECK(rt_resize(R.ctx, M.W, M.H), "rt_resize");
        for(int par = 0; par < 2 && R.rc == RT_OK; par++) {
          void* p = nullptr; size_t b = 0, pitch = 0;
          MG_CHECK(rt_select_frame(R.ctx, par), "rt_select_frame");
          MG_CHECK(rt_device_ptr(R.ctx, RT_BUF_DENOISE_IND_A, &p, &b, &pitch), "rt_device_ptr");
          R.indA[par] = p;
        }
        break;
      case Cmd::FRAME: frameOnRank(M, R, cmd.f); break;
      case Cmd::FRAMEP: framePipelined(M, R, cmd.f); break;
      case Cmd::DRAIN: drainRank(M, R); break;
      default: break;
    }
    {
      std::lock_guard<std::mutex> l(R.qm);
      R.q.pop_front();
    }
    R.qcv.notify_all();
    {
      std::lock_guard<std::mutex> l(M.pm);
      M.pending--;
    }
    M.pcv.notify_all();
  }
}

void post(rt_mgpu* M, const Cmd& cmd, size_t maxQueued)
{
  for(Rank& R : M->ranks) {
    std::unique_lock<std::mutex> l(R.qm);
    R.qcv.wait(l, [&] { return R.q.size() < maxQueued; });
    R.q.push_back(cmd);
    { std::lock_guard<std::mutex> p(M->pm); M->pending++; }
    l.unlock();
    R.qcv.notify_all();
  }
}
void waitIdle(rt_mgpu* M)
{
  std::unique_lock<std::mutex> l(M->pm);
  M->pcv.wait(l, [&] { return M->pending == 0; });
}
int firstError(rt_mgpu* M)
{
  for(const Rank& R : M->ranks) if(R.rc != RT_OK) return R.rc;
  return RT_OK;
}
// run `kind` on every rank and wait for it
int dispatch(rt_mgpu* M, Cmd::Kind kind, const FrameCmd* f = nullptr)
{
  Cmd c; c.kind = kind; if(f) c.f = *f;
  post(M, c, 1u << 20);
  waitIdle(M);
  return firstError(M);
}
// finish every frame in flight (their deferred second halves included)
int drainAll(rt_mgpu* M)
{
  if(!M->pipeActive) { waitIdle(M); return firstError(M); }
  M->pipeActive = false;
  return dispatch(M, Cmd::DRAIN);
}
struct DeviceGuard {   // entry points called on the application's thread leave its current device alone
  int dev = -1;
  DeviceGuard() { if(hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~DeviceGuard() { if(dev >= 0) (void)hipSetDevice(dev); }
};
// Every entry point starts here.  An error of an earlier call (reported by that call) must not poison this one: the workers are let run dry (with `abort`
// set they no longer wait for each other), every device is drained, and the per-rank error codes, the abort flag and the frames-in-flight bookkeeping are
// reset; the screen-space history is dropped because a frame that failed half-way leaves it inconsistent.  (Round-3 advisor: the codes were sticky, and a
// context that had seen one recoverable failure — e.g. rt_mgpu_upload_scene with a bad description — stayed unusable until it was destroyed.)
void beginCall(rt_mgpu* M)
{
  if(firstError(M) != RT_OK || M->abort.load()) {
    waitIdle(M);
    for(Rank& R : M->ranks) {
      (void)hipSetDevice(R.dev);
      (void)hipDeviceSynchronize();
      R.rc = RT_OK;
      for(auto& a : R.issued) a.store(-1);
      R.havePrev = false; R.aIssued = -1;
    }
    M->abort = false;
    M->pipeActive = false; M->seq = 0; M->lastFrames = -2;
    M->haveHistory = false;
  }
  std::lock_guard<std::mutex> l(M->errLock);
  M->err.clear();
}

// the balancer starts over: the smoothed per-rank times were measured on bands that no longer exist (a stale rankMs would steer the first diffusion steps)
void resetBalancer(rt_mgpu* M) { M->balanceFrames = 0; M->rankMs.clear(); M->boundaryCool.clear(); }

void equalBands(rt_mgpu* M)
{
  resetBalancer(M);
  const int stripes = (M->H + 15) / 16;
  M->bands.assign(size_t(M->n) + 1, 0);
  for(int r = 0; r <= M->n; r++) M->bands[size_t(r)] = std::min(M->H, 16 * int((int64_t(stripes) * r + M->n - 1) / M->n));
  M->bands[size_t(M->n)] = M->H;
}

// next frame's boundaries from the smoothed per-stripe costs: equalise the summed cost, move at most 2 stripes per frame
void rebalance(rt_mgpu* M)
{
  const int stripes = (M->H + 15) / 16, n = M->n;
  if(int(M->stripeCost.size()) != stripes) M->stripeCost.assign(size_t(stripes), 1.0f);
  for(int r = 0; r < n; r++) {
    const int a = M->bands[size_t(r)] / 16, b = (M->bands[size_t(r) + 1] + 15) / 16;
    if(b <= a) continue;
    const float per = std::max(1e-4f, M->ranks[size_t(r)].tracedMs + M->ranks[size_t(r)].filterMs) / float(b - a);
    for(int s = a; s < b; s++) M->stripeCost[size_t(s)] = 0.5f * M->stripeCost[size_t(s)] + 0.5f * per;
  }
  if(!M->balance || n == 1) return;
  // Two phases.  The first frames plan from the per-stripe cost model (a rank's time spread over its stripes): coarse, fast.  A rank's time is not the sum of its
  // stripes' costs, though — a band that holds horizon rows takes what its slowest tile takes however few rows it has — so the model settles with the slowest rank
  // 1.35x the fastest.  From then on the boundaries DIFFUSE: a boundary moves one stripe towards the slower of its two ranks when their smoothed times differ
  // by more than 6 %; the two ranks it separates are then re-measured on their new bands before either of their boundaries moves again.
  const int MODEL_FRAMES = 10, HOLD = 6;   // HOLD: frames until a rank's measured time belongs to its new band (the frames-in-flight schedule harvests timings 3 frames late)
  if(int(M->rankMs.size()) != n) { M->rankMs.assign(size_t(n), 0.f); M->boundaryCool.assign(size_t(n), 0); M->balanceFrames = 0; }
  std::vector<int>& hold = M->boundaryCool;   // per RANK: frames its band still has to be re-measured
  for(int r = 0; r < n; r++) {
    const float t = M->ranks[size_t(r)].tracedMs + M->ranks[size_t(r)].filterMs;
    if(hold[size_t(r)] > 0) { if(--hold[size_t(r)] == 0) M->rankMs[size_t(r)] = t; }
    else M->rankMs[size_t(r)] = M->rankMs[size_t(r)] > 0.f ? 0.5f * M->rankMs[size_t(r)] + 0.5f * t : t;
  }
  std::vector<int> nb(size_t(n) + 1, 0);
  if(M->balanceFrames++ < MODEL_FRAMES) {
    rt_mgpu_plan_bands(M->H, n, M->stripeCost.data(), M->bands.data(), 2, nb.data());
    if(nb != M->bands) for(int r = 0; r < n; r++) hold[size_t(r)] = HOLD;
    M->bands = nb;
    return;
  }
  nb = M->bands;
  // the boundaries of the slowest ranks go first (then the larger imbalance): scanning top to bottom would let an upper boundary that keeps moving starve the one below it
  std::vector<std::pair<std::pair<float, float>, int>> cand;
  for(int k = 1; k < n; k++) {
    const float a = std::max(1e-9f, M->rankMs[size_t(k) - 1]), b = std::max(1e-9f, M->rankMs[size_t(k)]);
    if(std::max(a, b) > std::min(a, b) * 1.06f) cand.push_back({{-std::max(a, b), -std::max(a, b) / std::min(a, b)}, k});
  }
  std::sort(cand.begin(), cand.end());
  for(const auto& c : cand) {
    const int k = c.second;
    if(hold[size_t(k) - 1] > 0 || hold[size_t(k)] > 0) continue;
    const float a = M->rankMs[size_t(k) - 1], b = M->rankMs[size_t(k)];
    if(a > b && nb[size_t(k)] - nb[size_t(k) - 1] > 16) { nb[size_t(k)] -= 16; hold[size_t(k) - 1] = hold[size_t(k)] = HOLD; }
    else if(b > a && nb[size_t(k) + 1] - nb[size_t(k)] > 16) { nb[size_t(k)] += 16; hold[size_t(k) - 1] = hold[size_t(k)] = HOLD; }
  }
  M->bands = nb;
}

void destroyRank(Rank& R)
{
  (void)hipSetDevice(R.dev);
  if(R.ctx) rt_destroy(R.ctx);   // (takes the rank's main / indirect / filter streams with it: they are the context's — round 6)
  for(auto& e : R.ev) if(e) (void)hipEventDestroy(e);
  for(auto& k : R.evp) for(auto& e : k) if(e) (void)hipEventDestroy(e);
  for(auto& k : R.tm) for(auto& e : k) if(e) (void)hipEventDestroy(e);
  for(auto& k : R.tl) for(auto& e : k) if(e) (void)hipEventDestroy(e);
  if(R.sCopy) (void)hipStreamDestroy(R.sCopy);
  R.stream = R.sInd = R.sSide = R.sCopy = nullptr;
  R.ctx = nullptr;
}

}  // namespace

extern "C" {

// Band boundaries that equalise the summed per-stripe cost (a stripe = 16 rows): every rank keeps at least one stripe; with prevBands a
// boundary moves at most maxMoveStripes stripes.  Pure host arithmetic (no device needed); the same rule as tiled.plan_bands.
int rt_mgpu_plan_bands(int height, int numRanks, const float* stripeCost, const int* prevBands, int maxMoveStripes, int* outBands)
{
  const int stripes = (height + 15) / 16, n = numRanks;
  if(height <= 0 || n < 1 || !stripeCost || !outBands || stripes < n) return RT_ERR_INVALID_ARG;
  double total = 0; for(int s = 0; s < stripes; s++) total += std::max(1e-9, double(stripeCost[s]));
  std::vector<int> nb(size_t(n) + 1, 0);
  double acc = 0; int r = 1;
  for(int s = 0; s < stripes && r < n; s++) {
    acc += std::max(1e-9, double(stripeCost[s]));
    while(r < n && acc >= total * r / n) { nb[size_t(r)] = s + 1; r++; }
  }
  for(; r < n; r++) nb[size_t(r)] = stripes;
  nb[size_t(n)] = stripes;
  for(int k = 1; k < n; k++) {
    int v = nb[size_t(k)];
    if(prevBands && maxMoveStripes >= 0) { const int old = prevBands[k] / 16; v = std::max(old - maxMoveStripes, std::min(old + maxMoveStripes, v)); }
    v = std::max(v, nb[size_t(k) - 1] + 1);                 // every rank keeps at least one stripe
    v = std::min(v, stripes - (n - k));
    nb[size_t(k)] = v;
  }
  for(int k = 0; k <= n; k++) outBands[k] = std::min(height, nb[size_t(k)] * 16);
  outBands[n] = height;
  return RT_OK;
}

int rt_mgpu_create(rt_mgpu** out, int numRanks, const int* devices)
{
  if(!out || numRanks < 1 || numRanks > MAX_RANKS) return RT_ERR_INVALID_ARG;
  DeviceGuard guard;
  rt_mgpu* M = new(std::nothrow) rt_mgpu();
  if(!M) return RT_ERR_OOM;
  M->n = numRanks;
#if RT_TEST_HOOKS   // (compiled into measurement / test builds only — restir_amd.build.build_hip(variant="testhooks", extra_flags=["-DRT_TEST_HOOKS=1"]): the product library cannot
                    //  be told to damage live halo data through an environment variable; advisor finding of round 5)
  if(const char* e = getenv("RESTIR_TEST_CORRUPT_HALO")) M->corruptHalo = e[0] == '1';
#endif
  M->ranks = std::vector<Rank>(size_t(numRanks));
  auto bail = [&](int rc) { for(Rank& R : M->ranks) destroyRank(R); delete M; return rc; };
  for(int r = 0; r < numRanks; r++) {
    Rank& R = M->ranks[size_t(r)];
    R.id = r; R.dev = devices ? devices[r] : r;
    // Round 6: a rank's three streams are its CONTEXT's streams (rt_get_streams): the main stream rt_create makes, and — created on the rank's first frame in flight
    // (ensurePipeStreams), filter stream first, then the indirect stream, nothing in between — the two others, i.e. the creation order and the levels the single-GPU
    // schedule was tuned on.  Before, every rank created three more streams next to its context's unused one, all of them up front: with N ranks emulated on one device
    // (bench.py --emulate-world) 8 x 4-5 streams were alive while ONE rank was being timed "alone", and a stream's worth depends on how many the process created before
    // it (profiles/r05_prio_by_config_ab.txt §2).  Levels: RESTIR_MGPU_PRIO = three characters over {-, 0, +} for the main / indirect / filter stream; the default was
    // re-measured per rank in fresh processes on the real scene (profiles/r06_mgpu_streams_ab.txt; round 3 had chosen "main high" on the lite scene under the old layout).
    int levels[3] = {MGPU_PRIO_DEFAULT[0], MGPU_PRIO_DEFAULT[1], MGPU_PRIO_DEFAULT[2]};
    if(const char* e = getenv("RESTIR_MGPU_PRIO")) if(strlen(e) == 3 && strspn(e, "-0+") == 3) for(int i = 0; i < 3; i++) levels[i] = e[i] == '+' ? 1 : (e[i] == '-' ? -1 : 0);
    for(int i = 0; i < 3; i++) M->levels[i] = levels[i];
    int rc = rtCreateWithLevels(&R.ctx, R.dev, levels);
    if(rc != RT_OK) return bail(rc);
    (void)hipSetDevice(R.dev);
    void* mainStream = nullptr;
    bool ok = rt_get_streams(R.ctx, &mainStream, nullptr, nullptr) == RT_OK && mainStream;
    R.stream = static_cast<hipStream_t>(mainStream);
    for(auto& e : R.ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    for(auto& k : R.evp) for(auto& e : k) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    for(auto& k : R.tm) for(auto& e : k) ok = ok && hipEventCreate(&e) == hipSuccess;
    for(auto& k : R.tl) for(auto& e : k) ok = ok && hipEventCreate(&e) == hipSuccess;
    if(!ok) return bail(RT_ERR_HIP);
    rt_set_stream(R.ctx, R.stream);
    rt_set_overlap(R.ctx, 0);   // stages are issued one by one through rt_run_stage
  }
  // direct xGMI copies between distinct devices (already-enabled is not an error)
  M->peerOk.assign(size_t(numRanks) * size_t(numRanks), 0);
  for(int a = 0; a < numRanks; a++)
    for(int b = 0; b < numRanks; b++) {
      const int da = M->ranks[size_t(a)].dev, db = M->ranks[size_t(b)].dev;
      if(da == db) { M->peerOk[size_t(a) * size_t(numRanks) + size_t(b)] = 1; continue; }
      int can = 0;
      if(hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) {
        (void)hipSetDevice(da);
        const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
        (void)hipGetLastError();
        if(e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) M->peerOk[size_t(a) * size_t(numRanks) + size_t(b)] = 1;
      }
    }
  M->step.init(numRanks);
  for(int r = 0; r < numRanks; r++) M->ranks[size_t(r)].th = std::thread(worker, M, r);
  *out = M;
  return RT_OK;
}

int rt_mgpu_destroy(rt_mgpu* M)
{
  if(!M) return RT_ERR_INVALID_ARG;
  DeviceGuard guard;
  (void)drainAll(M);
  Cmd quit; quit.kind = Cmd::QUIT;
  for(Rank& R : M->ranks) { { std::lock_guard<std::mutex> l(R.qm); R.q.push_back(quit); } R.qcv.notify_all(); }
  for(Rank& R : M->ranks) {
    if(R.th.joinable()) R.th.join();
    destroyRank(R);
  }
  delete M;
  return RT_OK;
}

int rt_mgpu_upload_scene(rt_mgpu* M, const rt_scene_desc* d)
{
  if(!M || !d) return RT_ERR_INVALID_ARG;
  DeviceGuard guard;
  beginCall(M);
  (void)drainAll(M);
  M->desc = d;
  const int rc = dispatch(M, Cmd::UPLOAD);
  M->desc = nullptr; M->haveHistory = false;
  resetBalancer(M); M->stripeCost.clear();   // another scene: another cost per stripe
  return rc;
}

int rt_mgpu_resize(rt_mgpu* M, int w, int h)
{
  if(!M || w <= 0 || h <= 0) return RT_ERR_INVALID_ARG;
  if((h + 15) / 16 < M->n) { M->err = "rt_mgpu_resize: fewer 16-row stripes than ranks"; return RT_ERR_INVALID_ARG; }
  DeviceGuard guard;
  beginCall(M);
  (void)drainAll(M);
  M->W = w; M->H = h;
  const int rc = dispatch(M, Cmd::RESIZE);
  equalBands(M);
  M->prevBands = M->bands;
  M->stripeCost.clear(); M->rankMs.clear();
  M->haveHistory = false;
  for(Rank& R : M->ranks) for(auto& g : R.gptr) g = nullptr;
  return rc;
}

int rt_mgpu_set_camera(rt_mgpu* M, const rt_scene_camera* cam)
{
  if(!M || !cam) return RT_ERR_INVALID_ARG;
  M->cam = *cam;
  return RT_OK;
}

int rt_mgpu_render_frame(rt_mgpu* M, const rt_state* st, int frames)
{
  if(!M || !st) return RT_ERR_INVALID_ARG;
  if(M->W == 0) { M->err = "rt_mgpu_render_frame: rt_mgpu_resize has not been called"; return RT_ERR_NO_TARGET; }
  DeviceGuard guard;
  beginCall(M);
  const bool spatial = st->ReSTIRState == RT_RESTIR_SPATIAL || st->ReSTIRState == RT_RESTIR_SPATIOTEMPORAL;
  const bool pipe = M->pipeline && !M->serialize && !(spatial && M->n > 1);
  // the frames in flight assume consecutive frames (ping-pong parity, rotating G-buffers): anything else restarts the pipeline
  if(M->pipeActive && (!pipe || frames != M->lastFrames + 1)) { const int rc = drainAll(M); if(rc != RT_OK) return rc; }
  FrameCmd c;
  c.st = *st; c.frames = frames; c.cam = M->cam; c.haveHistory = M->haveHistory; c.gather = M->gatherResults; c.solo = M->solo;
  for(int r = 0; r <= M->n; r++) { c.bands[r] = M->bands[size_t(r)]; c.prev[r] = M->prevBands[size_t(r)]; }
  {  // history halo of this frame (see HIST_HALO_MIN): widened after a fallback, narrowed again after a calm stretch
    const uint32_t fb = M->fallbacks.load();
    if(fb != M->fallbacksSeen) { M->fallbacksSeen = fb; M->calmFrames = 0; M->histHalo = std::min(HIST_HALO_MAX, M->histHalo * 2); }
    else if(++M->calmFrames >= HIST_HALO_CALM) { M->calmFrames = 0; M->histHalo = std::max(HIST_HALO_MIN, M->histHalo / 2); }
    c.histHalo = M->histHalo;
    c.samePartition = M->haveHistory && M->lastDenoise && M->bands == M->prevBands;
    M->lastDenoise = st->denoise > 0;
  }
  int rc = RT_OK;
  if(pipe) {
    if(!M->pipeActive) { M->seq = 0; for(Rank& R : M->ranks) { for(auto& a : R.issued) a.store(-1); R.havePrev = false; R.aIssued = -1; } M->pipeActive = true; }
    c.seq = M->seq++;
    Cmd cmd; cmd.kind = Cmd::FRAMEP; cmd.f = c;
    post(M, cmd, 2);               // at most two frames queued per rank: the call returns while the frame is being issued
    rc = firstError(M);
  } else {
    rc = dispatch(M, Cmd::FRAME, &c);
    // the barrier schedule does not bracket its pulls: rt_mgpu_get_link_stats reports zeros rather than the pair of some earlier pipelined frame
    for(Rank& R : M->ranks) for(int g = 0; g < Rank::LG_COUNT; g++) { R.linkMs[g] = 0.f; R.linkBytes[g] = 0; }
  }
  M->lastFrames = frames;
  // statistics of the latest finished frame, then the partition of the next one
  rt_mgpu_stats& S = M->stats;
  S.numRanks = M->n;
  if(rc != RT_OK) return rc;
  S.frames++;
  S.haloBytes = 0;
  for(auto& k : S.haloBytesKind) k = 0;
  S.historyFallbacks = M->fallbacks.load();
  for(int r = 0; r < M->n; r++) {
    S.bandBegin[r] = M->bands[size_t(r)]; S.bandEnd[r] = M->bands[size_t(r) + 1];
    S.tracedMs[r] = M->ranks[size_t(r)].tracedMs; S.filterMs[r] = M->ranks[size_t(r)].filterMs;
    for(int k = 0; k < HK_COUNT; k++) { S.haloBytes += M->ranks[size_t(r)].lastPulled[k]; S.haloBytesKind[k] += M->ranks[size_t(r)].lastPulled[k]; S.haloBytesRankKind[r][k] = M->ranks[size_t(r)].lastPulled[k]; }
  }
  M->prevBands = M->bands;
  M->haveHistory = true;
  rebalance(M);
  return rc;
}

int rt_mgpu_sync(rt_mgpu* M)
{
  if(!M) return RT_ERR_INVALID_ARG;
  DeviceGuard guard;
  beginCall(M);
  int rc = drainAll(M);
  if(rc == RT_OK) rc = dispatch(M, Cmd::DRAIN);
  return rc;
}

int rt_mgpu_set_balance(rt_mgpu* M, int mode)
{
  if(!M || mode < 0 || mode > 2) return RT_ERR_INVALID_ARG;
  if(mode == 1 && !M->balance) resetBalancer(M);   // (re-enabled after a freeze: the times of the frozen partition are kept only if the balancer never stopped)
  M->balance = mode == 1;
  if(mode == 0 && M->H) { equalBands(M); /* the next frame pulls what moved through the history exchange */ }
  return RT_OK;   // mode 2: keep the partition as it is now
}
// explicit partition (n + 1 boundaries, multiples of 16, first 0, last the image height); implies "freeze" (rt_mgpu_set_balance(m, 2)).  The rows that change
// owner travel with the next frame's history pulls.
int rt_mgpu_set_bands(rt_mgpu* M, const int* bands)
{
  if(!M || !bands || M->H == 0) return RT_ERR_INVALID_ARG;
  if(bands[0] != 0 || bands[M->n] != M->H) return RT_ERR_INVALID_ARG;
  for(int r = 0; r < M->n; r++) if(bands[r + 1] <= bands[r] || (bands[r] & 15)) return RT_ERR_INVALID_ARG;
  M->bands.assign(bands, bands + M->n + 1);
  M->balance = false;
  resetBalancer(M);
  return RT_OK;
}
int rt_mgpu_set_serialize(rt_mgpu* M, int on) { if(!M) return RT_ERR_INVALID_ARG; M->serialize = on != 0; return RT_OK; }
int rt_mgpu_set_pipeline(rt_mgpu* M, int on) { if(!M) return RT_ERR_INVALID_ARG; M->pipeline = on != 0; return RT_OK; }
int rt_mgpu_set_gather(rt_mgpu* M, int on) { if(!M) return RT_ERR_INVALID_ARG; M->gatherResults = on != 0; return RT_OK; }
int rt_mgpu_set_solo(rt_mgpu* M, int rank) { if(!M || rank >= M->n) return RT_ERR_INVALID_ARG; M->solo = rank < 0 ? -1 : rank; return RT_OK; }
int rt_mgpu_get_stats(rt_mgpu* M, rt_mgpu_stats* out)
{
  if(!M || !out) return RT_ERR_INVALID_ARG;
  {   // the numbers of the latest complete frame (finishes what is in flight)
    DeviceGuard guard;
    const bool wasActive = M->pipeActive;
    (void)drainAll(M);
    rt_mgpu_stats& S = M->stats;
    S.haloBytes = 0; for(auto& k : S.haloBytesKind) k = 0;
    S.historyFallbacks = M->fallbacks.load();
    for(int r = 0; r < M->n; r++) {
      Rank& R = M->ranks[size_t(r)];
      if(wasActive) harvestTiming(R, M->seq - 1);
      S.tracedMs[r] = R.tracedMs; S.filterMs[r] = R.filterMs;
      for(int k = 0; k < HK_COUNT; k++) { S.haloBytes += R.lastPulled[k]; S.haloBytesKind[k] += R.lastPulled[k]; S.haloBytesRankKind[r][k] = R.lastPulled[k]; }
    }
  }
  *out = M->stats;
  return RT_OK;
}
const char* rt_mgpu_last_error(rt_mgpu* M) { return (M && !M->err.empty()) ? M->err.c_str() : ""; }

// What the links did for the latest complete frame of the frames-in-flight schedule: the four pull groups of every rank, event-timed on the stream that
// carried them, with their bytes; plus the device of every rank and whether the puller has direct peer access to the owner (xGMI on an MI355X node).
int rt_mgpu_get_stream_layout(rt_mgpu* M, int* created, int* index, int levels[3])
{
  if(!M) return RT_ERR_INVALID_ARG;
  for(int r = 0; r < M->n; r++) {
    int idx[3] = {-1, -1, -1};
    const int rc = rt_get_stream_layout(M->ranks[size_t(r)].ctx, created, idx);
    if(rc != RT_OK) return rc;
    if(index) for(int k = 0; k < 3; k++) index[3 * r + k] = idx[k];
  }
  if(levels) { levels[0] = M->levels[0]; (void)rt_get_stream_priorities(M->ranks[0].ctx, &levels[1], &levels[2], nullptr, nullptr); }
  return RT_OK;
}

int rt_mgpu_get_link_stats(rt_mgpu* M, rt_mgpu_link_stats* out)
{
  if(!M || !out) return RT_ERR_INVALID_ARG;
  DeviceGuard guard;
  const bool wasActive = M->pipeActive;
  (void)drainAll(M);
  memset(out, 0, sizeof(*out));
  out->numRanks = M->n;
  for(int r = 0; r < M->n; r++) {
    Rank& R = M->ranks[size_t(r)];
    if(wasActive) harvestTiming(R, M->seq - 1);
    out->devices[r] = R.dev;
    for(int q = 0; q < M->n; q++) out->peerAccess[r][q] = M->peerOk[size_t(r) * size_t(M->n) + size_t(q)];
    for(int g = 0; g < Rank::LG_COUNT; g++) { out->pullMs[r][g] = R.linkMs[g]; out->pullBytes[r][g] = R.linkBytes[g]; }
  }
  return RT_OK;
}

// Assemble a buffer of the LAST rendered frame from the ranks that own its rows (caller-side layout == rt_readback's).
int rt_mgpu_readback(rt_mgpu* M, int buffer, void* dst, size_t bytes)
{
  if(!M || !dst || buffer < 0 || buffer >= RT_BUF_COUNT) return RT_ERR_INVALID_ARG;
  if(M->W == 0) return RT_ERR_NO_TARGET;
  DeviceGuard guard;
  beginCall(M);
  { const int rc = drainAll(M); if(rc != RT_OK) return rc; }
  const size_t want = rt_buffer_bytes(M->ranks[0].ctx, buffer);
  if(bytes != want) { M->err = "rt_mgpu_readback: size mismatch"; return RT_ERR_INVALID_ARG; }
  const bool indTemp = buffer == RT_BUF_DENOISE_IND_A || buffer == RT_BUF_DENOISE_IND_B;
  const bool half = halfRows(buffer);
  const int rowsTotal = (half && !indTemp) ? M->H / 2 : M->H;   // the half-res temporaries are full-size allocations
  const size_t pitch = want / size_t(rowsTotal);
  const std::vector<int>& part = M->prevBands;                  // the partition the last frame was rendered with
  for(int r = 0; r < M->n; r++) {
    Rank& R = M->ranks[size_t(r)];
    int a = half ? part[size_t(r)] / 2 : part[size_t(r)], b = half ? part[size_t(r) + 1] / 2 : part[size_t(r) + 1];
    if(r == M->n - 1) b = std::max(b, half ? M->H / 2 : M->H);
    if(indTemp && r == M->n - 1) b = M->H;                       // rows below H/2 of the temporaries: unused, taken from the last rank
    b = std::min(b, rowsTotal);
    if(b <= a) continue;
    void* src = nullptr; size_t sb = 0, sp = 0;
    if(buffer == RT_BUF_DENOISE_IND_A) (void)rt_select_frame(R.ctx, M->lastFrames);   // one buffer per frame parity: the last frame's
    if(rt_device_ptr(R.ctx, buffer, &src, &sb, &sp) != RT_OK) return RT_ERR_INVALID_ARG;
    (void)hipSetDevice(R.dev);
    if(hipMemcpy(static_cast<char*>(dst) + size_t(a) * pitch, static_cast<char*>(src) + size_t(a) * pitch, size_t(b - a) * pitch, hipMemcpyDeviceToHost) != hipSuccess)
      return RT_ERR_HIP;
  }
  return RT_OK;
}

}  // extern "C"
