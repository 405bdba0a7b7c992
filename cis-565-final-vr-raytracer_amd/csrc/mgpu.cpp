// mgpu.cpp — rt_mgpu_*: the row-tiled multi-GPU frame as ONE native context that drives N devices from one
// process: one host thread, one rt_ctx and one HIP stream per device, halos moved with hipMemcpyPeerAsync over xGMI.
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
// Frame schedule per rank (host barriers between the numbered steps; every copy is a PULL by the rank that needs the rows):
//   1. history: last frame's G-buffer / direct reservoirs / light ids / indirect reservoirs for the band +- 32 rows, from the ranks
//      that owned those rows LAST frame (this also moves state when a boundary moved)
//   2. direct stage on the band, indirect stage on the band's half-res rows; a temporal lookup outside band + halo raises a flag;
//      if any rank's flag is up, every rank pulls the full history and runs the two stages again (exact for any camera motion)
//   3. one exchange for all nine A-Trous passes: 144 G-buffer rows, 40 rows of noisy direct colour, 72 half-res rows of noisy
//      indirect colour from the owners; then the filters on regions that start wider than the band and shrink per level
//      (direct +32/+24/+16/+0, indirect +64/+56/+48/+32/+0 rows: the overlap is recomputed instead of exchanged 9 times); compose
//   4. rank 0 pulls the two result images' bands (the display rank of SURVEY §8e(3))
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/rt_abi.h"

namespace {

constexpr int HIST_HALO = 32;                       // full-res rows of last-frame history around the band
constexpr int DIRECT_GROW[4] = {32, 24, 16, 0};     // rows added to the band for A-Trous level l's output (multiples of 8)
constexpr int INDIRECT_GROW[5] = {64, 56, 48, 32, 0};
constexpr int HALO_DIRECT_COLOR = 40, HALO_INDIRECT_COLOR = 72, HALO_GBUFFER = 144;
constexpr int MAX_RANKS = RT_MGPU_MAX_RANKS;

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

struct Rank {
  int id = 0, dev = 0;
  rt_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {};   // start, traced, filters-begin, end
  std::thread th;
  int rc = RT_OK;
  float tracedMs = 0, filterMs = 0;
  uint64_t pulled = 0;
};

}  // namespace

extern "C" int rt_mgpu_plan_bands(int height, int numRanks, const float* stripeCost, const int* prevBands, int maxMoveStripes, int* outBands);

struct rt_mgpu {
  int n = 0;
  std::vector<Rank> ranks;
  int W = 0, H = 0;
  std::vector<int> bands, prevBands;   // n + 1 row boundaries (multiples of 16, last = H)
  std::vector<float> stripeCost;       // smoothed cost per 16-row stripe
  bool haveHistory = false, balance = true, serialize = false, gatherResults = true;
  // per-frame command
  rt_state st{}; int frames = 0; rt_scene_camera cam{}; const rt_scene_desc* desc = nullptr;
  enum Cmd { NONE, UPLOAD, RESIZE, FRAME, SYNC, QUIT } cmd = NONE;
  Barrier start, done, step;
  int missFlags[MAX_RANKS] = {};
  std::mutex turn;                     // serialize mode: one rank's kernels at a time
  std::string err;
  rt_mgpu_stats stats{};
  std::mutex errLock;
  void fail(int rank, int rc, const char* what, const char* detail)
  {
    std::lock_guard<std::mutex> l(errLock);
    if(err.empty()) err = std::string("rt_mgpu rank ") + std::to_string(rank) + ": " + what + " failed (" + std::to_string(rc) + "): " + (detail ? detail : "");
  }
};

namespace {

bool halfRows(int buf)
{
  return buf == RT_BUF_INDIRECT_RESV0 || buf == RT_BUF_INDIRECT_RESV1 || buf == RT_BUF_INDIRECT_RESV_TEMP || buf == RT_BUF_DENOISE_IND_A || buf == RT_BUF_DENOISE_IND_B;
}
// row pitch of `buf` in its own ctx; the half-res filter temporaries live in the top-left quarter of full-pitch images
size_t pitchOf(rt_ctx* c, int buf)
{
  void* p; size_t bytes, pitch;
  if(rt_device_ptr(c, buf, &p, &bytes, &pitch) != RT_OK) return 0;
  return pitch;
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

// copy rows [a, b) of `buf` into rank R's copy from the ranks that own them under `part` (full-res boundaries)
void pullRows(rt_mgpu& M, Rank& R, int buf, int a, int b, const std::vector<int>& part)
{
  const bool half = halfRows(buf);
  const int limit = half ? M.H / 2 : M.H;
  a = std::max(0, a); b = std::min(limit, b);
  if(b <= a) return;
  void* dst = nullptr; size_t bytes = 0, pitch = 0;
  MG_CHECK(rt_device_ptr(R.ctx, buf, &dst, &bytes, &pitch), "rt_device_ptr");
  for(int q = 0; q < M.n; q++) {
    if(q == R.id) continue;
    int lo = half ? part[q] / 2 : part[q], hi = half ? part[q + 1] / 2 : part[q + 1];
    hi = std::min(hi, limit);
    lo = std::max(lo, a); hi = std::min(hi, b);
    if(hi <= lo) continue;
    void* src = nullptr; size_t sb = 0, sp = 0;
    Rank& Q = M.ranks[q];
    if(rt_device_ptr(Q.ctx, buf, &src, &sb, &sp) != RT_OK || sp != pitch) { M.fail(R.id, RT_ERR_INVALID_ARG, "peer rt_device_ptr", ""); R.rc = RT_ERR_INVALID_ARG; return; }
    const size_t off = size_t(lo) * pitch, len = size_t(hi - lo) * pitch;
    if(Q.dev == R.dev) MG_HIP(hipMemcpyAsync(static_cast<char*>(dst) + off, static_cast<char*>(src) + off, len, hipMemcpyDeviceToDevice, R.stream), "hipMemcpyAsync");
    else MG_HIP(hipMemcpyPeerAsync(static_cast<char*>(dst) + off, R.dev, static_cast<char*>(src) + off, Q.dev, len, R.stream), "hipMemcpyPeerAsync");
    R.pulled += len;
  }
}

void runStage(rt_mgpu& M, Rank& R, int stage, int level, int r0, int r1, int limit)
{
  r0 = std::max(0, r0); r1 = std::min(limit, r1);
  if(r1 > r0) MG_CHECK(rt_run_stage(R.ctx, &M.st, M.frames, stage, level, r0, r1), "rt_run_stage");
}

// direct stage on the band, indirect stage on its half-res rows.  With spatial reuse the direct stage runs in two halves around an
// exchange: every pixel caches its reservoir (level 1), the ranks pull the rows next to their band from RT_BUF_DIRECT_RESV_TEMP — the
// neighbour picks of direct_stage.comp:86-107 reach one pixel up / down — and then merge and shade (level 2).  Contains one barrier in
// the spatial modes (uniform over ranks: the mode comes from the shared RtxState).
void tracedStages(rt_mgpu& M, Rank& R, int y0, int y1, int h0, int h1)
{
  const bool spatial = M.n > 1 && (M.st.ReSTIRState == RT_RESTIR_SPATIAL || M.st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL);
  if(!spatial) runStage(M, R, RT_STAGE_DIRECT, 0, y0, y1, M.H);
  else {
    runStage(M, R, RT_STAGE_DIRECT, 1, y0, y1, M.H);
    MG_HIP(hipStreamSynchronize(R.stream), "sync");
    M.step.wait();
    pullRows(M, R, RT_BUF_DIRECT_RESV_TEMP, y0 - 2, y0, M.bands);
    pullRows(M, R, RT_BUF_DIRECT_RESV_TEMP, y1, y1 + 2, M.bands);
    runStage(M, R, RT_STAGE_DIRECT, 2, y0, y1, M.H);
  }
  runStage(M, R, RT_STAGE_INDIRECT, 0, h0, h1, M.H / 2);
}

void frameOnRank(rt_mgpu& M, Rank& R)
{
  const int f = M.frames, cur = f & 1, last = cur ^ 1, H = M.H, Hh = H / 2;
  const int y0 = M.bands[R.id], y1 = M.bands[R.id + 1];
  const int h0 = std::min(y0 / 2, Hh), h1 = std::min(y1 / 2, Hh);
  const bool multi = M.n > 1;
  R.pulled = 0;
  MG_CHECK(rt_set_camera(R.ctx, &M.cam), "rt_set_camera");
  // ---- 1. history rows for the band + halo, from last frame's owners ----
  if(multi && M.haveHistory) {
    for(int buf : {RT_BUF_GBUFFER0 + last, RT_BUF_DIRECT_RESV0 + last, RT_BUF_LIGHT_ID0 + last}) pullRows(M, R, buf, y0 - HIST_HALO, y1 + HIST_HALO, M.prevBands);
    pullRows(M, R, RT_BUF_INDIRECT_RESV0 + last, h0 - HIST_HALO / 2, h1 + HIST_HALO / 2, M.prevBands);
    // rows that changed owner also bring the OTHER parity of the reservoir buffers along: pixels that return early (miss, emitter,
    // debug view) leave their slot untouched (reference quirk), so a slot of this frame's buffer can keep the value of
    // two frames ago.  Nothing is copied while the partition stands still (pullRows only copies rows other ranks owned).
    for(int buf : {RT_BUF_DIRECT_RESV0 + cur, RT_BUF_LIGHT_ID0 + cur, int(RT_BUF_DIRECT_RESV_TEMP)}) pullRows(M, R, buf, y0, y1, M.prevBands);
    pullRows(M, R, RT_BUF_INDIRECT_RESV0 + cur, h0, h1, M.prevBands);
  }
  MG_CHECK(rt_set_history_rows(R.ctx, multi ? std::max(0, y0 - HIST_HALO) : 0, multi ? std::min(H, y1 + HIST_HALO) : H), "rt_set_history_rows");
  // ---- 2. ray-traced stages ----
  {
    const bool spatialSplit = multi && (M.st.ReSTIRState == RT_RESTIR_SPATIAL || M.st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL);
    std::unique_lock<std::mutex> turn(M.turn, std::defer_lock);
    if(M.serialize && !spatialSplit) { turn.lock(); MG_HIP(hipStreamSynchronize(R.stream), "sync"); }   // (the split stage has a barrier inside: ranks cannot take turns)
    MG_HIP(hipEventRecord(R.ev[0], R.stream), "hipEventRecord");
    tracedStages(M, R, y0, y1, h0, h1);
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
    for(int buf : {RT_BUF_GBUFFER0 + last, RT_BUF_DIRECT_RESV0 + last, RT_BUF_LIGHT_ID0 + last}) pullRows(M, R, buf, 0, H, M.prevBands);
    pullRows(M, R, RT_BUF_INDIRECT_RESV0 + last, 0, Hh, M.prevBands);
    MG_CHECK(rt_set_history_rows(R.ctx, 0, H), "rt_set_history_rows");
    const bool spatialSplit = M.st.ReSTIRState == RT_RESTIR_SPATIAL || M.st.ReSTIRState == RT_RESTIR_SPATIOTEMPORAL;
    std::unique_lock<std::mutex> turn(M.turn, std::defer_lock);
    if(M.serialize && !spatialSplit) turn.lock();
    tracedStages(M, R, y0, y1, h0, h1);
    int miss = 0;
    MG_CHECK(rt_history_miss(R.ctx, &miss), "rt_history_miss");   // clears the flag, waits for the stream
    if(R.id == 0) M.stats.historyFallbacks++;
  }
  M.step.wait();   // every rank's G-buffer / reservoirs / noisy colours of this frame are complete
  // ---- 3. one exchange for the nine filter passes, filters, compose ----
  if(multi && M.st.denoise > 0) {
    pullRows(M, R, RT_BUF_GBUFFER0 + cur, y0 - HALO_GBUFFER, y1 + HALO_GBUFFER, M.bands);
    pullRows(M, R, RT_BUF_DIRECT_RESULT0 + cur, y0 - HALO_DIRECT_COLOR, y1 + HALO_DIRECT_COLOR, M.bands);
    pullRows(M, R, RT_BUF_DENOISE_IND_A, h0 - HALO_INDIRECT_COLOR, h1 + HALO_INDIRECT_COLOR, M.bands);
    MG_HIP(hipStreamSynchronize(R.stream), "sync");
  }
  M.step.wait();   // nobody overwrites a source row (level 3 / compose write the result image) before every pull has landed
  {
    std::unique_lock<std::mutex> turn(M.turn, std::defer_lock);
    if(M.serialize) turn.lock();
    MG_HIP(hipEventRecord(R.ev[2], R.stream), "hipEventRecord");
    if(M.st.denoise > 0) {
      for(int l = 0; l < 4; l++) { const int g = multi ? DIRECT_GROW[l] : 0; runStage(M, R, RT_STAGE_DENOISE_DIRECT, l, y0 - g, y1 + g, H); }
      for(int l = 0; l < 5; l++) { const int g = multi ? INDIRECT_GROW[l] : 0; runStage(M, R, RT_STAGE_DENOISE_INDIRECT, l, h0 - g, h1 + g, Hh); }
    }
    runStage(M, R, RT_STAGE_COMPOSE, 0, y0, y1, H);
    MG_HIP(hipEventRecord(R.ev[3], R.stream), "hipEventRecord");
    MG_HIP(hipStreamSynchronize(R.stream), "sync");
  }
  M.step.wait();
  // ---- 4. result bands to the display rank ----
  if(multi && R.id == 0 && M.gatherResults) {
    pullRows(M, R, RT_BUF_DIRECT_RESULT0 + cur, 0, H, M.bands);
    pullRows(M, R, RT_BUF_INDIRECT_RESULT0 + cur, 0, H, M.bands);
    MG_HIP(hipStreamSynchronize(R.stream), "sync");
  }
  float ms = 0.f;
  if(hipEventElapsedTime(&ms, R.ev[0], R.ev[1]) == hipSuccess) R.tracedMs = ms;
  if(hipEventElapsedTime(&ms, R.ev[2], R.ev[3]) == hipSuccess) R.filterMs = ms;
}

void worker(rt_mgpu* Mp, int id)
{
  rt_mgpu& M = *Mp;
  Rank& R = M.ranks[id];
  (void)hipSetDevice(R.dev);
  for(;;) {
    M.start.wait();
    const rt_mgpu::Cmd cmd = M.cmd;
    if(cmd == rt_mgpu::QUIT) break;
    R.rc = RT_OK;
    switch(cmd) {
      case rt_mgpu::UPLOAD:
        MG_CHECK(rt_upload_scene(R.ctx, M.desc), "rt_upload_scene");
        if(R.rc == RT_OK) MG_CHECK(rt_build_accel(R.ctx), "rt_build_accel");
        break;
      case rt_mgpu::RESIZE: MG_CHECK(rt_resize(R.ctx, M.W, M.H), "rt_resize"); break;
      case rt_mgpu::FRAME: frameOnRank(M, R); break;
      case rt_mgpu::SYNC: MG_CHECK(rt_sync(R.ctx), "rt_sync"); break;
      default: break;
    }
    M.done.wait();
  }
}

int dispatch(rt_mgpu* M, rt_mgpu::Cmd cmd)
{
  M->cmd = cmd;
  M->start.wait();
  M->done.wait();
  for(const Rank& R : M->ranks) if(R.rc != RT_OK) return R.rc;
  return RT_OK;
}

void equalBands(rt_mgpu* M)
{
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
  std::vector<int> nb(size_t(n) + 1, 0);
  rt_mgpu_plan_bands(M->H, n, M->stripeCost.data(), M->bands.data(), 2, nb.data());
  M->bands = nb;
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
  rt_mgpu* M = new(std::nothrow) rt_mgpu();
  if(!M) return RT_ERR_OOM;
  M->n = numRanks;
  M->ranks.resize(size_t(numRanks));
  for(int r = 0; r < numRanks; r++) {
    Rank& R = M->ranks[size_t(r)];
    R.id = r; R.dev = devices ? devices[r] : r;
    int rc = rt_create(&R.ctx, R.dev);
    if(rc != RT_OK) { for(int q = 0; q < r; q++) rt_destroy(M->ranks[size_t(q)].ctx); delete M; return rc; }
    (void)hipSetDevice(R.dev);
    bool ok = hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking) == hipSuccess;
    for(auto& e : R.ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    if(!ok) { for(int q = 0; q <= r; q++) rt_destroy(M->ranks[size_t(q)].ctx); delete M; return RT_ERR_HIP; }
    rt_set_stream(R.ctx, R.stream);
    rt_set_overlap(R.ctx, 0);   // stages are issued one by one through rt_run_stage
  }
  // direct xGMI copies between distinct devices (already-enabled is not an error)
  for(int a = 0; a < numRanks; a++)
    for(int b = 0; b < numRanks; b++) {
      const int da = M->ranks[size_t(a)].dev, db = M->ranks[size_t(b)].dev;
      if(da == db) continue;
      int can = 0;
      if(hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) { (void)hipSetDevice(da); (void)hipDeviceEnablePeerAccess(db, 0); (void)hipGetLastError(); }
    }
  M->start.init(numRanks + 1); M->done.init(numRanks + 1); M->step.init(numRanks);
  for(int r = 0; r < numRanks; r++) M->ranks[size_t(r)].th = std::thread(worker, M, r);
  *out = M;
  return RT_OK;
}

int rt_mgpu_destroy(rt_mgpu* M)
{
  if(!M) return RT_ERR_INVALID_ARG;
  M->cmd = rt_mgpu::QUIT;
  M->start.wait();
  for(Rank& R : M->ranks) {
    if(R.th.joinable()) R.th.join();
    (void)hipSetDevice(R.dev);
    rt_destroy(R.ctx);
    for(auto& e : R.ev) if(e) (void)hipEventDestroy(e);
    if(R.stream) (void)hipStreamDestroy(R.stream);
  }
  delete M;
  return RT_OK;
}

int rt_mgpu_upload_scene(rt_mgpu* M, const rt_scene_desc* d)
{
  if(!M || !d) return RT_ERR_INVALID_ARG;
  M->desc = d;
  const int rc = dispatch(M, rt_mgpu::UPLOAD);
  M->desc = nullptr; M->haveHistory = false;
  return rc;
}

int rt_mgpu_resize(rt_mgpu* M, int w, int h)
{
  if(!M || w <= 0 || h <= 0) return RT_ERR_INVALID_ARG;
  if((h + 15) / 16 < M->n) { M->err = "rt_mgpu_resize: fewer 16-row stripes than ranks"; return RT_ERR_INVALID_ARG; }
  M->W = w; M->H = h;
  const int rc = dispatch(M, rt_mgpu::RESIZE);
  equalBands(M);
  M->prevBands = M->bands;
  M->stripeCost.clear();
  M->haveHistory = false;
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
  M->st = *st; M->frames = frames;
  const int rc = dispatch(M, rt_mgpu::FRAME);
  // statistics of this frame, then the partition of the next one
  rt_mgpu_stats& S = M->stats;
  S.numRanks = M->n; S.frames++;
  S.haloBytes = 0;
  for(int r = 0; r < M->n; r++) {
    S.bandBegin[r] = M->bands[size_t(r)]; S.bandEnd[r] = M->bands[size_t(r) + 1];
    S.tracedMs[r] = M->ranks[size_t(r)].tracedMs; S.filterMs[r] = M->ranks[size_t(r)].filterMs;
    S.haloBytes += M->ranks[size_t(r)].pulled;
  }
  M->prevBands = M->bands;
  M->haveHistory = true;
  rebalance(M);
  return rc;
}

int rt_mgpu_sync(rt_mgpu* M) { return M ? dispatch(M, rt_mgpu::SYNC) : RT_ERR_INVALID_ARG; }

int rt_mgpu_set_balance(rt_mgpu* M, int mode)
{
  if(!M) return RT_ERR_INVALID_ARG;
  M->balance = mode != 0;
  if(!M->balance && M->H) { equalBands(M); /* the next frame pulls what moved through the history exchange */ }
  return RT_OK;
}
int rt_mgpu_set_serialize(rt_mgpu* M, int on) { if(!M) return RT_ERR_INVALID_ARG; M->serialize = on != 0; return RT_OK; }
int rt_mgpu_get_stats(rt_mgpu* M, rt_mgpu_stats* out) { if(!M || !out) return RT_ERR_INVALID_ARG; *out = M->stats; return RT_OK; }
const char* rt_mgpu_last_error(rt_mgpu* M) { return (M && !M->err.empty()) ? M->err.c_str() : ""; }

// Assemble a buffer of the LAST rendered frame from the ranks that own its rows (caller-side layout == rt_readback's).
int rt_mgpu_readback(rt_mgpu* M, int buffer, void* dst, size_t bytes)
{
  if(!M || !dst || buffer < 0 || buffer >= RT_BUF_COUNT) return RT_ERR_INVALID_ARG;
  if(M->W == 0) return RT_ERR_NO_TARGET;
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
    if(rt_device_ptr(R.ctx, buffer, &src, &sb, &sp) != RT_OK) return RT_ERR_INVALID_ARG;
    (void)hipSetDevice(R.dev);
    if(hipMemcpy(static_cast<char*>(dst) + size_t(a) * pitch, static_cast<char*>(src) + size_t(a) * pitch, size_t(b - a) * pitch, hipMemcpyDeviceToHost) != hipSuccess)
      return RT_ERR_HIP;
  }
  return RT_OK;
}

}  // extern "C"
