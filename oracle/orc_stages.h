// orc_stages.h — CPU oracle frame state (screen-space buffers of renderer.cpp:227-302) and stage entry points.
// TEST INFRASTRUCTURE, not product code.
#pragma once
#include <thread>
#include <atomic>
#include <algorithm>
#include <atomic>
#include <vector>
#include <sched.h>
#include <pthread.h>
#include "orc_shading.h"

namespace orc {

struct Frame {
  const Scene* scene = nullptr;
  rt_scene_camera cam{};
  int W = 0, H = 0;
  int threads = 1;
  bool pin = false;   // bench.py's cpu_baseline leg: worker t runs on the t-th CPU of the process's affinity mask (orc_set_threads), so that a pass measures cores, not the scheduler
  // row-tiled runs: rows of the last-frame buffers valid on this rank; lookups outside raise histMiss (rt_abi.h rt_set_history_rows)
  int histRow0 = 0, histRow1 = 1 << 30;
  mutable std::atomic<uint32_t> histMiss{0}, histMissInd{0};  // raised by the direct stages / by the indirect stage (rt_history_miss_stage)

  // boundary layouts == reference layouts (rt_abi.h rt_buffer_id)
  std::vector<uint32_t> gbuffer[2];                 // RGBA32UI
  std::vector<int16_t> motion;                      // RG16_SINT
  std::vector<rt_direct_reservoir> directResv[2], directResvTemp;
  std::vector<rt_indirect_reservoir> indirectResv[2], indirectResvTemp;
  std::vector<float> denoiseTemp[4];                // DirA, DirB, IndA, IndB (RGBA32F, full-res allocation)
  std::vector<float> directResult[2], indirectResult[2];
  std::vector<uint32_t> lightId2[2];
  std::vector<uint32_t> ldr;                        // RGBA8 UNORM: rt_tonemap output

  std::vector<uint32_t>& lightId_cur(int cur) { return lightId2[cur]; }
  const std::vector<uint32_t>& lightId_last(int last) const { return lightId2[last]; }

  void resize(int w, int h);
  void renderFrame(const rt_state& st, int frames);
  void runStage(const rt_state& st, int frames, int stage, int level, int rowBegin, int rowEnd);

  void directStage(const rt_state& st, int frames, int rowBegin, int rowEnd, int phase = 0);  // phase: rt_run_stage's `level` (0 both halves, 1 / 2 one)
  void directGen(const rt_state& st, int frames, int rowBegin, int rowEnd);
  void directReuse(const rt_state& st, int frames, int rowBegin, int rowEnd);
  void indirectStage(const rt_state& st, int frames, int rowBegin, int rowEnd);
  void denoiseDirect(const rt_state& st, int frames, int level, int rowBegin, int rowEnd);
  void denoiseIndirect(const rt_state& st, int frames, int level, int rowBegin, int rowEnd);
  void compose(const rt_state& st, int frames, int rowBegin, int rowEnd);
  void tonemap(const rt_tonemapper& tm, int debugging_mode, int frames);  // orc_post.cpp

  // image access
  uvec4 loadG(int which, ivec2 c) const;
  void storeG(int which, ivec2 c, uvec4 v);
  vec4 loadImg(const std::vector<float>& img, ivec2 c) const;
  void storeImg(std::vector<float>& img, ivec2 c, vec4 v);
  void storeMotion(ivec2 c, ivec2 v);
  ivec2 loadMotion(ivec2 c) const;

 private:
  template <class F> void parallelRows(int rows, int rowBegin, int rowEnd, F&& fn) const;
  void loadLastGeometryInfo(int last, ivec2 c, vec3& normal, float& depth, uint32_t& matHash) const;
  bool findTemporalNeighborDirect(const rt_state& st, int last, vec3 norm, float reprojDepth, uint32_t matId, ivec2 lastCoord,
                                  rt_direct_reservoir& resv, uint32_t& lid) const;
  // spatial / spatiotemporal reuse (direct_stage.comp:224-255): a pixel that reaches the reuse step parks what the rest of
  // ReSTIRDirect needs and finishes after every pixel of the frame has cached its reservoir (SpatialPending / finishSpatial)
  struct SpatialPending { bool active = false; State state; vec3 wo; float hitT = 0; uint32_t seed = 0; rt_direct_reservoir resv; };
  std::vector<SpatialPending> spatialPend;   // parked pixels of the first half (kept between the two rt_run_stage calls of a split stage)
  vec3 ReSTIRDirect(Shader& sh, const Ray& r, int cur, int last, SpatialPending* pending);
  vec3 finishSpatial(Shader& sh, const SpatialPending& P, int cur);
  vec3 ReSTIRIndirect(Shader& sh, float dist, float primSamplePdf, vec3 primWo, State primState, rt_gi_sample gi, int cur, int last);
  void loadThisGeometry(int cur, ivec2 coord, vec3& normal, vec3& pos, uint32_t& matHash, ivec2 imageSize) const;
  vec3 waveletFilter(const rt_state& st, int cur, const std::vector<float>& inImage, ivec2 coord, vec3 norm, vec3 pos, uint32_t matHash,
                     float sigLumin, float sigNormal, float sigDepth, int level, bool indirect) const;
};

template <class F>
void Frame::parallelRows(int rows, int rowBegin, int rowEnd, F&& fn) const
{
  if(rowEnd <= 0 || rowEnd > rows) rowEnd = rows;
  if(rowBegin < 0) rowBegin = 0;
  int nt = std::max(1, std::min(threads, rowEnd - rowBegin));
  Counters& cnt = const_cast<Counters&>(scene->counters);
  if(nt == 1) { for(int y = rowBegin; y < rowEnd; y++) fn(y); cnt.flush(); return; }
  std::vector<std::thread> pool;
  std::atomic<int> next{rowBegin};
  std::vector<int> cpus;
  if(pin) {
    cpu_set_t set; CPU_ZERO(&set);
    if(sched_getaffinity(0, sizeof(set), &set) == 0) for(int c = 0; c < CPU_SETSIZE; c++) if(CPU_ISSET(c, &set)) cpus.push_back(c);
  }
  for(int t = 0; t < nt; t++) {
    // (round 6: the workers are SPREAD over the CPUs of the mask — worker t on CPU t * ncpu / nt — instead of packed on the first nt: 32 workers on CPUs 0-31 shared four L3
    //  slices and two memory channels' worth of a 256-CPU host and ran at 0.53 of the single-thread rate each, profiles/r05_cpu_baseline.txt.)
    // A worker pins ITSELF.  Until the end of round 6 the spawning thread pinned worker t through its handle after creating it — and glibc's pthread_setaffinity_np on a
    // thread that has already exited (a worker that found no row left: short stages, more threads than rows) is sched_setaffinity(0, ...): it pinned the CALLER to that one
    // CPU.  From then on `cpus` had one entry and every later worker — inheriting the caller's mask when not pinned — ran on that CPU: 128 / 256 threads measured 1.00 CPUs
    // busy (scripts/r06_oracle_scaling.py, profiles/r06_cpu_baseline.txt).  That, not the allocator, is why the baseline "peaked at 32 threads" in rounds 4-6.
    pool.emplace_back([&, t] {
      if(!cpus.empty()) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpus[(size_t(t) * cpus.size() / size_t(nt)) % cpus.size()], &one); (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one); }
      for(;;) { int y = next.fetch_add(1); if(y >= rowEnd) break; fn(y); }
      cnt.flush();
    });
  }
  for(auto& th : pool) th.join();
}


}  // namespace orc
