#pragma once
#include <hip/hip_runtime.h>
#include "dev_scene.h"
namespace rt {
constexpr int STACK_MAX = 64;  // LDS traversal stack entries per lane (8 B each) upper bound; rt_build_accel rejects deeper trees
// one entry of Renderer::run's dispatch list (renderer.cpp:163-205) on `stream`.  stages.hip is compiled six times:
//   base / sky          HDR environment only / sun & sky code paths compiled in
//   base_cnt / sky_cnt  the same with the counters of rt_set_counting flushed
//   base_lat / sky_lat  the traced kernels for small launches (latency-mode traversal); every other stage forwards to base / sky
#define RT_DECL_LAUNCH(ns)                                                                                                                              \
  namespace ns {                                                                                                                                        \
  hipError_t launchStage(hipStream_t stream, const DevScene& S, const DevFrame& F, const rt_state& st, const rt_scene_camera& cam, int stage, int level, \
                         int rowBegin, int rowEnd);                                                                                                     \
  }
RT_DECL_LAUNCH(base)
RT_DECL_LAUNCH(sky)
RT_DECL_LAUNCH(base_cnt)
RT_DECL_LAUNCH(sky_cnt)
RT_DECL_LAUNCH(base_lat)
RT_DECL_LAUNCH(sky_lat)
#undef RT_DECL_LAUNCH

// uniform-only terms of sun_and_sky() (sky.h), one thread
struct SkyPre;
hipError_t launchSkyPrepare(hipStream_t stream, const rt_sun_and_sky& ss, SkyPre* out);
hipError_t launchTraceRays(hipStream_t stream, const DevScene& S, int n, const float4* rays, float4* out, int anyHit);
hipError_t launchPick(hipStream_t stream, const DevScene& S, const rt_mat4& viewInv, const rt_mat4& projInv, float pickX, float pickY, rt_pick_result* out);
// RenderOutput::run + post.frag as compute (post.hip)
// csrc/microbench.hip
hipError_t measureValuIssue(hipStream_t stream, int variant, int wavesPerSimd, double* waveInstPerSec, double* seconds);
hipError_t launchTonemap(hipStream_t stream, const float4* direct, const float4* indirect, double* rowSums, float* mean, const rt_tonemapper& tm, int dbg, int W, int H,
                         uint32_t* ldr, float4* mipD, float4* mipI);
}
