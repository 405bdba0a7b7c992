#pragma once
#include <hip/hip_runtime.h>
#include "dev_scene.h"
namespace rt {
constexpr int STACK_MAX = 64;  // LDS traversal stack entries per lane (8 B each) upper bound; rt_build_accel rejects deeper trees
// one entry of Renderer::run's dispatch list (renderer.cpp:163-205) on `stream`
hipError_t launchStage(hipStream_t stream, const DevScene& S, const DevFrame& F, const rt_state& st, const rt_scene_camera& cam, int stage, int level,
                       int rowBegin, int rowEnd);
// the same dispatch entry as a sequence of lean trace kernels + shading kernels with ray compaction (wavefront.hip)
hipError_t launchStageWavefront(hipStream_t stream, const DevScene& S, const DevFrame& F, const rt_state& st, const rt_scene_camera& cam, int stage, int level,
                                int rowBegin, int rowEnd);
// RenderOutput::run + post.frag as compute (post.hip)
hipError_t launchTonemap(hipStream_t stream, const float4* direct, const float4* indirect, double* rowSums, float* mean, const rt_tonemapper& tm, int dbg, int W, int H,
                         uint32_t* ldr);
}
