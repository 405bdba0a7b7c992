// renderer.hpp — C++ host classes with the reference's stage-dispatcher / acceleration-structure interfaces
// (src/renderer.hpp:49-61, src/accelstruct.hpp:40-46) whose bodies are calls into the C-ABI of include/rt_abi.h.
// Header-only: link the application with librestir_hip.so (+ librestir_host.so for Scene / HdrSampling).
#pragma once
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/rt_abi.h"
#include "scene.hpp"

namespace rth {

// src/accelstruct.hpp:40-46: setup / create / destroy.  create() = BLAS per prim mesh + TLAS per node in the reference
// (accelstruct.cpp:55-162); here the scene arrays go to HBM and a flat BVH8 is built over them.
class AccelStructure {
 public:
  void setup(rt_ctx* ctx) { m_ctx = ctx; }
  bool create(const Scene& scene, const HdrSampling* env)
  {
    rt_scene_desc d = scene.getDesc(env);
    if(rt_upload_scene(m_ctx, &d) != RT_OK || rt_build_accel(m_ctx) != RT_OK) { fprintf(stderr, "AccelStructure::create: %s\n", rt_last_error(m_ctx)); return false; }
    return true;
  }
  void destroy() {}  // owned by the context
 private:
  rt_ctx* m_ctx = nullptr;
};

// src/renderer.hpp:49-61
class Renderer {
 public:
  bool setup(int device = 0)  // renderer.cpp:62-73
  {
    if(rt_create(&m_ctx, device) != RT_OK) { fprintf(stderr, "Renderer::setup: %s\n", rt_last_error(nullptr)); return false; }
    return true;
  }
  void destroy() { if(m_ctx) rt_destroy(m_ctx); m_ctx = nullptr; }  // renderer.cpp:75-91
  // renderer.cpp:97-148: screen-space buffers for `size` (the reference also creates 7 pipelines + descriptor sets here)
  bool create(int width, int height, Scene* scene) { (void)scene; return update(width, height); }
  // renderer.cpp:154-206: the 12 dispatches of one frame
  bool run(const rt_state& state, int frames)
  {
    if(rt_render_frame(m_ctx, &state, frames) != RT_OK) { fprintf(stderr, "Renderer::run: %s\n", rt_last_error(m_ctx)); return false; }
    return true;
  }
  // (no counterpart in the reference, which records into one queue) priorities of the schedule's indirect / filter streams; unset, the context decides at its first frame
  bool setStreamPriorities(int indirectLevel, int filterLevel)
  {
    if(rt_set_stream_priorities(m_ctx, indirectLevel, filterLevel) != RT_OK) { fprintf(stderr, "Renderer::setStreamPriorities: %s\n", rt_last_error(m_ctx)); return false; }
    return true;
  }
  const std::string name() { return std::string("HIP-gfx950"); }  // renderer.hpp:55 returns "RQ"
  bool update(int width, int height)  // renderer.cpp:209-225
  {
    m_width = width; m_height = height;
    if(rt_resize(m_ctx, width, height) != RT_OK) { fprintf(stderr, "Renderer::update: %s\n", rt_last_error(m_ctx)); return false; }
    return true;
  }
  bool setCamera(const rt_scene_camera& cam) { return rt_set_camera(m_ctx, &cam) == RT_OK; }  // Scene::updateCamera's UBO upload
  // the two HDR images post.frag sums (post.frag:129); `frames` selects the ping-pong side like RenderOutput::getDescSet
  bool readResult(int frames, std::vector<float>& direct, std::vector<float>& indirect)
  {
    const size_t n = size_t(m_width) * m_height * 4;
    direct.resize(n); indirect.resize(n);
    return rt_readback(m_ctx, RT_BUF_DIRECT_RESULT0 + (frames & 1), direct.data(), n * sizeof(float)) == RT_OK
           && rt_readback(m_ctx, RT_BUF_INDIRECT_RESULT0 + (frames & 1), indirect.data(), n * sizeof(float)) == RT_OK;
  }
  rt_ctx* context() { return m_ctx; }
 private:
  rt_ctx* m_ctx = nullptr;
  int m_width = 0, m_height = 0;
};

// src/render_output.hpp:37-73: the display pass.  run() is the fullscreen post.frag draw (render_output.cpp:224-237) as a
// compute pass into an RGBA8 image; the two HDR result images themselves live in the context (RT_BUF_*_RESULT*).
class RenderOutput {
 public:
  rt_tonemapper m_tm{1.0f, 1.0f, 1.0f, 0.0f, 1.0f, 1.0f, {1.0f, 1.0f}, 0, 0.5f, 0.5f, 0};      // render_output.hpp:44-55
  rt_tonemapper m_depthTm{0.0f, 2.2f, 0.0f, 0.0f, 0.0f, 0.0f, {0.0f, 0.0f}, 0, 0.0f, 0.0f, 0};  // render_output.hpp:56-60
  void setup(rt_ctx* ctx) { m_ctx = ctx; }
  void create(int width, int height) { update(width, height); }
  void update(int width, int height) { m_width = width; m_height = height; }
  void destroy() {}
  bool run(const rt_state& state, float zoom, rt_vec2 ratio, int frames)
  {
    rt_tonemapper tm = (state.debugging_mode == RT_DBG_DEPTH) ? m_depthTm : m_tm;  // render_output.cpp:228-230
    tm.zoom = zoom; tm.renderingRatio = ratio;
    if(rt_tonemap(m_ctx, &tm, state.debugging_mode, frames) != RT_OK) { fprintf(stderr, "RenderOutput::run: %s\n", rt_last_error(m_ctx)); return false; }
    return true;
  }
  // the image the reference presents: RGBA8, row-major, top row first
  bool readImage(std::vector<uint8_t>& rgba)
  {
    rgba.resize(size_t(m_width) * m_height * 4);
    return rt_readback(m_ctx, RT_BUF_LDR, rgba.data(), rgba.size()) == RT_OK;
  }
 private:
  rt_ctx* m_ctx = nullptr;
  int m_width = 0, m_height = 0;
};

}  // namespace rth
