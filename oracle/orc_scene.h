// orc_scene.h — CPU oracle scene storage + naive binary BVH (TEST INFRASTRUCTURE, not product code).
//
// Restates, on the host and with a software BVH, what the reference delegates to Vulkan:
//   * ray/triangle + acceleration-structure traversal: VK_KHR_ray_query in the driver / RT cores
//     (call sites shaders/traceray_rq.glsl:114-145, 160-184; build src/accelstruct.cpp:110-162).
//     Not in the reference tree, no pinned version => PARITY UNPINNED for this part.  The algorithm
//     restated here is the published Möller–Trumbore test ("Fast, Minimum Storage Ray/Triangle
//     Intersection", JGT 1997) with the Vulkan ray-query semantics the reference asks for:
//     tmin < t < tmax, gl_RayFlagsCullBackFacingTrianglesEXT (facing decided in object space, CCW =
//     front), per-instance FORCE_OPAQUE / TRIANGLE_FACING_CULL_DISABLE (accelstruct.cpp:145-149),
//     closest committed hit; ties in t are resolved towards the lowest flattened triangle index so the
//     answer does not depend on traversal order.
//   * texture sampling: Vulkan samplers (src/scene.cpp:513-548, 628-640; hdr_sampling.cpp:69-77).
#pragma once
#include <vector>
#include <algorithm>
#include <atomic>
#include "../include/rt_abi.h"
#include "orc_math.h"

namespace orc {

struct Tri {
  vec3 v0, v1, v2;  // world space
  uint32_t inst;    // TLAS instance (node) index
  uint32_t prim;    // primitive index inside the prim mesh
  uint32_t flags;   // bit0 opaque, bit1 cull disabled, bit2 winding flipped (mirroring instance transform)
};
enum { TRI_OPAQUE = 1, TRI_NOCULL = 2, TRI_FLIP = 4 };

struct BvhNode {
  float lo[3], hi[3];
  uint32_t left;   // internal: index of left child (right = left+1); leaf: first triangle
  uint32_t count;  // 0 = internal, else number of triangles
};

struct Hit {
  float t = RT_INFINITY;
  uint32_t tri = 0xffffffffu;
  float u = 0, v = 0;
};

// Ray / step counters.  Every thread counts into its own thread-local block (`local()`) and adds it to the shared totals when it has finished a row
// (`flush()`; Frame::parallelRows, the C entry points): until round 5 every ray did four atomic increments on ONE cache line shared by all worker threads,
// which capped the 256-thread CPU baseline of bench.py at 1.6-3.3x of a single thread and made it bimodal (profiles/r05_cpu_baseline.txt).
struct Counters {
  struct Local { uint64_t closestHitRays = 0, anyHitRays = 0, nodesVisited = 0, trisTested = 0, hitsShaded = 0, risCandidates = 0; };
  std::atomic<uint64_t> closestHitRays{0}, anyHitRays{0}, nodesVisited{0}, trisTested{0}, hitsShaded{0}, risCandidates{0};
  static Local& local() { static thread_local Local l; return l; }
  void flush()
  {
    Local& l = local();
    if(l.closestHitRays) closestHitRays += l.closestHitRays;
    if(l.anyHitRays) anyHitRays += l.anyHitRays;
    if(l.nodesVisited) nodesVisited += l.nodesVisited;
    if(l.trisTested) trisTested += l.trisTested;
    if(l.hitsShaded) hitsShaded += l.hitsShaded;
    if(l.risCandidates) risCandidates += l.risCandidates;
    l = Local{};
  }
  // (the calling thread's local block is empty between C entry points — each flushes before it returns — so there is nothing of another context to discard here)
  void reset() { flush(); closestHitRays = anyHitRays = nodesVisited = trisTested = hitsShaded = risCandidates = 0; }
};

struct Texture {
  std::vector<uint8_t> bgra;
  int w = 1, h = 1, wrapS = RT_WRAP_REPEAT, wrapT = RT_WRAP_REPEAT, filter = RT_FILTER_LINEAR;
};

struct Scene {
  std::vector<rt_prim_mesh> primMeshes;
  std::vector<rt_vertex> vertices;
  std::vector<uint32_t> indices;
  std::vector<rt_instance> instances;
  std::vector<affine> objectToWorld, worldToObject;
  std::vector<rt_material> materials;
  std::vector<Texture> textures;
  std::vector<rt_punc_light> puncLights;
  std::vector<rt_trig_light> trigLights;
  rt_light_buf_info lightInfo{};
  rt_sun_and_sky sunAndSky{};  // in_use = 0 by default (uniform block _sunAndSky, layouts.glsl:53)
  int envW = 1, envH = 1;
  std::vector<float> env;  // rgba32f
  std::vector<rt_impt_samp> envAccel;

  std::vector<Tri> tris;
  float triPad = 0.0f;   // box padding of the build; also bounds where an accepted hit may lie (intersectTri)          // flattened, index = (instance order, primitive order)
  std::vector<uint32_t> leafTris_; // BVH leaf order -> index into tris
  std::vector<BvhNode> nodes;
  mutable Counters counters;

  void upload(const rt_scene_desc* d);
  void build();

  // ---- texture fetch (Vulkan sampler restatement, LOD 0) ------------------------------------------
  static int wrapCoord(int i, int n, int mode)
  {
    if(mode == RT_WRAP_CLAMP) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    if(mode == RT_WRAP_MIRROR) {
      int p = 2 * n;
      int m = i % p; if(m < 0) m += p;
      return m < n ? m : p - 1 - m;
    }
    int m = i % n; if(m < 0) m += n;
    return m;
  }
  static vec4 texel(const Texture& t, int x, int y)
  {
    const uint8_t* p = &t.bgra[(size_t(y) * t.w + x) * 4];
    // VK_FORMAT_B8G8R8A8_UNORM (scene.cpp:559): byte0 = B, byte1 = G, byte2 = R, byte3 = A
    return V4(float(p[2]) / 255.0f, float(p[1]) / 255.0f, float(p[0]) / 255.0f, float(p[3]) / 255.0f);
  }
  vec4 sampleTexture(int id, vec2 uv) const
  {
    const Texture& t = textures[id];
    float fx = uv.x * float(t.w), fy = uv.y * float(t.h);
    if(t.filter == RT_FILTER_NEAREST) {
      int x = wrapCoord(rt_ftoi(rt_floor(fx)), t.w, t.wrapS), y = wrapCoord(rt_ftoi(rt_floor(fy)), t.h, t.wrapT);
      return texel(t, x, y);
    }
    fx = fx - 0.5f; fy = fy - 0.5f;
    float x0f = rt_floor(fx), y0f = rt_floor(fy);
    float ax = fx - x0f, ay = fy - y0f;
    int x0 = rt_ftoi(x0f), y0 = rt_ftoi(y0f);
    int xa = wrapCoord(x0, t.w, t.wrapS), xb = wrapCoord(x0 + 1, t.w, t.wrapS);
    int ya = wrapCoord(y0, t.h, t.wrapT), yb = wrapCoord(y0 + 1, t.h, t.wrapT);
    vec4 top = mix(texel(t, xa, ya), texel(t, xb, ya), ax);
    vec4 bot = mix(texel(t, xa, yb), texel(t, xb, yb), ax);
    return mix(top, bot, ay);
  }
  vec4 envTexel(int x, int y) const { const float* p = &env[(size_t(y) * envW + x) * 4]; return V4(p[0], p[1], p[2], p[3]); }
  // environmentTexture: linear, U repeat, V clamp-to-edge (hdr_sampling.cpp:69-77)
  vec4 sampleEnv(vec2 uv) const
  {
    float fx = uv.x * float(envW) - 0.5f, fy = uv.y * float(envH) - 0.5f;
    float x0f = rt_floor(fx), y0f = rt_floor(fy);
    float ax = fx - x0f, ay = fy - y0f;
    int x0 = rt_ftoi(x0f), y0 = rt_ftoi(y0f);
    int xa = wrapCoord(x0, envW, RT_WRAP_REPEAT), xb = wrapCoord(x0 + 1, envW, RT_WRAP_REPEAT);
    int ya = wrapCoord(y0, envH, RT_WRAP_CLAMP), yb = wrapCoord(y0 + 1, envH, RT_WRAP_CLAMP);
    vec4 top = mix(envTexel(xa, ya), envTexel(xb, ya), ax);
    vec4 bot = mix(envTexel(xa, yb), envTexel(xb, yb), ax);
    return mix(top, bot, ay);
  }

  // ---- ray queries --------------------------------------------------------------------------------
  bool intersectTri(const Tri& T, vec3 o, vec3 d, float& t, float& u, float& v) const;
  bool hitTest(const Tri& T, uint32_t triIndex, float u, float v, uint32_t raySeed) const;
  // seed of HitTest's stochastic draw for one (ray, triangle) pair (DESIGN.md §6 deviation 1).  Never equal to the ray's own
  // seed for triangle 0 (the pixel's next rand() would otherwise repeat the alpha draw).
  static uint32_t candidateSeed(uint32_t raySeed, uint32_t triIndex) { return (raySeed ^ 0x9e3779b9u) + (triIndex + 1u) * 2654435761u; }
  // ClosestHit (traceray_rq.glsl:108-147): tmin 0, tmax INFINITY
  Hit closestHit(vec3 o, vec3 d, uint32_t raySeed) const;
  // AnyHit (traceray_rq.glsl:153-185)
  bool anyHit(vec3 o, vec3 d, float tmax, uint32_t raySeed) const;
};

}  // namespace orc
