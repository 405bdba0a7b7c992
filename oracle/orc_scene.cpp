// orc_scene.cpp — oracle scene upload, triangle flattening, naive median-split BVH2 and ray queries.
// TEST INFRASTRUCTURE (see oracle/README.md): never linked into the product library.
#include "orc_scene.h"
#include <cmath>
#include <numeric>

namespace orc {

void Scene::upload(const rt_scene_desc* d)
{
  primMeshes.assign(d->primMeshes, d->primMeshes + d->numPrimMeshes);
  vertices.assign(d->vertices, d->vertices + d->numVertices);
  indices.assign(d->indices, d->indices + d->numIndices);
  instances.assign(d->instances, d->instances + d->numInstances);
  materials.assign(d->materials, d->materials + d->numMaterials);
  textures.clear();
  for(uint32_t i = 0; i < d->numTextures; i++) {
    const rt_texture& s = d->textures[i];
    Texture t;
    t.w = s.width; t.h = s.height; t.wrapS = s.wrapS; t.wrapT = s.wrapT; t.filter = s.magFilter;
    t.bgra.assign(s.bgra8, s.bgra8 + size_t(s.width) * s.height * 4);
    textures.push_back(std::move(t));
  }
  lightInfo = d->lightInfo;
  puncLights.clear(); trigLights.clear();
  if(d->puncLights) puncLights.assign(d->puncLights, d->puncLights + lightInfo.puncLightSize);
  if(d->trigLights) trigLights.assign(d->trigLights, d->trigLights + lightInfo.trigLightSize);
  if(d->envRgba32f && d->envWidth > 0 && d->envHeight > 0) {
    envW = d->envWidth; envH = d->envHeight;
    env.assign(d->envRgba32f, d->envRgba32f + size_t(envW) * envH * 4);
    if(d->envAccel) envAccel.assign(d->envAccel, d->envAccel + size_t(envW) * envH);
    else envAccel.assign(size_t(envW) * envH, rt_impt_samp{0, 1.0f, 0.0f, 0.0f});
  } else {
    envW = envH = 1;
    env.assign(4, 0.0f);
    envAccel.assign(1, rt_impt_samp{0, 1.0f, 0.0f, 0.0f});
  }
}

// One world-space triangle per (instance, primitive) — replaces BLAS-per-primMesh + TLAS-per-node
// (accelstruct.cpp:110-162); FORCE_OPAQUE / CULL_DISABLE flags come from the instance.
void Scene::build()
{
  objectToWorld.resize(instances.size());
  worldToObject.resize(instances.size());
  tris.clear();
  for(size_t i = 0; i < instances.size(); i++) {
    affine M; memcpy(M.a, instances[i].objectToWorld, sizeof(M.a));
    float det;
    objectToWorld[i] = M;
    worldToObject[i] = inverseAffine(M, &det);
    const rt_prim_mesh& pm = primMeshes[instances[i].primMesh];
    uint32_t f = 0;
    if(instances[i].flags & RT_INST_FORCE_OPAQUE) f |= TRI_OPAQUE;
    if(instances[i].flags & RT_INST_CULL_DISABLE) f |= TRI_NOCULL;
    if(det < 0.0f) f |= TRI_FLIP;
    for(uint32_t p = 0; p < pm.indexCount / 3; p++) {
      const uint32_t* ix = &indices[pm.firstIndex + 3 * p];
      auto P = [&](uint32_t k) { const rt_vec3& q = vertices[pm.vertexOffset + ix[k]].position; return V3(q.x, q.y, q.z); };
      Tri T;
      T.v0 = xformPoint(M, P(0)); T.v1 = xformPoint(M, P(1)); T.v2 = xformPoint(M, P(2));
      T.inst = uint32_t(i); T.prim = p; T.flags = f;
      tris.push_back(T);
    }
  }

  // ---- naive BVH2: median split on the widest centroid axis, <= 4 triangles per leaf -------------
  const size_t n = tris.size();
  nodes.clear();
  if(n == 0) { nodes.push_back(BvhNode{{0, 0, 0}, {0, 0, 0}, 0, 0}); nodes[0].count = 0; nodes[0].left = 0; return; }
  float scale = 1e-3f;
  for(const Tri& T : tris)
    for(const vec3* v : {&T.v0, &T.v1, &T.v2}) scale = std::max(scale, std::max(std::fabs(v->x), std::max(std::fabs(v->y), std::fabs(v->z))));
  const float pad = 2e-5f * scale;  // makes box culling strictly weaker than the triangle test (DESIGN.md §Traversal soundness)
  triPad = pad;

  std::vector<uint32_t> order(n);
  std::iota(order.begin(), order.end(), 0u);
  std::vector<vec3> cen(n);
  for(size_t i = 0; i < n; i++) cen[i] = (tris[i].v0 + tris[i].v1 + tris[i].v2) * (1.0f / 3.0f);

  struct Job { uint32_t node, begin, end; };
  std::vector<Job> stack;
  nodes.reserve(2 * n / 2 + 16);
  nodes.push_back(BvhNode{});
  stack.push_back({0, 0, uint32_t(n)});
  auto bounds = [&](uint32_t b, uint32_t e, float* lo, float* hi) {
    for(int a = 0; a < 3; a++) { lo[a] = 3e38f; hi[a] = -3e38f; }
    for(uint32_t k = b; k < e; k++) {
      const Tri& T = tris[order[k]];
      for(const vec3* v : {&T.v0, &T.v1, &T.v2}) {
        const float c[3] = {v->x, v->y, v->z};
        for(int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], c[a] - pad); hi[a] = std::max(hi[a], c[a] + pad); }
      }
    }
  };
  while(!stack.empty()) {
    Job j = stack.back(); stack.pop_back();
    BvhNode nd{};
    bounds(j.begin, j.end, nd.lo, nd.hi);
    uint32_t cnt = j.end - j.begin;
    if(cnt <= 4) { nd.left = j.begin; nd.count = cnt; nodes[j.node] = nd; continue; }
    float clo[3] = {3e38f, 3e38f, 3e38f}, chi[3] = {-3e38f, -3e38f, -3e38f};
    for(uint32_t k = j.begin; k < j.end; k++) {
      const float c[3] = {cen[order[k]].x, cen[order[k]].y, cen[order[k]].z};
      for(int a = 0; a < 3; a++) { clo[a] = std::min(clo[a], c[a]); chi[a] = std::max(chi[a], c[a]); }
    }
    int ax = 0;
    if(chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
    if(chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
    uint32_t mid = j.begin + cnt / 2;
    std::nth_element(order.begin() + j.begin, order.begin() + mid, order.begin() + j.end, [&](uint32_t a, uint32_t b) {
      const float ca = ax == 0 ? cen[a].x : (ax == 1 ? cen[a].y : cen[a].z);
      const float cb = ax == 0 ? cen[b].x : (ax == 1 ? cen[b].y : cen[b].z);
      return ca < cb || (ca == cb && a < b);
    });
    nd.left = uint32_t(nodes.size()); nd.count = 0;
    nodes[j.node] = nd;
    nodes.push_back(BvhNode{}); nodes.push_back(BvhNode{});
    stack.push_back({nd.left, j.begin, mid});
    stack.push_back({nd.left + 1, mid, j.end});
  }
  // leaves index `order`; bake the permutation into a leaf-ordered triangle index list
  leafTris_.swap(order);
}

// Möller–Trumbore, fixed operation order.  Returns t,u,v without range checks on t.
bool Scene::intersectTri(const Tri& T, vec3 o, vec3 d, float& t, float& u, float& v) const
{
  vec3 e1 = T.v1 - T.v0, e2 = T.v2 - T.v0;
  vec3 p = cross(d, e2);
  float det = dot(e1, p);
  if(T.flags & TRI_NOCULL) { if(det == 0.0f || rt_isnan(det)) return false; }
  else {
    // gl_RayFlagsCullBackFacingTrianglesEXT: front face = counter-clockwise seen from the ray origin, decided
    // in object space (a mirroring instance transform flips the world-space sign => TRI_FLIP)
    float sdet = (T.flags & TRI_FLIP) ? -det : det;
    if(!(sdet > 0.0f)) return false;
  }
  float inv = 1.0f / det;
  vec3 tv = o - T.v0;
  u = dot(tv, p) * inv;
  if(!(u >= 0.0f && u <= 1.0f)) return false;
  vec3 q = cross(tv, e1);
  v = dot(d, q) * inv;
  if(!(v >= 0.0f && u + v <= 1.0f)) return false;
  t = dot(e2, q) * inv;
  if(rt_isnan(t)) return false;
  // an accepted hit point lies inside the triangle's box widened by the build's padding (csrc/traverse.h intersectTri: a sliver's
  // quotients are rounding noise; without this the result would depend on which triangles share a leaf with it)
  // The box is widened by the build's pad plus a term for the rounding error of the hit point itself (a far-away ray origin): the expression of
  // csrc/traverse.h hitPointPad, operation for operation.
  const vec3 h = o + d * t, w1 = T.v0 + e1, w2 = T.v0 + e2;
  const float om = rt_max(rt_max(rt_abs(o.x), rt_abs(o.y)), rt_abs(o.z)), dm = rt_max(rt_max(rt_abs(d.x), rt_abs(d.y)), rt_abs(d.z));
  const float pad = triPad + (om + t * dm) * 4.76837158203125e-07f;
  const vec3 lo = V3(rt_min(rt_min(T.v0.x, w1.x), w2.x) - pad, rt_min(rt_min(T.v0.y, w1.y), w2.y) - pad, rt_min(rt_min(T.v0.z, w1.z), w2.z) - pad);
  const vec3 hi = V3(rt_max(rt_max(T.v0.x, w1.x), w2.x) + pad, rt_max(rt_max(T.v0.y, w1.y), w2.y) + pad, rt_max(rt_max(T.v0.z, w1.z), w2.z) + pad);
  return h.x >= lo.x && h.x <= hi.x && h.y >= lo.y && h.y <= hi.y && h.z >= lo.z && h.z <= hi.z;
}

static inline uint32_t pcgStep(uint32_t& state)  // random.glsl:59-65
{
  uint32_t prev = state * 747796405u + 2891336453u;
  uint32_t word = ((prev >> ((prev >> 28u) + 4u)) ^ prev) * 277803737u;
  state = prev;
  return (word >> 22u) ^ word;
}

// HitTest (traceray_rq.glsl:32-102).  DEVIATION: the reference draws
// rand(prd.seed) once per candidate, which makes the RNG stream depend on the driver's candidate order.
// Here the draw comes from a hash of (ray seed, flattened triangle index) and prd.seed is not advanced.
bool Scene::hitTest(const Tri& T, uint32_t triIndex, float u, float v, uint32_t raySeed) const
{
  const rt_prim_mesh& pm = primMeshes[instances[T.inst].primMesh];
  const uint32_t matIndex = uint32_t(std::max(0, pm.materialIndex));
  const rt_material& mat = materials[matIndex];
  float baseColorAlpha = mat.pbrBaseColorFactor.w;
  if(mat.pbrBaseColorTexture > -1) {
    const uint32_t* ix = &indices[pm.firstIndex + 3 * T.prim];
    const rt_vertex& a0 = vertices[pm.vertexOffset + ix[0]];
    const rt_vertex& a1 = vertices[pm.vertexOffset + ix[1]];
    const rt_vertex& a2 = vertices[pm.vertexOffset + ix[2]];
    const vec3 bary = V3((1.0f - u) - v, u, v);
    vec2 uv = (V2(a0.texcoord.x, a0.texcoord.y) * bary.x + V2(a1.texcoord.x, a1.texcoord.y) * bary.y)
              + V2(a2.texcoord.x, a2.texcoord.y) * bary.z;
    baseColorAlpha = baseColorAlpha * sampleTexture(mat.pbrBaseColorTexture, uv).w;
  }
  float opacity;
  if(mat.alphaMode == RT_ALPHA_MASK) opacity = baseColorAlpha > mat.alphaCutoff ? 1.0f : 0.0f;
  else opacity = baseColorAlpha;
  uint32_t s = candidateSeed(raySeed, triIndex);
  float r = rt_u2f(0x3f800000u | (pcgStep(s) >> 9)) - 1.0f;
  return !(r > opacity);
}

namespace {
struct RaySlab {
  float o[3], d[3], inv[3];
  bool zero[3];
  RaySlab(vec3 O, vec3 D)
  {
    o[0] = O.x; o[1] = O.y; o[2] = O.z; d[0] = D.x; d[1] = D.y; d[2] = D.z;
    for(int a = 0; a < 3; a++) { zero[a] = !(std::fabs(d[a]) > 1e-30f); inv[a] = zero[a] ? 0.0f : 1.0f / d[a]; }
  }
  // returns entry distance, or -1 when the box is missed inside (0, tmax]
  bool hit(const BvhNode& n, float tmax, float& tn) const
  {
    float t0 = 0.0f, t1 = tmax;
    for(int a = 0; a < 3; a++) {
      if(zero[a]) { if(o[a] < n.lo[a] || o[a] > n.hi[a]) return false; continue; }
      float ta = (n.lo[a] - o[a]) * inv[a], tb = (n.hi[a] - o[a]) * inv[a];
      if(ta > tb) std::swap(ta, tb);
      if(ta > t0) t0 = ta;
      if(tb < t1) t1 = tb;
    }
    tn = t0;
    return t0 <= t1;
  }
};
}  // namespace

Hit Scene::closestHit(vec3 o, vec3 d, uint32_t raySeed) const
{
  Counters::local().closestHitRays++;
  Hit best;
  if(tris.empty() || hasNan(o) || hasNan(d)) return best;
  RaySlab R(o, d);
  uint32_t stack[128]; int sp = 0;
  uint64_t nv = 0, tt = 0;
  float tn;
  if(R.hit(nodes[0], best.t, tn)) stack[sp++] = 0;
  while(sp) {
    const BvhNode& n = nodes[stack[--sp]];
    nv++;
    if(!R.hit(n, best.t, tn)) continue;
    if(n.count) {
      for(uint32_t k = 0; k < n.count; k++) {
        uint32_t ti = leafTris_[n.left + k];
        const Tri& T = tris[ti];
        float t, u, v;
        tt++;
        if(!intersectTri(T, o, d, t, u, v)) continue;
        if(!(t > 0.0f && t < RT_INFINITY)) continue;
        if(!(t < best.t || (t == best.t && ti < best.tri))) continue;
        if(!(T.flags & TRI_OPAQUE) && !hitTest(T, ti, u, v, raySeed)) continue;
        best.t = t; best.tri = ti; best.u = u; best.v = v;
      }
    } else {
      float ta, tb;
      bool ha = R.hit(nodes[n.left], best.t, ta), hb = R.hit(nodes[n.left + 1], best.t, tb);
      if(ha && hb) {
        if(ta <= tb) { stack[sp++] = n.left + 1; stack[sp++] = n.left; }
        else { stack[sp++] = n.left; stack[sp++] = n.left + 1; }
      } else if(ha) stack[sp++] = n.left;
      else if(hb) stack[sp++] = n.left + 1;
    }
  }
  Counters::local().nodesVisited += nv; Counters::local().trisTested += tt;
  return best;
}

bool Scene::anyHit(vec3 o, vec3 d, float tmax, uint32_t raySeed) const
{
  Counters::local().anyHitRays++;
  if(tris.empty() || hasNan(o) || hasNan(d) || !(tmax > 0.0f)) return false;
  RaySlab R(o, d);
  uint32_t stack[128]; int sp = 0;
  uint64_t nv = 0, tt = 0;
  bool found = false;
  stack[sp++] = 0;
  float tn;
  while(sp && !found) {
    const BvhNode& n = nodes[stack[--sp]];
    nv++;
    if(!R.hit(n, tmax, tn)) continue;
    if(n.count) {
      for(uint32_t k = 0; k < n.count && !found; k++) {
        uint32_t ti = leafTris_[n.left + k];
        const Tri& T = tris[ti];
        float t, u, v;
        tt++;
        if(!intersectTri(T, o, d, t, u, v)) continue;
        if(!(t > 0.0f && t < tmax)) continue;
        if(!(T.flags & TRI_OPAQUE) && !hitTest(T, ti, u, v, raySeed)) continue;
        found = true;
      }
    } else { stack[sp++] = n.left; stack[sp++] = n.left + 1; }
  }
  Counters::local().nodesVisited += nv; Counters::local().trisTested += tt;
  return found;
}

}  // namespace orc
