// scene_gen.cpp — seeded procedural stand-ins for the assets the benchmark configurations name.
// The reference's CMake downloads robot_toon.zip / daytime.hdr / std_env.hdr (CMakeLists.txt:51-52) and the README
// numbers use Sponza and Bistro; none of these exist here and there is no network, so each
// BASELINE.json config gets a generator with the same triangle count class, material/texture mix, emissive-mesh
// count and alpha-masked foliage fraction.  Everything is a function of (kind, scale, seed).
#include "scene.hpp"
#include "../../include/rt_cpus.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <thread>

namespace rth {
namespace {

struct Rng {
  uint32_t s;
  explicit Rng(uint32_t seed) : s(seed * 747796405u + 2891336453u) {}
  uint32_t next() { s = s * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }
  float uni() { return float(next() >> 8) * (1.0f / 16777216.0f); }
  float range(float a, float b) { return a + (b - a) * uni(); }
};

M4 translate(V3 t) { M4 m = M4::identity(); m.at(0, 3) = t.x; m.at(1, 3) = t.y; m.at(2, 3) = t.z; return m; }
M4 scaleM(V3 s) { M4 m = M4::identity(); m.at(0, 0) = s.x; m.at(1, 1) = s.y; m.at(2, 2) = s.z; return m; }
M4 rotateY(float a) { M4 m = M4::identity(); float c = std::cos(a), s = std::sin(a); m.at(0, 0) = c; m.at(0, 2) = s; m.at(2, 0) = -s; m.at(2, 2) = c; return m; }

float hashNoise(int x, int y, uint32_t seed)
{
  uint32_t h = uint32_t(x) * 374761393u + uint32_t(y) * 668265263u + seed * 2246822519u;
  h = (h ^ (h >> 13)) * 1274126177u;
  return float((h ^ (h >> 16)) & 0xffffffu) * (1.0f / 16777215.0f);
}
float valueNoise(float x, float y, uint32_t seed)
{
  int xi = int(std::floor(x)), yi = int(std::floor(y));
  float fx = x - xi, fy = y - yi;
  fx = fx * fx * (3 - 2 * fx); fy = fy * fy * (3 - 2 * fy);
  float a = hashNoise(xi, yi, seed), b = hashNoise(xi + 1, yi, seed), c = hashNoise(xi, yi + 1, seed), d = hashNoise(xi + 1, yi + 1, seed);
  return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy;
}
float fbm(float x, float y, uint32_t seed) { return 0.5f * valueNoise(x, y, seed) + 0.3f * valueNoise(2 * x, 2 * y, seed + 1) + 0.2f * valueNoise(4 * x, 4 * y, seed + 2); }

struct Builder {
  GltfScene g;
  Rng rng;
  uint32_t meshVertexOffset = 0, meshFirstIndex = 0;
  int meshMaterial = 0;
  explicit Builder(uint32_t seed) : rng(seed) {}

  int addMaterial(const GltfMaterial& m) { g.materials.push_back(m); return int(g.materials.size()) - 1; }
  int addTexture(TextureImage&& t) { g.textures.push_back(std::move(t)); return int(g.textures.size()) - 1; }

  void beginMesh(int material) { meshVertexOffset = uint32_t(g.positions.size()); meshFirstIndex = uint32_t(g.indices.size()); meshMaterial = material; }
  int endMesh()
  {
    GltfPrimMesh pm;
    pm.vertexOffset = meshVertexOffset; pm.vertexCount = uint32_t(g.positions.size()) - meshVertexOffset;
    pm.firstIndex = meshFirstIndex; pm.indexCount = uint32_t(g.indices.size()) - meshFirstIndex;
    pm.materialIndex = meshMaterial;
    g.primMeshes.push_back(pm);
    return int(g.primMeshes.size()) - 1;
  }
  int addNode(int primMesh, const M4& m = M4::identity()) { GltfNode n; n.primMesh = primMesh; n.worldMatrix = m; g.nodes.push_back(n); return int(g.nodes.size()) - 1; }

  uint32_t vert(V3 p, V3 n, float u, float v, V3 t = V3{0, 0, 0})
  {
    if(length(t) < 1e-6f) { V3 c1 = cross(n, V3{0, 0, 1}), c2 = cross(n, V3{0, 1, 0}); t = normalize(length(c1) > length(c2) ? c1 : c2); }
    g.positions.push_back(p); g.normals.push_back(n); g.texcoords0.push_back({u, v});
    g.tangents.push_back({t.x, t.y, t.z, 1.f}); g.colors0.push_back({1.f, 1.f, 1.f, 1.f});
    return uint32_t(g.positions.size()) - 1 - meshVertexOffset;
  }
  void tri(uint32_t a, uint32_t b, uint32_t c) { g.indices.push_back(a); g.indices.push_back(b); g.indices.push_back(c); }

  // planar quad p0..p3 (any consistent order); wound so that the geometric normal agrees with `n`
  void quad(V3 p0, V3 p1, V3 p2, V3 p3, V3 n, float uvScale = 1.f)
  {
    if(dot(cross(p1 - p0, p2 - p0), n) < 0) std::swap(p1, p3);
    V3 t = normalize(p1 - p0);
    uint32_t a = vert(p0, n, 0, 0, t), b = vert(p1, n, uvScale, 0, t), c = vert(p2, n, uvScale, uvScale, t), d = vert(p3, n, 0, uvScale, t);
    tri(a, b, c); tri(a, c, d);
  }
  // experiment (RESTIR_SCENE_TESS_LEAVES=n, profiles/r04_leaf_blocks_ab.txt): the same cut-out quad as n x n sub-quads, without the sub-quads whose texels are
  // all transparent — what clipping alpha-masked geometry to its opaque part could buy in traversal steps.  A different scene (other triangle ids), never a default.
  void quadTessellated(V3 p0, V3 p1, V3 p2, V3 p3, V3 n, int nsub, const TextureImage& tex)
  {
    if(dot(cross(p1 - p0, p2 - p0), n) < 0) std::swap(p1, p3);
    V3 t = normalize(p1 - p0);
    for(int j = 0; j < nsub; j++)
      for(int i = 0; i < nsub; i++) {
        const float s0 = float(i) / nsub, s1 = float(i + 1) / nsub, t0 = float(j) / nsub, t1 = float(j + 1) / nsub;
        bool any = false;
        const int x0 = std::max(0, int(s0 * tex.width) - 2), x1 = std::min(tex.width - 1, int(s1 * tex.width) + 2);
        const int y0 = std::max(0, int(t0 * tex.height) - 2), y1 = std::min(tex.height - 1, int(t1 * tex.height) + 2);
        for(int y = y0; y <= y1 && !any; y++) for(int x = x0; x <= x1; x++) if(tex.bgra[(size_t(y) * tex.width + x) * 4 + 3] != 0) { any = true; break; }
        if(!any) continue;
        auto P = [&](float s, float tt) { return p0 + (p1 - p0) * s + (p3 - p0) * tt; };
        uint32_t a = vert(P(s0, t0), n, s0, t0, t), b = vert(P(s1, t0), n, s1, t0, t), c = vert(P(s1, t1), n, s1, t1, t), d = vert(P(s0, t1), n, s0, t1, t);
        tri(a, b, c); tri(a, c, d);
      }
  }
  // axis-aligned box, optionally rotated about Y, faces outward (inward when `inside`)
  void box(V3 c, V3 h, float rotY = 0.f, bool inside = false, float uvScale = 1.f)
  {
    M4 R = rotateY(rotY);
    auto P = [&](float x, float y, float z) { return xformPoint(R, V3{x * h.x, y * h.y, z * h.z}) + c; };
    auto N = [&](float x, float y, float z) { V3 n = xformDir(R, V3{x, y, z}); return inside ? n * -1.f : n; };
    quad(P(-1, -1, 1), P(1, -1, 1), P(1, 1, 1), P(-1, 1, 1), N(0, 0, 1), uvScale);
    quad(P(-1, -1, -1), P(1, -1, -1), P(1, 1, -1), P(-1, 1, -1), N(0, 0, -1), uvScale);
    quad(P(1, -1, -1), P(1, -1, 1), P(1, 1, 1), P(1, 1, -1), N(1, 0, 0), uvScale);
    quad(P(-1, -1, -1), P(-1, -1, 1), P(-1, 1, 1), P(-1, 1, -1), N(-1, 0, 0), uvScale);
    quad(P(-1, 1, -1), P(1, 1, -1), P(1, 1, 1), P(-1, 1, 1), N(0, 1, 0), uvScale);
    quad(P(-1, -1, -1), P(1, -1, -1), P(1, -1, 1), P(-1, -1, 1), N(0, -1, 0), uvScale);
  }
  // square-section beam from a to b (rails, balusters, cable segments, chair legs): 12 long thin triangles in any orientation
  void beam(V3 a, V3 b, float half, float uvScale = 1.f)
  {
    V3 d = normalize(b - a);
    V3 s = std::fabs(d.y) < 0.9f ? normalize(cross(d, V3{0, 1, 0})) : normalize(cross(d, V3{1, 0, 0}));
    V3 t = cross(d, s);
    s = s * half; t = t * half;
    quad(a - s - t, b - s - t, b + s - t, a + s - t, t * (-1.f / half), uvScale);
    quad(a - s + t, b - s + t, b + s + t, a + s + t, t * (1.f / half), uvScale);
    quad(a - s - t, b - s - t, b - s + t, a - s + t, s * (-1.f / half), uvScale);
    quad(a + s - t, b + s - t, b + s + t, a + s + t, s * (1.f / half), uvScale);
    quad(a - s - t, a + s - t, a + s + t, a - s + t, d * -1.f, uvScale);
    quad(b - s - t, b + s - t, b + s + t, b - s + t, d, uvScale);
  }
  // a parallelogram o + s*ax + t*ay as `n` strips along ax (awning fabric, shutters): 2n triangles |ay| long and |ax|/n wide, UVs over the whole sheet
  void strips(V3 o, V3 ax, V3 ay, V3 nrm, int n)
  {
    if(dot(cross(ax, ay), nrm) < 0) { o = o + ax; ax = ax * -1.f; }
    V3 t = normalize(ax);
    for(int i = 0; i < n; i++) {
      const float s0 = float(i) / n, s1 = float(i + 1) / n;
      uint32_t a = vert(o + ax * s0, nrm, s0, 0, t), b = vert(o + ax * s1, nrm, s1, 0, t), c = vert(o + ax * s1 + ay, nrm, s1, 1, t), d = vert(o + ax * s0 + ay, nrm, s0, 1, t);
      tri(a, b, c); tri(a, c, d);
    }
  }
  // UV sphere with optional noise displacement; nu x nv quads => 2*nu*nv triangles
  void sphere(V3 c, V3 radii, int nu, int nv, float disp = 0.f, uint32_t seed = 0)
  {
    uint32_t base = uint32_t(g.positions.size()) - meshVertexOffset;
    for(int j = 0; j <= nv; j++)
      for(int i = 0; i <= nu; i++) {
        float u = float(i) / nu, v = float(j) / nv;
        float th = v * 3.14159265f, ph = u * 6.2831853f;
        V3 n{std::sin(th) * std::cos(ph), std::cos(th), std::sin(th) * std::sin(ph)};
        float r = 1.f + (disp > 0 ? disp * (fbm(u * 12.f, v * 6.f, seed) - 0.5f) : 0.f);
        V3 t{-std::sin(ph), 0, std::cos(ph)};
        vert(V3{n.x * radii.x * r, n.y * radii.y * r, n.z * radii.z * r} + c, n, u * 2.f, v, t);
      }
    for(int j = 0; j < nv; j++)
      for(int i = 0; i < nu; i++) {
        uint32_t a = base + j * (nu + 1) + i, b = a + 1, d = a + (nu + 1), e = d + 1;
        tri(a, b, e); tri(a, e, d);
      }
  }
  // cylinder along Y with caps omitted (columns, trunks, poles); ns segments x nr rings
  void cylinder(V3 c, float radius, float height, int ns, int nr, float flute = 0.f)
  {
    uint32_t base = uint32_t(g.positions.size()) - meshVertexOffset;
    for(int j = 0; j <= nr; j++)
      for(int i = 0; i <= ns; i++) {
        float u = float(i) / ns, v = float(j) / nr, ph = u * 6.2831853f;
        float r = radius * (1.f + flute * std::cos(ph * 12.f)) * (1.f - 0.15f * v);
        V3 n{std::cos(ph), 0, std::sin(ph)};
        vert(V3{n.x * r, v * height, n.z * r} + c, n, u * 3.f, v * height, V3{-std::sin(ph), 0, std::cos(ph)});
      }
    for(int j = 0; j < nr; j++)
      for(int i = 0; i < ns; i++) {
        uint32_t a = base + j * (ns + 1) + i, b = a + 1, d = a + (ns + 1), e = d + 1;
        tri(a, e, b); tri(a, d, e);
      }
  }
  // height-field grid in the XZ plane (ground, drapes when rotated through `frame`)
  void grid(V3 origin, V3 ax, V3 az, V3 up, int nx, int nz, float amp, float freq, uint32_t seed, float uvScale)
  {
    uint32_t base = uint32_t(g.positions.size()) - meshVertexOffset;
    auto H = [&](float u, float v) { return amp * (fbm(u * freq, v * freq, seed) - 0.5f); };
    for(int j = 0; j <= nz; j++)
      for(int i = 0; i <= nx; i++) {
        float u = float(i) / nx, v = float(j) / nz, e = 1.f / float(std::max(nx, nz));
        float h = H(u, v), hx = H(u + e, v), hz = H(u, v + e);
        V3 p = origin + ax * u + az * v + up * h;
        V3 dx = ax * e + up * (hx - h), dz = az * e + up * (hz - h);
        V3 n = normalize(cross(dz, dx));
        if(dot(n, up) < 0) n = n * -1.f;
        vert(p, n, u * uvScale, v * uvScale, normalize(ax));
      }
    bool flip = dot(cross(ax, az), up) > 0;  // keep CCW as seen from +up
    for(int j = 0; j < nz; j++)
      for(int i = 0; i < nx; i++) {
        uint32_t a = base + j * (nx + 1) + i, b = a + 1, d = a + (nx + 1), e = d + 1;
        if(flip) { tri(a, b, e); tri(a, e, d); } else { tri(a, e, b); tri(a, d, e); }
      }
  }
};

// rows [y0, y1) of a procedural texture.  kind 0-5: the round-1 patterns (bit-identical: every generated scene of rounds 1-4 keeps its texels);
// kind 6: cut-out with `tint[3]`-selected silhouette variants (the "real footprint" scenes: 16 distinct foliage cards)
void fillTextureRows(TextureImage& t, int kind, uint32_t seed, const float tint[3], int y0, int y1, int variant = 0)
{
  const int size = t.width;
  for(int y = y0; y < y1; y++)
    for(int x = 0; x < size; x++) {
      float u = float(x) / size, v = float(y) / size;
      float r = 1, g = 1, b = 1, a = 1;
      switch(kind) {
        case 0: { float n = 0.55f + 0.45f * fbm(u * 16, v * 16, seed); r = tint[0] * n; g = tint[1] * n; b = tint[2] * n; break; }                 // mottled
        case 1: { bool c = ((int(u * 8) + int(v * 8)) & 1) != 0; float n = c ? 0.9f : 0.35f; r = tint[0] * n; g = tint[1] * n; b = tint[2] * n; break; }  // checker
        case 2: {  // bricks
          float row = std::floor(v * 16), bu = u * 8 + (int(row) & 1) * 0.5f;
          bool mortar = (v * 16 - row) < 0.08f || (bu - std::floor(bu)) < 0.04f;
          float n = mortar ? 0.75f : (0.45f + 0.4f * hashNoise(int(std::floor(bu)), int(row), seed));
          r = mortar ? n : tint[0] * n; g = mortar ? n : tint[1] * n; b = mortar ? n : tint[2] * n; break;
        }
        case 3: {  // leaf cut-out: alpha = 1 inside a lobed shape
          float cx = u - 0.5f, cy = v - 0.5f, rad = std::sqrt(cx * cx + cy * cy), ang = std::atan2(cy, cx);
          float lim = 0.32f + 0.12f * std::cos(ang * 5.f) + 0.05f * fbm(u * 9, v * 9, seed);
          a = rad < lim ? 1.f : 0.f;
          float n = 0.6f + 0.4f * fbm(u * 20, v * 20, seed);
          r = tint[0] * n; g = tint[1] * n; b = tint[2] * n; break;
        }
        case 4: {  // tangent-space normal map from a height field
          float e = 1.f / size, h0 = fbm(u * 24, v * 24, seed), hx = fbm((u + e) * 24, v * 24, seed), hy = fbm(u * 24, (v + e) * 24, seed);
          float nx = (h0 - hx) * 6.f, ny = (h0 - hy) * 6.f, nz = 1.f, l = std::sqrt(nx * nx + ny * ny + nz * nz);
          r = nx / l * 0.5f + 0.5f; g = ny / l * 0.5f + 0.5f; b = nz / l * 0.5f + 0.5f; break;
        }
        case 5: { float n = fbm(u * 10, v * 10, seed); r = 1; g = 0.25f + 0.7f * n; b = n > 0.6f ? 1.f : 0.f; break; }  // occlusion/roughness/metallic
        case 6: {  // foliage card: `variant` picks the silhouette — one lobed leaf (3..8 lobes), or a spray of 3..5 small leaves on a stem (coverage 18..40 %)
          const int lobes = 3 + variant % 6;
          float n = 0.6f + 0.4f * fbm(u * 20, v * 20, seed);
          r = tint[0] * n; g = tint[1] * n; b = tint[2] * n;
          if(variant < 8) {
            float cx = u - 0.5f, cy = v - 0.5f, rad = std::sqrt(cx * cx + cy * cy), ang = std::atan2(cy, cx);
            float lim = 0.27f + 0.02f * (variant & 3) + (0.08f + 0.01f * variant) * std::cos(ang * float(lobes)) + 0.06f * fbm(u * 9, v * 9, seed);
            a = rad < lim ? 1.f : 0.f;
          } else {
            a = 0.f;
            const int leaves = 3 + variant % 3;
            for(int k = 0; k < leaves; k++) {
              const float la = 6.2831853f * (float(k) + 0.3f * hashNoise(k, variant, seed)) / float(leaves);
              const float lx = 0.5f + 0.24f * std::cos(la), ly = 0.5f + 0.24f * std::sin(la);
              // an ellipse pointing away from the centre
              const float dx = u - lx, dy = v - ly, al = dx * std::cos(la) + dy * std::sin(la), ac = -dx * std::sin(la) + dy * std::cos(la);
              const float w = 0.075f + 0.02f * hashNoise(k, 7, seed);
              if((al * al) / (0.2f * 0.2f) + (ac * ac) / (w * w) < 1.f + 0.25f * (fbm(u * 30, v * 30, seed + 5) - 0.5f)) a = 1.f;
              // stem towards the centre
              const float tt = ((u - 0.5f) * (lx - 0.5f) + (v - 0.5f) * (ly - 0.5f)) / (0.24f * 0.24f);
              if(tt > 0.f && tt < 1.f) { const float px = (u - 0.5f) - tt * (lx - 0.5f), py = (v - 0.5f) - tt * (ly - 0.5f); if(px * px + py * py < 0.008f * 0.008f) { a = 1.f; r *= 0.5f; g *= 0.45f; } }
            }
          }
          break;
        }
        case 7: {  // painted / weathered wall: large stains over fine grain, faint horizontal courses
          float st = fbm(u * 3, v * 3, seed), gr = fbm(u * 64, v * 64, seed + 9), course = (v * 24 - std::floor(v * 24)) < 0.06f ? 0.85f : 1.f;
          float n = (0.5f + 0.35f * st + 0.15f * gr) * course; r = tint[0] * n; g = tint[1] * n; b = tint[2] * n; break;
        }
        case 8: {  // cobbles / pavers: offset rows of rounded stones
          float row = std::floor(v * 32), bu = u * 32 + (int(row) & 1) * 0.5f, fx = bu - std::floor(bu) - 0.5f, fy = v * 32 - row - 0.5f;
          float d = std::sqrt(fx * fx + fy * fy), n = d > 0.46f ? 0.3f : (0.55f + 0.35f * hashNoise(int(std::floor(bu)), int(row), seed)) * (1.f - 0.5f * d);
          n *= 0.8f + 0.2f * fbm(u * 48, v * 48, seed + 3); r = tint[0] * n; g = tint[1] * n; b = tint[2] * n; break;
        }
        case 9: {  // striped fabric (awnings)
          bool c = (int(u * 24) & 1) != 0; float n = (c ? 0.95f : 0.6f) * (0.85f + 0.15f * fbm(u * 40, v * 40, seed));
          r = c ? n : tint[0] * n; g = c ? n : tint[1] * n; b = c ? n : tint[2] * n; break;
        }
      }
      uint8_t* p = &t.bgra[(size_t(y) * size + x) * 4];
      auto q = [](float f) { f = f < 0 ? 0 : (f > 1 ? 1 : f); return uint8_t(f * 255.f + 0.5f); };
      p[0] = q(b); p[1] = q(g); p[2] = q(r); p[3] = q(a);  // BGRA (scene.cpp:559)
    }
}
TextureImage makeTexture(int size, int kind, uint32_t seed, const float tint[3])
{
  TextureImage t; t.width = t.height = size; t.bgra.resize(size_t(size) * size * 4);
  fillTextureRows(t, kind, seed, tint, 0, size);
  return t;
}
// The real-footprint scenes hold gigabytes of texels (below): their textures are declared first and filled by every host thread afterwards, 32 rows at a time.
struct TexJob { int index, kind, variant; uint32_t seed; float tint[3]; };
int declareTexture(Builder& B, std::vector<TexJob>& jobs, int size, int kind, uint32_t seed, const float tint[3], int variant = 0)
{
  TextureImage t; t.width = t.height = size; t.bgra.resize(size_t(size) * size * 4);
  const int id = B.addTexture(std::move(t));
  jobs.push_back(TexJob{id, kind, variant, seed, {tint[0], tint[1], tint[2]}});
  return id;
}
void fillTextures(Builder& B, const std::vector<TexJob>& jobs)
{
  struct Chunk { int job, y0, y1; };
  std::vector<Chunk> chunks;
  for(size_t j = 0; j < jobs.size(); j++) {
    const int h = B.g.textures[size_t(jobs[j].index)].height;
    for(int y = 0; y < h; y += 32) chunks.push_back({int(j), y, std::min(h, y + 32)});
  }
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for(size_t c; (c = next.fetch_add(1)) < chunks.size();) {
      const TexJob& J = jobs[size_t(chunks[c].job)];
      fillTextureRows(B.g.textures[size_t(J.index)], J.kind, J.seed, J.tint, chunks[c].y0, chunks[c].y1, J.variant);
    }
  };
  const unsigned nt = unsigned(std::min(256, rt_cpu_budget()));
  std::vector<std::thread> pool;
  for(unsigned i = 1; i < nt; i++) pool.emplace_back(work);
  work();
  for(auto& th : pool) th.join();
}

GltfMaterial diffuse(float r, float g, float b, float rough = 1.f, float metal = 0.f)
{
  GltfMaterial m; m.baseColorFactor[0] = r; m.baseColorFactor[1] = g; m.baseColorFactor[2] = b; m.roughnessFactor = rough; m.metallicFactor = metal;
  return m;
}
GltfMaterial emissive(float r, float g, float b)
{
  GltfMaterial m = diffuse(0, 0, 0); m.emissiveFactor[0] = r; m.emissiveFactor[1] = g; m.emissiveFactor[2] = b;
  return m;
}

// ---- config 2: Cornell box --------------------------------------------------------------------------------
GltfScene makeCornell()
{
  Builder B(1);
  int white = B.addMaterial(diffuse(0.73f, 0.73f, 0.73f)), red = B.addMaterial(diffuse(0.65f, 0.05f, 0.05f)), green = B.addMaterial(diffuse(0.12f, 0.45f, 0.15f));
  int light = B.addMaterial(emissive(17.f, 12.f, 4.f));
  B.beginMesh(white);
  B.quad({-1, 0, 1}, {1, 0, 1}, {1, 0, -1}, {-1, 0, -1}, {0, 1, 0});   // floor
  B.quad({-1, 2, 1}, {1, 2, 1}, {1, 2, -1}, {-1, 2, -1}, {0, -1, 0});  // ceiling
  B.quad({-1, 0, -1}, {1, 0, -1}, {1, 2, -1}, {-1, 2, -1}, {0, 0, 1}); // back wall
  B.addNode(B.endMesh());
  B.beginMesh(red); B.quad({-1, 0, -1}, {-1, 0, 1}, {-1, 2, 1}, {-1, 2, -1}, {1, 0, 0}); B.addNode(B.endMesh());
  B.beginMesh(green); B.quad({1, 0, -1}, {1, 0, 1}, {1, 2, 1}, {1, 2, -1}, {-1, 0, 0}); B.addNode(B.endMesh());
  B.beginMesh(white); B.box({0.33f, 0.3f, 0.35f}, {0.3f, 0.3f, 0.3f}, -0.3f); B.addNode(B.endMesh());  // short box
  B.beginMesh(white); B.box({-0.35f, 0.6f, -0.3f}, {0.3f, 0.6f, 0.3f}, 0.3f); B.addNode(B.endMesh());  // tall box
  B.beginMesh(light); B.quad({-0.25f, 1.98f, -0.25f}, {0.25f, 1.98f, -0.25f}, {0.25f, 1.98f, 0.25f}, {-0.25f, 1.98f, 0.25f}, {0, -1, 0}); B.addNode(B.endMesh());
  GltfCamera cam; cam.eye = {0, 1, 3.4f}; cam.center = {0, 1, 0}; cam.yfovDeg = 40.f;
  B.g.cameras.push_back(cam);
  return std::move(B.g);
}

// ---- config 1: "DamagedHelmet-class" — one textured, displaced ~70k-triangle mesh -----------------------------
GltfScene makeHelmet(float scale, uint32_t seed)
{
  Builder B(seed);
  const float tint[3] = {0.8f, 0.7f, 0.6f}, one[3] = {1, 1, 1};
  GltfMaterial m = diffuse(1, 1, 1, 1.f, 1.f);
  m.baseColorTexture = B.addTexture(makeTexture(512, 0, seed, tint));
  m.metallicRoughnessTexture = B.addTexture(makeTexture(512, 5, seed + 3, one));
  m.normalTexture = B.addTexture(makeTexture(512, 4, seed + 5, one));
  int mat = B.addMaterial(m);
  int nu = std::max(8, int(190 * std::sqrt(scale))), nv = std::max(6, int(185 * std::sqrt(scale)));
  B.beginMesh(mat); B.sphere({0, 0, 0}, {1.f, 1.1f, 1.f}, nu, nv, 0.25f, seed); B.addNode(B.endMesh(), translate({-1, 2, -1}));
  int lamp = B.addMaterial(emissive(30, 28, 25));
  B.beginMesh(lamp); B.quad({-2.f, 5.f, -2.f}, {0.f, 5.f, -2.f}, {0.f, 5.f, 0.f}, {-2.f, 5.f, 0.f}, {0, -1, 0}); B.addNode(B.endMesh());
  GltfCamera cam; cam.eye = {2, 2, -5}; cam.center = {-1, 2, -1}; cam.yfovDeg = 45.f;  // main.cpp:68 default look-at
  B.g.cameras.push_back(cam);
  return std::move(B.g);
}

struct Palette { std::vector<int> opaque; int leaf = 0, glass = 0, lamp = 0, metal = 0; };
Palette makePalette(Builder& B, int numTextured, int texSize, uint32_t seed, float lampR, float lampG, float lampB)
{
  Palette P;
  const float one[3] = {1, 1, 1};
  for(int i = 0; i < numTextured; i++) {
    float tint[3] = {B.rng.range(0.35f, 0.95f), B.rng.range(0.35f, 0.9f), B.rng.range(0.3f, 0.85f)};
    GltfMaterial m = diffuse(1, 1, 1, B.rng.range(0.2f, 1.f), B.rng.uni() < 0.2f ? 1.f : 0.f);  // SURVEY §8d value distributions
    m.baseColorTexture = B.addTexture(makeTexture(texSize, i % 3, seed + 11u * i, tint));
    if(i % 4 == 0) m.normalTexture = B.addTexture(makeTexture(texSize, 4, seed + 101u * i, one));
    if(i % 5 == 0) m.metallicRoughnessTexture = B.addTexture(makeTexture(texSize, 5, seed + 1001u * i, one));
    P.opaque.push_back(B.addMaterial(m));
  }
  const float leafTint[3] = {0.25f, 0.55f, 0.18f};
  GltfMaterial leaf = diffuse(1, 1, 1, 0.8f);
  leaf.baseColorTexture = B.addTexture(makeTexture(texSize / 2, 3, seed + 77, leafTint));
  leaf.alphaMode = RT_ALPHA_MASK; leaf.alphaCutoff = 0.5f; leaf.doubleSided = 1;
  if(getenv("RESTIR_DEBUG_OPAQUE_LEAVES")) leaf.alphaMode = RT_ALPHA_OPAQUE;  // experiment switch: cost of the alpha test
  P.leaf = B.addMaterial(leaf);
  GltfMaterial glass = diffuse(0.6f, 0.7f, 0.75f, 0.05f, 1.f);
  P.glass = B.addMaterial(glass);
  P.metal = B.addMaterial(diffuse(0.9f, 0.85f, 0.7f, 0.25f, 1.f));
  P.lamp = B.addMaterial(emissive(lampR, lampG, lampB));
  return P;
}

// ---- config 3: "Sponza-class" atrium, ~262k triangles at scale 1 ------------------------------------------------
GltfScene makeSponza(float scale, uint32_t seed, int texSize = 512)
{
  Builder B(seed);
  Palette P = makePalette(B, 25, texSize, seed, 40.f, 34.f, 26.f);
  float s = std::sqrt(scale);
  auto mat = [&](int i) { return P.opaque[size_t(i) % P.opaque.size()]; };
  B.beginMesh(mat(0)); B.grid({-15, 0, -6}, {30, 0, 0}, {0, 0, 12}, {0, 1, 0}, std::max(2, int(200 * s)), std::max(2, int(100 * s)), 0.04f, 40.f, seed, 30.f); B.addNode(B.endMesh());
  // enclosing walls (inward-facing box shell without a roof: the sky lights the atrium)
  B.beginMesh(mat(1));
  B.quad({-15, 0, -6}, {15, 0, -6}, {15, 10, -6}, {-15, 10, -6}, {0, 0, 1}, 10.f);
  B.quad({-15, 0, 6}, {15, 0, 6}, {15, 10, 6}, {-15, 10, 6}, {0, 0, -1}, 10.f);
  B.quad({-15, 0, -6}, {-15, 0, 6}, {-15, 10, 6}, {-15, 10, -6}, {1, 0, 0}, 6.f);
  B.quad({15, 0, -6}, {15, 0, 6}, {15, 10, 6}, {15, 10, -6}, {-1, 0, 0}, 6.f);
  B.addNode(B.endMesh());
  // two storeys of fluted columns: one prim mesh, instanced 48 times
  B.beginMesh(mat(2)); B.cylinder({0, 0, 0}, 0.35f, 4.2f, std::max(6, int(48 * s)), std::max(2, int(32 * s)), 0.06f); int column = B.endMesh();
  for(int storey = 0; storey < 2; storey++)
    for(int i = 0; i < 12; i++)
      for(int side = 0; side < 2; side++) B.addNode(column, translate({-13.f + i * 2.36f, storey * 4.6f, side ? 4.2f : -4.2f}));
  // gallery slabs + balustrades
  B.beginMesh(mat(3));
  for(int side = 0; side < 2; side++) { B.box({0, 4.4f, side ? 5.1f : -5.1f}, {15, 0.2f, 0.9f}, 0, false, 8.f); B.box({0, 5.1f, side ? 4.25f : -4.25f}, {15, 0.45f, 0.05f}, 0, false, 8.f); }
  B.addNode(B.endMesh());
  // drapes: wavy grids hanging between columns
  for(int k = 0; k < 6; k++) {
    B.beginMesh(mat(4 + k));
    int n = std::max(2, int(80 * s));
    B.grid({-12.f + k * 4.2f, 8.4f, (k & 1) ? 3.6f : -3.6f}, {2.8f, 0, 0}, {0, -3.6f, 0}, {0, 0, 1}, n, n, 0.5f, 5.f, seed + k, 2.f);
    B.addNode(B.endMesh());
  }
  // one emissive mesh (a row of lanterns) — "1 emissive mesh" of config 3
  B.beginMesh(P.lamp);
  for(int i = 0; i < 6; i++) B.sphere({-10.f + i * 4.f, 3.6f, 0.f}, {0.18f, 0.18f, 0.18f}, 6, 4);
  B.addNode(B.endMesh());
  GltfCamera cam; cam.eye = {-12.5f, 2.2f, 0.6f}; cam.center = {6.f, 3.2f, -0.4f}; cam.yfovDeg = 60.f;
  B.g.cameras.push_back(cam);
  return std::move(B.g);
}

// ---- configs 4/5: "Bistro-class" street / interior ------------------------------------------------------------
GltfScene makeBistro(bool interior, float scale, uint32_t seed)
{
  // RESTIR_SCENE_STRESS=1: a harder variant of the exterior scene for robustness measurements: rotated buildings, twice the foliage, wires.
  // The benchmark scene (unset) is unchanged, down to the order of its random draws.
  const bool stress = !interior && getenv("RESTIR_SCENE_STRESS") != nullptr;
  Builder B(seed);
  Palette P = interior ? makePalette(B, 30, 512, seed, 26.f, 22.f, 16.f) : makePalette(B, 40, 512, seed, 60.f, 48.f, 30.f);
  float s = std::sqrt(scale);
  auto mat = [&](int i) { return P.opaque[size_t(i) % P.opaque.size()]; };
  const float X = interior ? 10.f : 60.f, Z = interior ? 6.f : 40.f;
  // ground / floor: displaced grid (cobblestones), the largest single mesh
  {
    int nx = std::max(2, int((interior ? 300 : 600) * s)), nz = std::max(2, int((interior ? 200 : 600) * s));
    B.beginMesh(mat(0)); B.grid({-X, 0, -Z}, {2 * X, 0, 0}, {0, 0, 2 * Z}, {0, 1, 0}, nx, nz, interior ? 0.01f : 0.06f, interior ? 30.f : 150.f, seed, interior ? 10.f : 60.f);
    B.addNode(B.endMesh());
  }
  if(interior) {
    B.beginMesh(mat(1)); B.box({0, 2.f, 0}, {X, 2.f, Z}, 0.f, true, 6.f); B.addNode(B.endMesh());  // room shell, facing inward
  } else {
    // buildings along both sides of the street: facade with window frames
    int nb = std::max(2, int(24 * scale + 0.5f));
    for(int b = 0; b < nb; b++) {
      const float wbx = -X + 6.f + (b / 2) * (2 * X - 12.f) / std::max(1, nb / 2 - 1 + (nb / 2 == 1)), wbz = (b & 1) ? 14.f : -14.f;
      float hw = B.rng.range(3.5f, 4.8f), hh = B.rng.range(6.f, 11.f), hd = B.rng.range(4.f, 6.f);
      // stress variant: every building (shell, window frames, glass) is modelled around its own origin and turned by a random angle, so the large
      // facade triangles and the thousands of frame boxes are no longer axis aligned
      const float bx = stress ? 0.f : wbx, bz = stress ? 0.f : wbz;
      const M4 bm = stress ? translate({wbx, 0, wbz}) * rotateY(B.rng.range(-0.45f, 0.45f)) : translate({0, 0, 0});
      B.beginMesh(mat(2 + b)); B.box({bx, hh, bz}, {hw, hh, hd}, 0.f, false, 4.f); B.addNode(B.endMesh(), bm);
      B.beginMesh(mat(7 + b));
      int floors = std::max(1, int(10 * s)), cols = std::max(1, int(12 * s));
      float face = (b & 1) ? bz - hd : bz + hd, dir = (b & 1) ? -1.f : 1.f;
      for(int f = 0; f < floors; f++)
        for(int c = 0; c < cols; c++) {
          float wx = bx - hw + (c + 0.5f) * 2 * hw / cols, wy = (f + 0.6f) * 2 * hh / floors, ww = 0.7f * hw / cols, wh = 0.6f * hh / floors;
          B.box({wx - ww, wy, face + dir * 0.06f}, {0.04f, wh, 0.06f});
          B.box({wx + ww, wy, face + dir * 0.06f}, {0.04f, wh, 0.06f});
          B.box({wx, wy - wh, face + dir * 0.06f}, {ww, 0.04f, 0.08f});
          B.box({wx, wy + wh, face + dir * 0.06f}, {ww, 0.04f, 0.06f});
        }
      B.addNode(B.endMesh(), bm);
      B.beginMesh(P.glass);
      for(int f = 0; f < floors; f++)
        for(int c = 0; c < cols; c++) {
          float wx = bx - hw + (c + 0.5f) * 2 * hw / cols, wy = (f + 0.6f) * 2 * hh / floors, ww = 0.7f * hw / cols, wh = 0.6f * hh / floors, z = face + dir * 0.02f;
          B.quad({wx - ww, wy - wh, z}, {wx + ww, wy - wh, z}, {wx + ww, wy + wh, z}, {wx - ww, wy + wh, z}, {0, 0, dir});
        }
      B.addNode(B.endMesh(), bm);
    }
    // trees: 4 prim meshes (trunk + alpha-masked leaf quads), instanced — ~10 % of all triangles
    int leafQuads = std::max(8, int((stress ? 6000 : 3000) * scale));
    int treeMesh[4], trunkMesh;
    B.beginMesh(mat(3)); B.cylinder({0, 0, 0}, 0.22f, 3.2f, std::max(5, int(24 * s)), std::max(2, int(20 * s))); trunkMesh = B.endMesh();
    for(int t = 0; t < 4; t++) {
      B.beginMesh(P.leaf);
      for(int q = 0; q < leafQuads; q++) {
        float th = B.rng.range(0, 3.14159f), ph = B.rng.range(0, 6.28318f), r = 1.7f * std::cbrt(B.rng.uni());
        V3 c{r * std::sin(th) * std::cos(ph), 4.2f + r * std::cos(th) * 0.8f, r * std::sin(th) * std::sin(ph)};
        V3 n = normalize(V3{B.rng.range(-1, 1), B.rng.range(-0.3f, 1), B.rng.range(-1, 1)});
        V3 t1 = normalize(cross(n, V3{0.3f, 1, 0.2f})), t2 = cross(n, t1);
        float sz = B.rng.range(0.12f, 0.22f);
        static const int tess = getenv("RESTIR_SCENE_TESS_LEAVES") ? atoi(getenv("RESTIR_SCENE_TESS_LEAVES")) : 0;
        if(tess > 1) B.quadTessellated(c - t1 * sz - t2 * sz, c + t1 * sz - t2 * sz, c + t1 * sz + t2 * sz, c - t1 * sz + t2 * sz, n, tess, B.g.textures[size_t(B.g.materials[size_t(P.leaf)].baseColorTexture)]);
        else B.quad(c - t1 * sz - t2 * sz, c + t1 * sz - t2 * sz, c + t1 * sz + t2 * sz, c - t1 * sz + t2 * sz, n);
      }
      treeMesh[t] = B.endMesh();
    }
    int nt = std::max(2, int((stress ? 80 : 40) * scale + 0.5f));
    for(int t = 0; t < nt; t++) {
      M4 m = translate({-X + 5.f + t * (2 * X - 10.f) / nt, 0, (t & 1) ? 7.5f : -7.5f}) * rotateY(B.rng.range(0, 6.28f)) * scaleM({1, B.rng.range(0.85f, 1.2f), 1});
      B.addNode(trunkMesh, m); B.addNode(treeMesh[t & 3], m);
    }
  }
  // lamps: emissive meshes (street lamps outside, ~30 ceiling lamps inside: config 5)
  {
    int nl = interior ? std::max(2, int(30 * std::min(1.f, scale * 4))) : std::max(2, int(40 * std::min(1.f, scale * 4)));
    for(int l = 0; l < nl; l++) {
      float lx = -X + 3.f + (l + 0.5f) * (2 * X - 6.f) / nl, lz = interior ? ((l % 3) - 1) * 3.2f : ((l & 1) ? 5.2f : -5.2f), ly = interior ? 3.6f : 4.4f;
      B.beginMesh(P.lamp); B.sphere({lx, ly, lz}, {0.16f, 0.12f, 0.16f}, interior ? 8 : 6, 4); B.addNode(B.endMesh());
      if(!interior) { B.beginMesh(P.metal); B.cylinder({lx, 0, lz}, 0.05f, 4.3f, 8, 2); B.addNode(B.endMesh()); }
      if(!interior && stress && l + 1 < nl) {   // stress variant: thin diagonal wires from lamp to lamp across the street
        const float nx2 = -X + 3.f + (l + 1.5f) * (2 * X - 6.f) / nl, nz2 = -lz;
        const float dx = nx2 - lx, dz = nz2 - lz, len = std::sqrt(dx * dx + dz * dz);
        B.beginMesh(P.metal); B.box({0, 0, 0}, {0.5f * len, 0.008f, 0.008f});
        B.addNode(B.endMesh(), translate({0.5f * (lx + nx2), 4.25f, 0.5f * (lz + nz2)}) * rotateY(-std::atan2(dz, dx)));
      }
    }
  }
  // props (chairs / bikes / crockery stand-ins): high-resolution blobs, the bulk of the triangle count
  {
    int np = std::max(2, int((interior ? 220 : 415) * s)), nu = std::max(6, int(50 * s)), nv = std::max(4, int(40 * s));
    int meshes[8];
    for(int k = 0; k < 8; k++) { B.beginMesh(k == 7 ? P.metal : mat(11 + k)); B.sphere({0, 0, 0}, {1.f, 0.7f + 0.1f * k, 1.f}, nu, nv, 0.35f, seed + 31u * k); meshes[k] = B.endMesh(); }
    for(int p = 0; p < np; p++) {
      float r = B.rng.range(0.15f, interior ? 0.45f : 0.7f);
      V3 pos{B.rng.range(-X + 1.f, X - 1.f), r * 0.8f, interior ? B.rng.range(-Z + 1.f, Z - 1.f) : B.rng.range(-6.5f, 6.5f)};
      float sgn = (p % 11 == 0) ? -1.f : 1.f;  // a few mirrored instances (negative determinant)
      B.addNode(meshes[p & 7], translate(pos) * rotateY(B.rng.range(0, 6.28f)) * scaleM({r * sgn, r, r}));
    }
  }
  GltfCamera cam;
  if(interior) { cam.eye = {-8.5f, 1.7f, 4.5f}; cam.center = {2.f, 1.2f, -1.f}; cam.yfovDeg = 65.f; }
  else { cam.eye = {-52.f, 2.4f, 1.5f}; cam.center = {10.f, 4.5f, -1.f}; cam.yfovDeg = 60.f; }
  B.g.cameras.push_back(cam);
  return std::move(B.g);
}

// ---- config 4, "real footprint": the exterior street with the memory behaviour of the asset it stands in for ------------------------------------------------
// The round-1 scene above (kept as the `lite` variant) holds 60 MB of 512^2 textures and ONE 256^2 cut-out: its whole working set sits in the 256 MiB Infinity
// Cache.  The reference uploads every glTF image at full size as BGRA8 without mips (src/scene.cpp:554-646); Bistro Exterior is ~130 materials with 2k
// base-colour / normal / specular sets and dozens of foliage cards.  This variant follows those rules at scale 1:
//   * 128 textured opaque materials, each a 2048^2 base colour; every second one a 2048^2 normal map, every third one a 1024^2 metallic-roughness map
//     (3.3 GB of texels), mapped about once per object (no dense tiling: minified, incoherent fetches — the reference never generates mips)
//   * 16 distinct 1024^2 cut-out textures (single lobed leaves and sprays of small leaves, 18-40 % coverage) spread over the tree crowns and the facade ivy
//   * long thin triangles in every orientation: balcony rails and balusters, sagging cables across the street, striped awnings cut into strips, chairs
//   * the same 2.8 M instanced triangles and the same camera
// Texture sizes follow sqrt(scale) (128^2 at the 0.01 scale of the quick parity tests).
int pow2Floor(int v) { int p = 1; while(p * 2 <= v) p *= 2; return p; }
GltfScene makeBistroExteriorReal(float scale, uint32_t seed)
{
  Builder B(seed);
  std::vector<TexJob> jobs;
  const float s = std::sqrt(scale), one[3] = {1, 1, 1};
  // experiment switch (round 5, profiles/r05_real_reask.txt): RESTIR_SCENE_TEXSIZE=<n> builds the same geometry and materials with n x n textures — what of the
  // real scene's cost is the texel footprint and what the triangle shapes
  const int texSize = getenv("RESTIR_SCENE_TEXSIZE") ? std::max(32, pow2Floor(atoi(getenv("RESTIR_SCENE_TEXSIZE")))) : std::max(64, pow2Floor(int(2048.f * s + 0.5f)));
  const float X = 60.f, Z = 40.f;
  // ---- materials ----------------------------------------------------------------------------------------------------------------------------------
  // pattern kinds by use: 0..15 ground (cobbles / mottled), 16..39 walls, 40..51 frames, 52..63 balconies, 64..71 awnings, 72..119 props, 120..123 bark, 124..127 furniture
  const int NMAT = 128;
  std::vector<int> M(NMAT);
  for(int i = 0; i < NMAT; i++) {
    float tint[3] = {B.rng.range(0.35f, 0.95f), B.rng.range(0.35f, 0.9f), B.rng.range(0.3f, 0.85f)};
    const int kind = i < 16 ? ((i & 1) ? 8 : 0) : i < 40 ? ((i % 3 == 0) ? 2 : 7) : i < 64 ? 0 : i < 72 ? 9 : (i % 3);
    GltfMaterial m = diffuse(1, 1, 1, B.rng.range(0.2f, 1.f), B.rng.uni() < 0.2f ? 1.f : 0.f);
    m.baseColorTexture = declareTexture(B, jobs, texSize, kind, seed + 11u * i, tint);
    if(i % 2 == 0) m.normalTexture = declareTexture(B, jobs, texSize, 4, seed + 101u * i, one);
    if(i % 3 == 0) m.metallicRoughnessTexture = declareTexture(B, jobs, std::max(32, texSize / 2), 5, seed + 1001u * i, one);
    M[size_t(i)] = B.addMaterial(m);
  }
  int leafMat[16];
  for(int i = 0; i < 16; i++) {
    float tint[3] = {B.rng.range(0.15f, 0.45f), B.rng.range(0.4f, 0.7f), B.rng.range(0.1f, 0.3f)};
    GltfMaterial leaf = diffuse(1, 1, 1, 0.8f);
    leaf.baseColorTexture = declareTexture(B, jobs, std::max(32, texSize / 2), 6, seed + 77u + 13u * i, tint, i);
    leaf.alphaMode = RT_ALPHA_MASK; leaf.alphaCutoff = 0.5f; leaf.doubleSided = 1;
    if(getenv("RESTIR_DEBUG_OPAQUE_LEAVES")) leaf.alphaMode = RT_ALPHA_OPAQUE;
    leafMat[i] = B.addMaterial(leaf);
  }
  const int glass = B.addMaterial(diffuse(0.6f, 0.7f, 0.75f, 0.05f, 1.f));
  const int metal = B.addMaterial(diffuse(0.9f, 0.85f, 0.7f, 0.25f, 1.f));
  const int lamp = B.addMaterial(emissive(60.f, 48.f, 30.f));
  // ---- ground: 4 x 4 patches, each its own material, mapped 3 x over a 30 x 20 m patch -----------------------------------------------------------------
  {
    const int n = std::max(2, int(125 * s));
    for(int pz = 0; pz < 4; pz++)
      for(int px = 0; px < 4; px++) {
        B.beginMesh(M[size_t(pz * 4 + px)]);
        // (one height field over the whole street: the patches meet without gaps because amp / freq / seed are global and the grid evaluates H(u, v) per patch —
        // patches use their own seed and a small amplitude, the seams are below a centimetre and hidden under the kerb beams)
        B.grid({-X + px * 2 * X / 4, 0, -Z + pz * 2 * Z / 4}, {2 * X / 4, 0, 0}, {0, 0, 2 * Z / 4}, {0, 1, 0}, n, n, 0.02f, 40.f, seed + 17u * (pz * 4 + px), 3.f);
        B.addNode(B.endMesh());
      }
    B.beginMesh(metal);
    for(int k = 1; k < 4; k++) { B.beam({-X, 0.01f, -Z + k * 2 * Z / 4}, {X, 0.01f, -Z + k * 2 * Z / 4}, 0.03f); B.beam({-X + k * 2 * X / 4, 0.01f, -Z}, {-X + k * 2 * X / 4, 0.01f, Z}, 0.03f); }
    B.addNode(B.endMesh());
  }
  // ---- buildings: shell, window frames, glass, balconies with rails and balusters, awnings --------------------------------------------------------------
  const int nb = std::max(2, int(24 * scale + 0.5f));
  for(int b = 0; b < nb; b++) {
    const float wbx = -X + 6.f + (b / 2) * (2 * X - 12.f) / std::max(1, nb / 2 - 1 + (nb / 2 == 1)), wbz = (b & 1) ? 14.f : -14.f;
    const float hw = B.rng.range(3.5f, 4.8f), hh = B.rng.range(6.f, 11.f), hd = B.rng.range(4.f, 6.f);
    // every second building is turned a little: its frames, rails and balusters are not axis aligned
    const M4 bm = translate({wbx, 0, wbz}) * rotateY((b % 2 == 0) ? B.rng.range(-0.2f, 0.2f) : 0.f);
    const int floors = std::max(1, int(10 * s)), cols = std::max(1, int(12 * s));
    const float face = (b & 1) ? -hd : hd, dir = (b & 1) ? -1.f : 1.f;
    B.beginMesh(M[size_t(16 + b % 24)]); B.box({0, hh, 0}, {hw, hh, hd}, 0.f, false, 2.f); B.addNode(B.endMesh(), bm);
    B.beginMesh(M[size_t(40 + b % 12)]);
    for(int f = 0; f < floors; f++)
      for(int c = 0; c < cols; c++) {
        const float wx = -hw + (c + 0.5f) * 2 * hw / cols, wy = (f + 0.6f) * 2 * hh / floors, ww = 0.7f * hw / cols, wh = 0.6f * hh / floors;
        B.box({wx - ww, wy, face + dir * 0.06f}, {0.04f, wh, 0.06f});
        B.box({wx + ww, wy, face + dir * 0.06f}, {0.04f, wh, 0.06f});
        B.box({wx, wy - wh, face + dir * 0.06f}, {ww, 0.04f, 0.08f});
        B.box({wx, wy + wh, face + dir * 0.06f}, {ww, 0.04f, 0.06f});
      }
    B.addNode(B.endMesh(), bm);
    B.beginMesh(glass);
    for(int f = 0; f < floors; f++)
      for(int c = 0; c < cols; c++) {
        const float wx = -hw + (c + 0.5f) * 2 * hw / cols, wy = (f + 0.6f) * 2 * hh / floors, ww = 0.7f * hw / cols, wh = 0.6f * hh / floors, z = face + dir * 0.02f;
        B.quad({wx - ww, wy - wh, z}, {wx + ww, wy - wh, z}, {wx + ww, wy + wh, z}, {wx - ww, wy + wh, z}, {0, 0, dir});
      }
    B.addNode(B.endMesh(), bm);
    // balconies: a slab, three full-width rails and a baluster every 12 cm, on every floor above the ground floor
    B.beginMesh(M[size_t(52 + b % 12)]);
    for(int f = 1; f < floors; f++) {
      const float y = (f + 0.6f) * 2 * hh / floors - 0.6f * hh / floors - 0.05f, z0 = face + dir * 0.1f, z1 = face + dir * 0.75f;
      B.box({0, y, 0.5f * (z0 + z1)}, {hw * 0.96f, 0.03f, 0.5f * std::fabs(z1 - z0)});
      for(int k = 0; k < 3; k++) B.beam({-hw * 0.96f, y + 0.3f + 0.3f * k, z1}, {hw * 0.96f, y + 0.3f + 0.3f * k, z1}, 0.012f);
      const int nbal = std::max(2, int(2 * hw * 0.96f / 0.12f * s));
      for(int k = 0; k <= nbal; k++) { const float x = -hw * 0.96f + k * 2 * hw * 0.96f / nbal; B.beam({x, y, z1}, {x, y + 0.9f, z1}, 0.007f); }
    }
    B.addNode(B.endMesh(), bm);
    // awnings over the ground floor: slanted sheets cut into strips along their width, on two thin struts each
    B.beginMesh(M[size_t(64 + b % 8)]);
    for(int a = 0; a < 3; a++) {
      const float ax0 = -hw + (a + 0.1f) * 2 * hw / 3, aw = 0.8f * 2 * hw / 3, ay = 3.1f, out = 1.6f, drop = 0.7f;
      B.strips({ax0, ay, face + dir * 0.05f}, {aw, 0, 0}, {0, -drop, dir * out}, normalize(V3{0, out, dir * drop}), std::max(2, int(60 * s)));
    }
    B.addNode(B.endMesh(), bm);
    B.beginMesh(metal);
    for(int a = 0; a < 3; a++) {
      const float ax0 = -hw + (a + 0.1f) * 2 * hw / 3, aw = 0.8f * 2 * hw / 3;
      for(int e = 0; e < 2; e++) B.beam({ax0 + e * aw, 3.1f - 0.7f, face + dir * 1.65f}, {ax0 + e * aw, 2.2f, face + dir * 0.05f}, 0.01f);
    }
    B.addNode(B.endMesh(), bm);
    // ivy on every third facade: cut-out cards a few centimetres off the wall
    if(b % 3 == 0) {
      const int cards = std::max(4, int(1500 * scale));
      for(int part = 0; part < 2; part++) {
        B.beginMesh(leafMat[(b + 8 * part + 3) % 16]);
        for(int q = 0; q < cards / 2; q++) {
          const float x = B.rng.range(-hw, hw), y = B.rng.range(0.2f, hh * 1.2f) * std::sqrt(B.rng.uni()), z = face + dir * B.rng.range(0.1f, 0.3f);
          V3 n = normalize(V3{B.rng.range(-0.5f, 0.5f), B.rng.range(-0.2f, 0.6f), dir});
          V3 t1 = normalize(cross(n, V3{0.3f, 1, 0.2f})), t2 = cross(n, t1);
          const float sz = B.rng.range(0.1f, 0.2f);
          V3 c{x, y, z};
          B.quad(c - t1 * sz - t2 * sz, c + t1 * sz - t2 * sz, c + t1 * sz + t2 * sz, c - t1 * sz + t2 * sz, n);
        }
        B.addNode(B.endMesh(), bm);
      }
    }
  }
  // ---- trees: 4 crown shapes x 4 card textures each (16 cut-out materials), instanced along the street ------------------------------------------------
  {
    const int leafQuads = std::max(8, int(3000 * scale));
    int trunkMesh[4], crownMesh[4][4];
    for(int t = 0; t < 4; t++) {
      B.beginMesh(M[size_t(120 + t)]); B.cylinder({0, 0, 0}, 0.22f, 3.2f, std::max(5, int(24 * s)), std::max(2, int(20 * s))); trunkMesh[t] = B.endMesh();
      for(int k = 0; k < 4; k++) {
        B.beginMesh(leafMat[t * 4 + k]);
        for(int q = k; q < leafQuads; q += 4) {
          float th = B.rng.range(0, 3.14159f), ph = B.rng.range(0, 6.28318f), r = 1.7f * std::cbrt(B.rng.uni());
          V3 c{r * std::sin(th) * std::cos(ph), 4.2f + r * std::cos(th) * 0.8f, r * std::sin(th) * std::sin(ph)};
          V3 n = normalize(V3{B.rng.range(-1, 1), B.rng.range(-0.3f, 1), B.rng.range(-1, 1)});
          V3 t1 = normalize(cross(n, V3{0.3f, 1, 0.2f})), t2 = cross(n, t1);
          float sz = B.rng.range(0.12f, 0.22f);
          B.quad(c - t1 * sz - t2 * sz, c + t1 * sz - t2 * sz, c + t1 * sz + t2 * sz, c - t1 * sz + t2 * sz, n);
        }
        crownMesh[t][k] = B.endMesh();
      }
    }
    const int nt = std::max(4, int(40 * scale + 0.5f));   // (at least one tree of every crown shape: all 16 card textures are referenced at any scale)
    for(int t = 0; t < nt; t++) {
      M4 m = translate({-X + 5.f + t * (2 * X - 10.f) / nt, 0, (t & 1) ? 7.5f : -7.5f}) * rotateY(B.rng.range(0, 6.28f)) * scaleM({1, B.rng.range(0.85f, 1.2f), 1});
      B.addNode(trunkMesh[t & 3], m);
      for(int k = 0; k < 4; k++) B.addNode(crownMesh[t & 3][k], m);
    }
  }
  // ---- lamps, posts, and cables sagging from post to post across the street ------------------------------------------------------------------------------
  {
    const int nl = std::max(2, int(40 * std::min(1.f, scale * 4)));
    for(int l = 0; l < nl; l++) {
      const float lx = -X + 3.f + (l + 0.5f) * (2 * X - 6.f) / nl, lz = (l & 1) ? 5.2f : -5.2f, ly = 4.4f;
      B.beginMesh(lamp); B.sphere({lx, ly, lz}, {0.16f, 0.12f, 0.16f}, 6, 4); B.addNode(B.endMesh());
      B.beginMesh(metal); B.cylinder({lx, 0, lz}, 0.05f, 4.3f, 8, 2); B.addNode(B.endMesh());
      if(l + 1 < nl) {
        const float nx2 = -X + 3.f + (l + 1.5f) * (2 * X - 6.f) / nl, nz2 = -lz;
        B.beginMesh(metal);
        const int seg = std::max(2, int(16 * s));
        for(int c = 0; c < 3; c++) {           // three strands with different sag
          const float sag = 0.25f + 0.2f * c, y0 = 4.25f - 0.1f * c;
          for(int k = 0; k < seg; k++) {
            const float a0 = float(k) / seg, a1 = float(k + 1) / seg;
            auto P = [&](float a) { return V3{lx + (nx2 - lx) * a, y0 - sag * 4.f * a * (1.f - a), lz + (nz2 - lz) * a}; };
            B.beam(P(a0), P(a1), 0.006f);
          }
        }
        B.addNode(B.endMesh());
      }
    }
  }
  // ---- furniture: chairs (thin legs, back slats) and tables on the pavement, instanced ---------------------------------------------------------------------
  {
    int chairMesh[4];
    for(int k = 0; k < 4; k++) {
      B.beginMesh(M[size_t(124 + k)]);
      B.box({0, 0.45f, 0}, {0.2f, 0.015f, 0.2f});
      for(int lx = -1; lx <= 1; lx += 2) for(int lz = -1; lz <= 1; lz += 2) B.beam({0.18f * lx, 0, 0.18f * lz}, {0.17f * lx, 0.45f, 0.17f * lz}, 0.01f);
      for(int lx = -1; lx <= 1; lx += 2) B.beam({0.18f * lx, 0.45f, -0.19f}, {0.19f * lx, 0.95f, -0.24f}, 0.01f);
      for(int sl = 0; sl < 4 + k; sl++) { const float y = 0.55f + 0.4f * sl / (4 + k); B.beam({-0.19f, y, -0.19f - 0.05f * (y - 0.45f) / 0.5f}, {0.19f, y, -0.19f - 0.05f * (y - 0.45f) / 0.5f}, 0.008f); }
      chairMesh[k] = B.endMesh();
    }
    const int nc = std::max(2, int(600 * scale));
    for(int c = 0; c < nc; c++) {
      V3 pos{B.rng.range(-X + 2.f, X - 2.f), 0.f, (c & 1) ? B.rng.range(8.3f, 9.4f) : B.rng.range(-9.4f, -8.3f)};
      B.addNode(chairMesh[c & 3], translate(pos) * rotateY(B.rng.range(0, 6.28f)));
    }
  }
  // ---- props: 48 distinct meshes, one material each -------------------------------------------------------------------------------------------------------
  {
    const int np = std::max(2, int(425 * s)), nu = std::max(6, int(48 * s)), nv = std::max(4, int(38 * s));
    int meshes[48];
    for(int k = 0; k < 48; k++) { B.beginMesh(M[size_t(72 + k)]); B.sphere({0, 0, 0}, {1.f, 0.6f + 0.02f * k, 1.f}, nu, nv, 0.35f, seed + 31u * k); meshes[k] = B.endMesh(); }
    for(int p = 0; p < np; p++) {
      float r = B.rng.range(0.15f, 0.7f);
      V3 pos{B.rng.range(-X + 1.f, X - 1.f), r * 0.8f, B.rng.range(-6.5f, 6.5f)};
      float sgn = (p % 11 == 0) ? -1.f : 1.f;  // a few mirrored instances (negative determinant)
      B.addNode(meshes[p % 48], translate(pos) * rotateY(B.rng.range(0, 6.28f)) * scaleM({r * sgn, r, r}));
    }
  }
  fillTextures(B, jobs);
  GltfCamera cam; cam.eye = {-52.f, 2.4f, 1.5f}; cam.center = {10.f, 4.5f, -1.f}; cam.yfovDeg = 60.f;
  B.g.cameras.push_back(cam);
  return std::move(B.g);
}

}  // namespace

GltfScene makeProceduralScene(ProcScene kind, float scale, uint32_t seed)
{
  scale = std::min(1.f, std::max(1e-4f, scale));
  switch(kind) {
    case PROC_CORNELL: return makeCornell();
    case PROC_HELMET: return makeHelmet(scale, seed);
    case PROC_SPONZA: return makeSponza(scale, seed);
    case PROC_BISTRO_EXT: return makeBistro(false, scale, seed);
    case PROC_BISTRO_INT: return makeBistro(true, scale, seed);
    case PROC_BISTRO_EXT_REAL: return makeBistroExteriorReal(scale, seed);
    case PROC_SPONZA_1K: return makeSponza(scale, seed, std::max(64, pow2Floor(int(1024.f * std::sqrt(scale) + 0.5f))));
  }
  return makeCornell();
}

}  // namespace rth
