// orc_shading.h — CPU oracle restatement of the reference's shader library (TEST INFRASTRUCTURE).
//
// Function-by-function restatement of (citations are /root/reference/shaders/<file>:<lines>):
//   random.glsl, common.glsl, globals.glsl, compress.glsl, reservoir.glsl, pbr_metallicworkflow.glsl,
//   gltf_material.glsl, shade_state.glsl (GetState), env_sampling.glsl, pathtrace.glsl.
// GLSL leaves the value of uninitialised variables / `out` parameters undefined; the oracle defines them as 0
// (DESIGN.md §Deviations #3).  Transcendentals come from include/rt_detmath.h.
#pragma once
#include <cstdio>
#include <cmath>
#include "orc_scene.h"
#include "orc_sky.h"

namespace orc {

// ---------------------------------------------------------------------------------------------- random.glsl
inline uint32_t tea(uint32_t val0, uint32_t val1)  // random.glsl:34-48
{
  uint32_t v0 = val0, v1 = val1, s0 = 0;
  for(uint32_t n = 0; n < 16; n++) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}
inline uint32_t pcg(uint32_t& state)  // random.glsl:59-65
{
  uint32_t prev = state * 747796405u + 2891336453u;
  uint32_t word = ((prev >> ((prev >> 28u) + 4u)) ^ prev) * 277803737u;
  state = prev;
  return (word >> 22u) ^ word;
}
inline float rnd(uint32_t& seed)  // rand(), random.glsl:98-102
{
  uint32_t r = pcg(seed);
  return rt_u2f(0x3f800000u | (r >> 9)) - 1.0f;
}

// ---------------------------------------------------------------------------------------------- globals.glsl
const float M_PI_F = 3.14159265358979323846f;         // globals.glsl:44
const float M_1_OVER_PI_F = 0.318309886183790671538f;  // globals.glsl:48
const float InvalidPdf = -1.0f;                        // common.glsl:24
const float Pi = M_PI_F;                               // pbr_metallicworkflow.glsl:8
const float PiInv = 1.0f / Pi;                         // pbr_metallicworkflow.glsl:9

struct Ray { vec3 origin, direction; };
struct Material {  // globals.glsl:77-85
  vec3 albedo{0, 0, 0}, emission{0, 0, 0};
  float metallic = 0, ior = 0, roughness = 0, transmission = 0;
};
struct State {  // globals.glsl:87-104
  int depth = 0;
  float eta = 0;
  vec3 position{0, 0, 0}, normal{0, 0, 0}, tangent{0, 0, 0}, bitangent{0, 0, 0}, ffnormal{0, 0, 0};
  vec2 texCoord{0, 0};
  bool isEmitter = false;
  uint32_t matID = 0;
  Material mat;
  float area = 0;
};

// ---------------------------------------------------------------------------------------------- common.glsl
inline vec2 GetSphericalUv(vec3 v)  // common.glsl:63-70
{
  float gamma = rt_asin(-v.y);
  float theta = rt_atan2(v.z, v.x);
  return V2(theta * M_1_OVER_PI_F * 0.5f, gamma * M_1_OVER_PI_F) + 0.5f;
}
inline void CreateCoordinateSystem(vec3 N, vec3& Nt, vec3& Nb)  // common.glsl:75-81
{
  Nt = normalize((rt_abs(N.z) > 0.99999f) ? V3(-N.x * N.y, 1.0f - N.y * N.y, -N.y * N.z) : V3(-N.x * N.z, -N.y * N.z, 1.0f - N.z * N.z));
  Nb = cross(Nt, N);
}
inline vec3 OffsetRay(vec3 p, vec3 n)  // common.glsl:89-105
{
  const float intScale = 256.0f, floatScale = 1.0f / 65536.0f, origin = 1.0f / 32.0f;
  int32_t ofx = rt_ftoi(intScale * n.x), ofy = rt_ftoi(intScale * n.y), ofz = rt_ftoi(intScale * n.z);
  auto nudge = [](float p, int32_t of) { return rt_u2f(rt_f2u(p) + uint32_t((p < 0) ? -of : of)); };
  vec3 p_i = V3(nudge(p.x, ofx), nudge(p.y, ofy), nudge(p.z, ofz));
  return V3(rt_abs(p.x) < origin ? p.x + floatScale * n.x : p_i.x, rt_abs(p.y) < origin ? p.y + floatScale * n.y : p_i.y,
            rt_abs(p.z) < origin ? p.z + floatScale * n.z : p_i.z);
}
inline uint32_t hash8bit(uint32_t a) { return (a ^ (a >> 8)) << 24; }  // common.glsl:141-143
inline vec2 toConcentricDisk(vec2 r)  // common.glsl:170-174
{
  float rx = rt_sqrt(r.x);
  float theta = r.y * 2.0f * M_PI_F;
  return V2(rt_cos(theta), rt_sin(theta)) * rx;
}
inline float powerHeuristic(float f, float g) { float f2 = f * f; return f2 / (f2 + g * g); }  // common.glsl:176-179
inline bool inBound(ivec2 p, ivec2 pMin, ivec2 pMax) { return p.x >= pMin.x && p.x < pMax.x && p.y >= pMin.y && p.y < pMax.y; }  // :185-187
inline bool inBound(ivec2 p, ivec2 b) { return inBound(p, ivec2{0, 0}, b); }
inline vec3 HDRToLDR(vec3 c) { return c / (c + 1.0f); }    // common.glsl:194-196
inline vec3 LDRToHDR(vec3 c) { return c / (1.01f - c); }   // common.glsl:198-200

// -------------------------------------------------------------------------------------------- compress.glsl
inline float roundHalfAway(float x) { float r = truncf(x); if(x - r >= 0.5f) r += 1.0f; return r; }  // std::round for x >= 0
inline uint32_t packUnorm4x8(vec4 v)  // compress.glsl:58-72 (C++ branch) == GLSL packUnorm4x8
{
  uint32_t r = uint32_t(roundHalfAway(rt_clamp(v.x, 0.0f, 1.0f) * 255.0f));
  uint32_t g = uint32_t(roundHalfAway(rt_clamp(v.y, 0.0f, 1.0f) * 255.0f));
  uint32_t b = uint32_t(roundHalfAway(rt_clamp(v.z, 0.0f, 1.0f) * 255.0f));
  uint32_t a = uint32_t(roundHalfAway(rt_clamp(v.w, 0.0f, 1.0f) * 255.0f));
  return r | (g << 8) | (b << 16) | (a << 24);
}
inline vec4 unpackUnorm4x8(uint32_t p)
{
  return V4(float(p & 0xffu) / 255.0f, float((p >> 8) & 0xffu) / 255.0f, float((p >> 16) & 0xffu) / 255.0f, float(p >> 24) / 255.0f);
}
inline uint32_t compress_unit_vec(vec3 nv)  // compress.glsl:111-139
{
  if((nv.x < 3.402823466e+38f) && !rt_isinf(nv.x)) {
    const float d = 32767.0f / ((rt_abs(nv.x) + rt_abs(nv.y)) + rt_abs(nv.z));
    int x = rt_ftoi(rintf(nv.x * d));  // roundEven
    int y = rt_ftoi(rintf(nv.y * d));
    if(nv.z < 0.0f) {
      const int maskx = x >> 31, masky = y >> 31;
      const int tmp = 32767 + maskx + masky;
      const int tmpx = x;
      x = (tmp - (y ^ masky)) ^ maskx;
      y = (tmp - (tmpx ^ maskx)) ^ masky;
    }
    uint32_t packed = (uint32_t(y + 32767) << 16) | (uint32_t(x + 32767) & 0xffffu);
    if(packed == ~0u) return ~0x1u;
    return packed;
  }
  return ~0u;
}
inline float short_to_floatm11(int v)  // compress.glsl:142-146
{
  return (v >= 0) ? (rt_u2f(0x3F800000u | (uint32_t(v) << 8)) - 1.0f) : (rt_u2f((0x80000000u | 0x3F800000u) | (uint32_t(-v) << 8)) + 1.0f);
}
inline vec3 decompress_unit_vec(uint32_t packed)  // compress.glsl:149-180
{
  if(packed != ~0u) {
    int x = int(packed & 0xFFFFu) - 32767;
    int y = int(packed >> 16) - 32767;
    const int maskx = x >> 31, masky = y >> 31;
    const int tmp0 = 32767 + maskx + masky;
    const int ymask = y ^ masky;
    const int tmp1 = tmp0 - (x ^ maskx);
    const int z = tmp1 - ymask;
    float zf;
    if(z < 0) {
      x = (tmp0 - ymask) ^ maskx;
      y = tmp1 ^ masky;
      zf = rt_u2f((0x80000000u | 0x3F800000u) | (uint32_t(-z) << 8)) + 1.0f;
    } else {
      zf = rt_u2f(0x3F800000u | (uint32_t(z) << 8)) - 1.0f;
    }
    return normalize(V3(short_to_floatm11(x), short_to_floatm11(y), zf));
  }
  return V3(3.402823466e+38f);
}

// ------------------------------------------------------------------------------------------- reservoir.glsl
inline rt_light_sample zeroLightSample() { rt_light_sample s; memset(&s, 0, sizeof(s)); return s; }
inline vec3 toV(rt_vec3 v) { return V3(v.x, v.y, v.z); }
inline rt_vec3 toR(vec3 v) { return rt_vec3{v.x, v.y, v.z}; }
inline float resvToScalar(vec3 x) { return luminance(x); }                                              // :7-9
inline void resvReset(rt_direct_reservoir& r) { r.num = 0; r.weight = 0; }                              // :11-14
inline void resvReset(rt_indirect_reservoir& r) { r.num = 0; r.weight = 0; r.bigW = 0; }                // :16-20
inline bool resvInvalid(const rt_direct_reservoir& r) { return rt_isnan(r.weight) || r.weight < 0.0f; }   // :26-28
inline bool resvInvalid(const rt_indirect_reservoir& r) { return rt_isnan(r.weight) || r.weight < 0.0f; } // :30-32
inline void resvCheckValidity(rt_direct_reservoir& r) { if(resvInvalid(r)) resvReset(r); }              // :34-38
inline void resvCheckValidity(rt_indirect_reservoir& r) { if(resvInvalid(r)) resvReset(r); }            // :40-44
inline bool resvUpdate(rt_direct_reservoir& r, const rt_light_sample& s, float w, float rr)              // :46-52
{
  r.weight += w; r.num += 1;
  if(rr * r.weight < w) { r.lightSample = s; return true; }
  return false;
}
inline void resvUpdate(rt_indirect_reservoir& r, const rt_gi_sample& s, float w, float rr)  // :54-60
{
  r.weight += w; r.num += 1;
  if(rr * r.weight < w) r.giSample = s;
}
inline bool resvMerge(rt_direct_reservoir& r, const rt_direct_reservoir& rhs, float rr)  // :68-74
{
  r.weight += rhs.weight; r.num += rhs.num;
  if(rr * r.weight < rhs.weight) { r.lightSample = rhs.lightSample; return true; }
  return false;
}
inline void resvClamp(rt_direct_reservoir& r, int clamp)  // :116-121
{
  if(r.num > uint32_t(clamp)) { r.weight *= float(clamp) / float(r.num); r.num = uint32_t(clamp); }
}
inline void resvClamp(rt_indirect_reservoir& r, int clamp)  // :123-128
{
  if(r.num > uint32_t(clamp)) { r.weight *= float(clamp) / float(r.num); r.num = uint32_t(clamp); }
}

// --------------------------------------------------------------------------------- pbr_metallicworkflow.glsl
inline mat3 localRefMatrix(vec3 n)  // :11-16
{
  vec3 t = (rt_abs(n.y) > 0.9999f) ? V3(0.0f, 0.0f, 1.0f) : V3(0.0f, 1.0f, 0.0f);
  vec3 b = normalize(cross(n, t));
  t = cross(b, n);
  return mat3{t, b, n};
}
inline vec3 localToWorld(vec3 n, vec3 v) { return normalize(mul(localRefMatrix(n), v)); }  // :18-20
inline vec3 sampleHemisphereCosine(vec3 n, vec2 r)                                         // :22-26
{
  vec2 d = toConcentricDisk(r);
  float z = rt_sqrt(1.0f - dot(d, d));
  return localToWorld(n, V3(d.x, d.y, z));
}
inline float satDot(vec3 a, vec3 b) { return rt_max(dot(a, b), 0.0f); }  // :28-30
inline float absDot(vec3 a, vec3 b) { return rt_abs(dot(a, b)); }        // :32-34
inline vec3 FresnelSchlick(float cosTheta, vec3 f0)                      // :36-41
{
  float cos4 = 1.0f - cosTheta;
  cos4 *= cos4;
  cos4 *= cos4;
  return mix(f0, V3(1.0f), cos4 * (1.0f - cosTheta));
}
inline float SchlickG(float cosTheta, float alpha) { float a = alpha * 0.5f; return cosTheta / (cosTheta * (1.0f - a) + a); }  // :43-46
inline float SmithG(float cosWo, float cosWi, float alpha) { return SchlickG(rt_abs(cosWo), alpha) * SchlickG(rt_abs(cosWi), alpha); }  // :48-50
inline float GTR2Distrib(float cosTheta, float alpha)  // :52-61
{
  if(cosTheta < 1e-6f) return 0.0f;
  float aa = alpha * alpha;
  float nom = aa;
  float denom = cosTheta * cosTheta * (aa - 1.0f) + 1.0f;
  denom = denom * denom * Pi;
  return nom / denom;
}
inline float GTR2Pdf(vec3 n, vec3 m, vec3 wo, float alpha)  // :63-65
{
  return GTR2Distrib(dot(n, m), alpha) * SchlickG(dot(n, wo), alpha) * absDot(m, wo) / absDot(n, wo);
}
inline vec3 GTR2Sample(vec3 n, vec3 wo, float alpha, vec2 r)  // :67-84
{
  mat3 transMat = localRefMatrix(n);
  mat3 transInv = inverse(transMat);
  vec3 vh = normalize(mul(transInv, wo) * V3(alpha, alpha, 1.0f));
  float lenSq = vh.x * vh.x + vh.y * vh.y;
  vec3 t = lenSq > 0.0f ? V3(-vh.y, vh.x, 0.0f) / rt_sqrt(lenSq) : V3(1.0f, 0.0f, 0.0f);
  vec3 b = cross(vh, t);
  vec2 p = toConcentricDisk(r);
  float s = 0.5f * (vh.z + 1.0f);
  p.y = (1.0f - s) * rt_sqrt(1.0f - p.x * p.x) + s * p.y;
  vec3 h = (t * p.x + b * p.y) + vh * rt_sqrt(rt_max(0.0f, 1.0f - dot(p, p)));
  h = V3(h.x * alpha, h.y * alpha, rt_max(0.0f, h.z));
  return normalize(mul(transMat, h));
}
inline vec3 metallicWorkflowBSDF(const State& state, vec3 n, vec3 wo, vec3 wi)  // :86-105
{
  vec3 baseColor = state.mat.albedo;
  float roughness = state.mat.roughness, metallic = state.mat.metallic;
  float alpha = roughness;
  vec3 h = normalize(wo + wi);
  float cosO = dot(n, wo), cosI = dot(n, wi);
  if(cosI * cosO < 1e-7f) return V3(0.0f);
  vec3 f = FresnelSchlick(dot(h, wo), mix(V3(.08f), baseColor, metallic));
  float g = SmithG(cosO, cosI, alpha);
  float d = GTR2Distrib(dot(n, h), alpha);
  return mix(baseColor * PiInv * (1.0f - metallic), V3(g * d / (4.0f * cosI * cosO)), f);
}
inline float metallicWorkflowPdf(const State& state, vec3 n, vec3 wo, vec3 wi)  // :107-121
{
  float roughness = state.mat.roughness, metallic = state.mat.metallic;
  float alpha = roughness;
  vec3 h = normalize(wo + wi);
  return mix(satDot(n, wi) * PiInv, GTR2Pdf(n, h, wo, alpha) / (4.0f * absDot(h, wo)), 1.0f / (2.0f - metallic));
}
inline vec3 metallicWorkflowEval(const State& state, vec3 n, vec3 wo, vec3 wi, float& pdf)  // :123-144
{
  vec3 baseColor = state.mat.albedo;
  float roughness = state.mat.roughness, metallic = state.mat.metallic;
  float alpha = roughness;
  vec3 h = normalize(wo + wi);
  float cosO = dot(n, wo), cosI = dot(n, wi);
  if(cosI * cosO < 1e-7f) return V3(0.0f);
  vec3 f = FresnelSchlick(dot(h, wo), mix(V3(.08f), baseColor, metallic));
  float g = SmithG(cosO, cosI, alpha);
  float d = GTR2Distrib(dot(n, h), alpha);
  pdf = mix(satDot(n, wi) * PiInv, GTR2Pdf(n, h, wo, alpha) / (4.0f * absDot(h, wo)), 1.0f / (2.0f - metallic));
  return mix(baseColor * PiInv * (1.0f - metallic), V3(g * d / (4.0f * cosI * cosO)), f);
}
inline float metallicWorkflowSample(const State& state, vec3 n, vec3 wo, vec3 r, vec3& bsdf, vec3& dir)  // :146-167
{
  float roughness = state.mat.roughness, metallic = state.mat.metallic;
  float alpha = roughness;
  if(r.z > (1.0f / (2.0f - metallic))) dir = sampleHemisphereCosine(n, V2(r.x, r.y));
  else {
    vec3 h = GTR2Sample(n, wo, alpha, V2(r.x, r.y));
    dir = -reflect(wo, h);
  }
  if(dot(n, dir) < 0.0f) return InvalidPdf;
  bsdf = metallicWorkflowBSDF(state, n, wo, dir);
  return metallicWorkflowPdf(state, n, wo, dir);
}

// ------------------------------------------------------------------------------------------------------------
// Per-invocation context: what the GLSL keeps in globals (prd, imageCoords, rtxState, sceneCamera, bindings).
struct Shader {
  const Scene& S;
  const rt_state& rtx;
  const rt_scene_camera& cam;
  uint32_t seed = 0;  // prd.seed
  ivec2 imageCoords{0, 0};
  // last hit payload (PtPayload, globals.glsl:54-65)
  float hitT = RT_INFINITY;
  uint32_t hitTri = 0xffffffffu;
  float hitU = 0, hitV = 0;
  uint32_t lastLightId = 0xffffffffu;  // oracle-side bookkeeping for RT_BUF_LIGHT_ID
  bool dbgPrint = false;

  Shader(const Scene& s, const rt_state& r, const rt_scene_camera& c) : S(s), rtx(r), cam(c) {}

  static const mat4& M(const rt_mat4& m) { return *reinterpret_cast<const mat4*>(&m); }

  // ----------------------------------------------------------------------------------- traceray_rq.glsl
  void ClosestHit(const Ray& r)  // :108-147
  {
    Hit h = S.closestHit(r.origin, r.direction, seed);
    hitT = h.t; hitTri = h.tri; hitU = h.u; hitV = h.v;
  }
  bool AnyHit(const Ray& r, float maxDist) { return S.anyHit(r.origin, r.direction, maxDist, seed); }  // :153-185

  // --------------------------------------------------------------------------------- gltf_material.glsl
  static vec4 SRGBtoLINEAR(vec4 c) { return V4(rt_pow(c.x, 2.2f), rt_pow(c.y, 2.2f), rt_pow(c.z, 2.2f), c.w); }  // :34-43
  void GetMetallicRoughness(State& state, const rt_material& material) const  // :52-91
  {
    float perceptualRoughness = material.pbrRoughnessFactor;
    float metallic = material.pbrMetallicFactor;
    if(material.pbrMetallicRoughnessTexture > -1) {
      vec4 mr = S.sampleTexture(material.pbrMetallicRoughnessTexture, state.texCoord);
      perceptualRoughness = mr.y * perceptualRoughness;
      metallic = mr.z * metallic;
    }
    vec4 baseColor = V4(material.pbrBaseColorFactor.x, material.pbrBaseColorFactor.y, material.pbrBaseColorFactor.z, material.pbrBaseColorFactor.w);
    if(material.pbrBaseColorTexture > -1) baseColor = baseColor * SRGBtoLINEAR(S.sampleTexture(material.pbrBaseColorTexture, state.texCoord));
    state.mat.albedo = xyz(baseColor);
    state.mat.metallic = metallic;
    state.mat.roughness = perceptualRoughness;
  }
  void GetMaterials(State& state, const Ray& r) const  // :130-176
  {
    const rt_material& material = S.materials[state.matID];
    mat3 TBN{state.tangent, state.bitangent, state.normal};
    if(material.normalTexture > -1) {
      vec3 normalVector = xyz(S.sampleTexture(material.normalTexture, state.texCoord));
      normalVector = normalize(normalVector * 2.0f - V3(1.0f));
      normalVector = normalVector * V3(material.normalTextureScale, material.normalTextureScale, 1.0f);
      state.normal = normalize(mul(TBN, normalVector));
      state.ffnormal = dot(state.normal, r.direction) <= 0.0f ? state.normal : -state.normal;
      CreateCoordinateSystem(state.ffnormal, state.tangent, state.bitangent);
    }
    state.mat.emission = toV(material.emissiveFactor);
    if(material.emissiveTexture > -1) state.mat.emission *= xyz(SRGBtoLINEAR(S.sampleTexture(material.emissiveTexture, state.texCoord)));
    state.isEmitter = ((state.mat.emission.x + state.mat.emission.y + state.mat.emission.z) > 1e-3f);
    GetMetallicRoughness(state, material);
    state.mat.roughness = rt_max(state.mat.roughness, 0.001f);
    state.mat.transmission = material.transmissionFactor;
    if(material.transmissionTexture > -1) state.mat.transmission *= S.sampleTexture(material.transmissionTexture, state.texCoord).x;
    state.mat.ior = material.ior;
    state.eta = dot(state.normal, state.ffnormal) > 0.0f ? (1.0f / state.mat.ior) : state.mat.ior;
  }

  // ------------------------------------------------------------------------------------ shade_state.glsl
  State GetState(vec3 rayDir) const  // GetState, shade_state.glsl:147-221 (payload = last ClosestHit)
  {
    Counters::local().hitsShaded++;
    State state;
    const Tri& T = S.tris[hitTri];
    const rt_instance& inst = S.instances[T.inst];
    const rt_prim_mesh& geo = S.primMeshes[inst.primMesh];  // geoInfo[instanceCustomIndex]
    const affine& o2w = S.objectToWorld[T.inst];
    const affine& w2o = S.worldToObject[T.inst];
    const vec3 bary = V3((1.0f - hitU) - hitV, hitU, hitV);
    const uint32_t* tri = &S.indices[geo.firstIndex + 3 * T.prim];
    const rt_vertex& attr0 = S.vertices[geo.vertexOffset + tri[0]];
    const rt_vertex& attr1 = S.vertices[geo.vertexOffset + tri[1]];
    const rt_vertex& attr2 = S.vertices[geo.vertexOffset + tri[2]];
    const uint32_t matIndex = uint32_t(geo.materialIndex > 0 ? geo.materialIndex : 0);

    const vec3 pos0 = toV(attr0.position), pos1 = toV(attr1.position), pos2 = toV(attr2.position);
    const vec3 position = (pos0 * bary.x + pos1 * bary.y) + pos2 * bary.z;
    const vec3 world_position = xformPoint(o2w, position);
    vec3 wpos0 = xformPoint(o2w, pos0), wpos1 = xformPoint(o2w, pos1), wpos2 = xformPoint(o2w, pos2);

    vec3 nrm0 = decompress_unit_vec(attr0.normal), nrm1 = decompress_unit_vec(attr1.normal), nrm2 = decompress_unit_vec(attr2.normal);
    vec3 normal = normalize((nrm0 * bary.x + nrm1 * bary.y) + nrm2 * bary.z);
    vec3 world_normal = normalize(xformNormal(w2o, normal));
    vec3 geom_normal = normalize(cross(pos1 - pos0, pos2 - pos0));
    vec3 wgeom_normal = normalize(xformNormal(w2o, geom_normal));

    float h0 = (rt_f2u(attr0.texcoord.y) & 1u) == 1u ? 1.0f : -1.0f;
    vec3 tng0 = decompress_unit_vec(attr0.tangent), tng1 = decompress_unit_vec(attr1.tangent), tng2 = decompress_unit_vec(attr2.tangent);
    vec3 tangent = (tng0 * bary.x + tng1 * bary.y) + tng2 * bary.z;
    tangent = normalize(tangent);
    vec3 world_tangent = normalize(xformDir(o2w, tangent));
    world_tangent = normalize(world_tangent - dot(world_tangent, world_normal) * world_normal);
    vec3 world_binormal = cross(world_normal, world_tangent) * h0;

    auto decode_texture = [](rt_vec2 t) { return V2(t.x, rt_u2f(rt_f2u(t.y) & ~1u)); };  // shade_state.glsl:54-57
    const vec2 uv0 = decode_texture(attr0.texcoord), uv1 = decode_texture(attr1.texcoord), uv2 = decode_texture(attr2.texcoord);
    const vec2 texcoord0 = (uv0 * bary.x + uv1 * bary.y) + uv2 * bary.z;

    state.position = world_position;
    state.normal = (dot(world_normal, wgeom_normal) > 0.0f) ? world_normal : -world_normal;
    state.ffnormal = dot(state.normal, rayDir) <= 0.0f ? state.normal : -state.normal;
    state.texCoord = texcoord0;
    state.tangent = world_tangent;
    state.bitangent = world_binormal;
    state.matID = matIndex;
    state.area = length(cross(wpos1 - wpos0, wpos2 - wpos0)) * 0.5f;
    return state;
  }

  // ----------------------------------------------------------------------------------- env_sampling.glsl
  vec3 Environment_sample(vec3 randVal, vec3& to_light, float& pdf, uint32_t& texelOut) const  // :38-99
  {
    vec3 xi = randVal;
    const uint32_t width = uint32_t(S.envW), height = uint32_t(S.envH);
    const uint32_t size = width * height;
    const uint32_t idx = std::min(rt_ftou(xi.x * float(size)), size - 1);
    const rt_impt_samp& sample_data = S.envAccel[idx];
    uint32_t env_idx;
    if(xi.y < sample_data.q) { env_idx = idx; xi.y /= sample_data.q; pdf = sample_data.pdf; }
    else { env_idx = uint32_t(sample_data.alias); xi.y = (xi.y - sample_data.q) / (1.0f - sample_data.q); pdf = sample_data.aliasPdf; }
    texelOut = env_idx;
    const uint32_t px = env_idx % width;
    uint32_t py = env_idx / width;
    const float u = (float(px) + xi.y) / float(width);
    const float phi = u * (2.0f * M_PI_F) - M_PI_F;
    float sin_phi = rt_sin(phi), cos_phi = rt_cos(phi);
    const float step_theta = M_PI_F / float(height);
    const float theta0 = float(py) * step_theta;
    const float cos_theta = rt_cos(theta0) * (1.0f - xi.z) + rt_cos(theta0 + step_theta) * xi.z;
    const float theta = rt_acos(cos_theta);
    const float sin_theta = rt_sin(theta);
    const float v = theta * M_1_OVER_PI_F;
    to_light = V3(cos_phi * sin_theta, cos_theta, sin_phi * sin_theta);
    return xyz(S.sampleEnv(V2(u, v)));
  }
  vec4 EnvSample(vec3& radiance)  // :105-135
  {
    if(S.sunAndSky.in_use == 1) {  // :111-125 ("#TODO: find proper light direction + PDF" in the reference: kept as is)
      const rt_sun_and_sky& ss = S.sunAndSky;
      const float sun_radius = (0.00465f * 10.0f) * ss.sun_disk_scale;
      vec3 T, B;
      CreateCoordinateSystem(toV(ss.sun_direction), T, B);
      vec3 d;
      d.x = rnd(seed) * sun_radius;
      d.y = rnd(seed) * sun_radius;
      d.z = rt_sqrt(rt_max(0.0f, (1.0f - d.x * d.x) - d.y * d.y));
      const vec3 lightDir = normalize((T * d.x + B * d.y) + toV(ss.sun_direction) * d.z);
      radiance = sky::sun_and_sky(ss, lightDir);
      lastLightId = 0xBFFFFFFFu;  // "sun & sky sample" (no texel)
      radiance *= rtx.hdrMultiplier;
      return V4(lightDir, 0.5f);
    }
    vec3 lightDir; float pdf; uint32_t texel;
    float r0 = rnd(seed), r1 = rnd(seed), r2 = rnd(seed);
    radiance = Environment_sample(V3(r0, r1, r2), lightDir, pdf, texel);
    lastLightId = 0x80000000u | texel;
    radiance *= rtx.hdrMultiplier;
    return V4(lightDir, pdf);
  }

  // -------------------------------------------------------------------------------------- pathtrace.glsl
  static bool IsPdfInvalid(float p) { return p <= 1e-8f || rt_isnan(p); }  // :14-16
  bool Occlusion(const Ray& ray, const State& state, float dist)            // :18-22
  {
    return AnyHit(ray, ((dist - rt_abs(ray.origin.x - state.position.x)) - rt_abs(ray.origin.y - state.position.y)) - rt_abs(ray.origin.z - state.position.z));
  }
  vec3 BSDF(const State& s, vec3 V, vec3 N, vec3 L) const { return metallicWorkflowBSDF(s, N, V, L); }    // :24-26
  float Pdf(const State& s, vec3 V, vec3 N, vec3 L) const { return metallicWorkflowPdf(s, N, V, L); }     // :28-30
  vec3 Eval(const State& s, vec3 V, vec3 N, vec3 L, float& pdf) const { return metallicWorkflowEval(s, N, V, L, pdf); }  // :32-34
  vec3 Sample(const State& s, vec3 V, vec3 N, vec3& L, float& pdf)                                         // :36-38
  {
    float r0 = rnd(seed), r1 = rnd(seed), r2 = rnd(seed);
    vec3 bsdf = V3(0.0f);
    pdf = metallicWorkflowSample(s, N, V, V3(r0, r1, r2), bsdf, L);
    return bsdf;
  }
  vec3 EnvRadiance(vec3 dir) const  // :40-47
  {
    if(S.sunAndSky.in_use == 1) return sky::sun_and_sky(S.sunAndSky, dir) * rtx.hdrMultiplier;
    return xyz(S.sampleEnv(GetSphericalUv(dir))) * rtx.hdrMultiplier;
  }
  vec3 EnvEval(vec3 dir, float& pdf) const  // :62-72
  {
    if(S.sunAndSky.in_use == 1) { pdf = 0.5f * rtx.environmentProb; return sky::sun_and_sky(S.sunAndSky, dir) * rtx.hdrMultiplier; }
    vec3 radiance = xyz(S.sampleEnv(GetSphericalUv(dir)));
    pdf = luminance(radiance) * rtx.envMapLuminIntegInv * rtx.environmentProb;
    return radiance;
  }
  vec3 LightEval(const State& state, float dist, vec3 dir, float& pdf) const  // :74-88
  {
    float lightProb = (1.0f - rtx.environmentProb);
    const rt_material& mat = S.materials[state.matID];
    vec3 emission = toV(mat.emissiveFactor);
    pdf = luminance(emission) * rtx.lightLuminIntegInv * lightProb;
    pdf *= dist * dist / absDot(state.ffnormal, dir);
    if(mat.emissiveTexture > -1) emission *= xyz(SRGBtoLINEAR(S.sampleTexture(mat.emissiveTexture, state.texCoord)));
    return emission / state.area;
  }
  float SampleTriangleLight(vec3 x, rt_light_sample& ls)  // :103-139
  {
    if(S.lightInfo.trigLightSize == 0) return InvalidPdf;
    int id = std::min(rt_ftoi(float(S.lightInfo.trigLightSize) * rnd(seed)), int(S.lightInfo.trigLightSize) - 1);
    if(rnd(seed) > S.trigLights[id].impSamp.q) id = S.trigLights[id].impSamp.alias;
    const rt_trig_light& light = S.trigLights[id];
    lastLightId = 0x40000000u | uint32_t(id);
    vec3 v0 = toV(light.v0), v1 = toV(light.v1), v2 = toV(light.v2);
    vec3 normal = cross(v1 - v0, v2 - v0);
    float area = length(normal) * 0.5f;
    normal = normalize(normal);
    // SampleTriangleUniform, :90-97
    float ru = rnd(seed), rv = rnd(seed);
    float r = rt_sqrt(rv);
    vec2 baryCoord = V2(1.0f - r, ru * r);
    vec3 y = (baryCoord.x * v0 + baryCoord.y * v1) + ((1.0f - baryCoord.x) - baryCoord.y) * v2;
    const rt_material& mat = S.materials[light.matIndex];
    vec3 emission = toV(mat.emissiveFactor);
    if(mat.emissiveTexture > -1) {
      vec2 uv = (baryCoord.x * V2(light.uv0.x, light.uv0.y) + baryCoord.y * V2(light.uv1.x, light.uv1.y))
                + ((1.0f - baryCoord.x) - baryCoord.y) * V2(light.uv2.x, light.uv2.y);
      emission *= xyz(SRGBtoLINEAR(S.sampleTexture(mat.emissiveTexture, uv)));
    }
    vec3 dir = y - x;
    float dist = length(dir);
    ls.Li = toR(emission / area);
    ls.wi = toR(dir / dist);
    ls.dist = dist;
    return light.impSamp.pdf * (dist * dist) / (area * rt_abs(dot(toV(ls.wi), normal)));
  }
  float SamplePuncLight(vec3 x, rt_light_sample& ls)  // :141-159
  {
    if(S.lightInfo.puncLightSize == 0) return InvalidPdf;
    int id = std::min(rt_ftoi(float(S.lightInfo.puncLightSize) * rnd(seed)), int(S.lightInfo.puncLightSize) - 1);
    if(rnd(seed) > S.puncLights[id].impSamp.q) id = S.puncLights[id].impSamp.alias;
    const rt_punc_light& light = S.puncLights[id];
    lastLightId = 0x20000000u | uint32_t(id);
    vec3 dir = toV(light.position) - x;
    float dist = length(dir);
    ls.Li = toR(toV(light.color) * light.intensity / (dist * dist));
    ls.wi = toR(dir / dist);
    ls.dist = dist;
    return light.impSamp.pdf;
  }
  float SampleDirectLightNoVisibility(vec3 pos, rt_light_sample& ls)  // :161-183
  {
    Counters::local().risCandidates++;
    lastLightId = 0xffffffffu;
    ls = zeroLightSample();
    float r = rnd(seed);
    if(r < rtx.environmentProb) {
      vec3 Li;
      vec4 dirAndPdf = EnvSample(Li);
      ls.Li = toR(Li);
      if(IsPdfInvalid(dirAndPdf.w)) return InvalidPdf;
      ls.wi = rt_vec3{dirAndPdf.x, dirAndPdf.y, dirAndPdf.z};
      ls.dist = RT_INFINITY;
      return dirAndPdf.w * rtx.environmentProb;
    }
    if(r < rtx.environmentProb + (1.0f - rtx.environmentProb) * S.lightInfo.trigSampProb)
      return (1.0f - rtx.environmentProb) * SampleTriangleLight(pos, ls) * S.lightInfo.trigSampProb;
    return (1.0f - rtx.environmentProb) * SamplePuncLight(pos, ls) * (1.0f - S.lightInfo.trigSampProb);
  }
  float SampleDirectLight(const State& state, vec3& radiance, vec3& dir)  // :185-203
  {
    rt_light_sample ls;
    float pdf = SampleDirectLightNoVisibility(state.position, ls);
    if(IsPdfInvalid(pdf)) return InvalidPdf;
    Ray shadowRay{OffsetRay(state.position, state.ffnormal), toV(ls.wi)};
    if(dbgPrint) {
      const float md = ((ls.dist - rt_abs(shadowRay.origin.x - state.position.x)) - rt_abs(shadowRay.origin.y - state.position.y)) - rt_abs(shadowRay.origin.z - state.position.z);
      fprintf(stderr, "ORC   shadow o %08x %08x %08x d %08x %08x %08x dist %08x tmax %08x pdf %08x seed %08x\n", rt_f2u(shadowRay.origin.x), rt_f2u(shadowRay.origin.y), rt_f2u(shadowRay.origin.z),
              rt_f2u(shadowRay.direction.x), rt_f2u(shadowRay.direction.y), rt_f2u(shadowRay.direction.z), rt_f2u(ls.dist), rt_f2u(md), rt_f2u(pdf), seed);
    }
    if(Occlusion(shadowRay, state, ls.dist)) return InvalidPdf;
    radiance = toV(ls.Li);
    dir = toV(ls.wi);
    return pdf;
  }
  vec3 DirectLight(const State& state, vec3 wo)  // :205-220
  {
    rt_light_sample ls;
    float pdf = SampleDirectLightNoVisibility(state.position, ls);
    if(IsPdfInvalid(pdf)) return V3(0.0f);
    Ray shadowRay{OffsetRay(state.position, state.ffnormal), toV(ls.wi)};
    if(Occlusion(shadowRay, state, ls.dist)) return V3(0.0f);
    float dummy = 0;
    return toV(ls.Li) * Eval(state, wo, state.ffnormal, toV(ls.wi), dummy) * rt_max(dot(state.ffnormal, toV(ls.wi)), 0.0f) / pdf;
  }
  vec3 clampRadiance(vec3 radiance) const  // :222-232
  {
    if(rt_isnan(radiance.x) || rt_isnan(radiance.y) || rt_isnan(radiance.z)) return V3(0.0f);
    float lum = luminance(radiance);
    if(lum > rtx.fireflyClampThreshold) radiance *= rtx.fireflyClampThreshold / lum;
    return radiance;
  }
  Ray raySpawn(ivec2 coord, ivec2 sizeImage) const  // :260-270
  {
    const vec2 pixelCenter = V2(float(coord.x), float(coord.y)) + 0.5f;
    const vec2 inUV = pixelCenter / V2(float(sizeImage.x), float(sizeImage.y));
    vec2 d = inUV * 2.0f - 1.0f;
    vec4 origin = mul(M(cam.viewInverse), V4(0, 0, 0, 1));
    vec4 target = mul(M(cam.projInverse), V4(d.x, d.y, 1, 1));
    vec4 direction = mul(M(cam.viewInverse), V4(normalize(xyz(target)), 0));
    return Ray{xyz(origin), normalize(xyz(direction))};
  }
  vec3 DebugInfo(const State& state) const  // :362-380
  {
    switch(rtx.debugging_mode) {
      case RT_DBG_METALLIC: return V3(state.mat.metallic);
      case RT_DBG_NORMAL: return (state.normal + V3(1.0f)) * .5f;
      case RT_DBG_DEPTH: return V3(0.0f);
      case RT_DBG_BASECOLOR: return state.mat.albedo;
      case RT_DBG_EMISSIVE: return state.mat.emission;
      case RT_DBG_ROUGHNESS: return V3(state.mat.roughness);
      case RT_DBG_TEXCOORD: return V3(state.texCoord.x, state.texCoord.y, 0);
    }
    return V3(1000, 0, 0);
  }
};

}  // namespace orc
