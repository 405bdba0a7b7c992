// shading.h — device shading library of the ReSTIR frame path (gfx950).
// Implements, over the DevScene pointers, the functions of the reference's shader library on the hot path
// (SURVEY.md §8a): random.glsl, common.glsl, compress.glsl, reservoir.glsl, pbr_metallicworkflow.glsl,
// gltf_material.glsl, shade_state.glsl:147-221, env_sampling.glsl:38-135, pathtrace.glsl.  Each function cites
// the GLSL it stands for.  Numerics follow include/rt_detmath.h + DESIGN.md §Numerics.
#pragma once
#include "traverse.h"
#include "sky.h"
#ifndef RT_SKY
#define RT_SKY 0
#endif

namespace rt {

// ------------------------------------------------------------------------------------------------ random.glsl
RT_DEV uint32_t tea(uint32_t val0, uint32_t val1)  // :34-48
{
  uint32_t v0 = val0, v1 = val1, s0 = 0;
#pragma unroll
  for(uint32_t n = 0; n < 16; n++) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}

// ----------------------------------------------------------------------------------------------- globals.glsl
constexpr float M_PI_F = 3.14159265358979323846f;
constexpr float M_1_OVER_PI_F = 0.318309886183790671538f;
constexpr float InvalidPdf = -1.0f;
constexpr float Pi = M_PI_F;
constexpr float PiInv = 1.0f / Pi;

struct Ray { f3 origin, direction; };
struct Material { f3 albedo, emission; float metallic, ior, roughness, transmission; };
struct State {  // globals.glsl:87-104 (depth, eta kept for parity of the material evaluation)
  float eta;
  f3 position, normal, tangent, bitangent, ffnormal;
  f2 texCoord;
  bool isEmitter;
  uint32_t matID;
  Material mat;
  float area;
};
RT_DEV State zeroState()
{
  State s;
  s.eta = 0.f; s.position = s.normal = s.tangent = s.bitangent = s.ffnormal = mk3(0.f);
  s.texCoord = mk2(0.f, 0.f); s.isEmitter = false; s.matID = 0;
  s.mat.albedo = s.mat.emission = mk3(0.f); s.mat.metallic = s.mat.ior = s.mat.roughness = s.mat.transmission = 0.f;
  s.area = 0.f;
  return s;
}

// ------------------------------------------------------------------------------------------------ common.glsl
RT_DEV f2 GetSphericalUv(f3 v)  // :63-70
{
  float gamma = rt_asin(-v.y);
  float theta = rt_atan2(v.z, v.x);
  return mk2(theta * M_1_OVER_PI_F * 0.5f, gamma * M_1_OVER_PI_F) + 0.5f;
}
RT_DEV void CreateCoordinateSystem(f3 N, f3& Nt, f3& Nb)  // :75-81
{
  Nt = normalize((rt_abs(N.z) > 0.99999f) ? mk3(-N.x * N.y, 1.0f - N.y * N.y, -N.y * N.z) : mk3(-N.x * N.z, -N.y * N.z, 1.0f - N.z * N.z));
  Nb = cross(Nt, N);
}
RT_DEV float offsetNudge(float p, int32_t of) { return rt_u2f(rt_f2u(p) + uint32_t((p < 0) ? -of : of)); }
RT_DEV f3 OffsetRay(f3 p, f3 n)  // :89-105
{
  const float intScale = 256.0f, floatScale = 1.0f / 65536.0f, origin = 1.0f / 32.0f;
  int32_t ofx = rt_ftoi(intScale * n.x), ofy = rt_ftoi(intScale * n.y), ofz = rt_ftoi(intScale * n.z);
  f3 p_i = mk3(offsetNudge(p.x, ofx), offsetNudge(p.y, ofy), offsetNudge(p.z, ofz));
  return mk3(rt_abs(p.x) < origin ? p.x + floatScale * n.x : p_i.x, rt_abs(p.y) < origin ? p.y + floatScale * n.y : p_i.y,
             rt_abs(p.z) < origin ? p.z + floatScale * n.z : p_i.z);
}
RT_DEV uint32_t hash8bit(uint32_t a) { return (a ^ (a >> 8)) << 24; }  // :141-143
RT_DEV f2 toConcentricDisk(f2 r)                                        // :170-174
{
  float rx = rt_sqrt(r.x);
  float theta = r.y * 2.0f * M_PI_F;
  return mk2(rt_cos(theta), rt_sin(theta)) * rx;
}
RT_DEV float powerHeuristic(float f, float g) { float f2v = f * f; return f2v / (f2v + g * g); }  // :176-179
RT_DEV bool inBound(i2 p, i2 pMin, i2 pMax) { return p.x >= pMin.x && p.x < pMax.x && p.y >= pMin.y && p.y < pMax.y; }
RT_DEV bool inBound(i2 p, i2 b) { return inBound(p, i2{0, 0}, b); }
RT_DEV f3 HDRToLDR(f3 c) { return c / (c + 1.0f); }   // :194-196
RT_DEV f3 LDRToHDR(f3 c) { return c / (1.01f - c); }  // :198-200

// ---------------------------------------------------------------------------------------------- compress.glsl
RT_DEV float roundHalfAway(float x) { float r = truncf(x); if(x - r >= 0.5f) r += 1.0f; return r; }
RT_DEV uint32_t packUnorm4x8(f4 v)
{
  uint32_t r = uint32_t(roundHalfAway(rt_clamp(v.x, 0.0f, 1.0f) * 255.0f));
  uint32_t g = uint32_t(roundHalfAway(rt_clamp(v.y, 0.0f, 1.0f) * 255.0f));
  uint32_t b = uint32_t(roundHalfAway(rt_clamp(v.z, 0.0f, 1.0f) * 255.0f));
  uint32_t a = uint32_t(roundHalfAway(rt_clamp(v.w, 0.0f, 1.0f) * 255.0f));
  return r | (g << 8) | (b << 16) | (a << 24);
}
RT_DEV f4 unpackUnorm4x8(uint32_t p)
{
  return mk4(unorm8ToFloat(p & 0xffu), unorm8ToFloat((p >> 8) & 0xffu), unorm8ToFloat((p >> 16) & 0xffu), unorm8ToFloat(p >> 24));
}
RT_DEV uint32_t compress_unit_vec(f3 nv)  // :111-139
{
  if((nv.x < 3.402823466e+38f) && !rt_isinf(nv.x)) {
    const float d = 32767.0f / ((rt_abs(nv.x) + rt_abs(nv.y)) + rt_abs(nv.z));
    int x = rt_ftoi(rintf(nv.x * d));
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
RT_DEV float short_to_floatm11(int v)  // :142-146
{
  return (v >= 0) ? (rt_u2f(0x3F800000u | (uint32_t(v) << 8)) - 1.0f) : (rt_u2f((0x80000000u | 0x3F800000u) | (uint32_t(-v) << 8)) + 1.0f);
}
RT_DEV f3 decompress_unit_vec(uint32_t packed)  // :149-180
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
    return normalize(mk3(short_to_floatm11(x), short_to_floatm11(y), zf));
  }
  return mk3(3.402823466e+38f);
}

// --------------------------------------------------------------------------------------------- reservoir.glsl
RT_DEV float resvToScalar(f3 x) { return luminance(x); }
RT_DEV bool resvInvalidW(float w) { return rt_isnan(w) || w < 0.0f; }  // resvInvalid :26-32

// ---------------------------------------------------------------------------------- pbr_metallicworkflow.glsl
RT_DEV m3 localRefMatrix(f3 n)  // :11-16
{
  f3 t = (rt_abs(n.y) > 0.9999f) ? mk3(0.0f, 0.0f, 1.0f) : mk3(0.0f, 1.0f, 0.0f);
  f3 b = normalize(cross(n, t));
  t = cross(b, n);
  return m3{t, b, n};
}
RT_DEV f3 localToWorld(f3 n, f3 v) { return normalize(mul(localRefMatrix(n), v)); }
RT_DEV f3 sampleHemisphereCosine(f3 n, f2 r)  // :22-26
{
  f2 d = toConcentricDisk(r);
  float z = rt_sqrt(1.0f - dot(d, d));
  return localToWorld(n, mk3(d.x, d.y, z));
}
RT_DEV float satDot(f3 a, f3 b) { return rt_max(dot(a, b), 0.0f); }
RT_DEV float absDot(f3 a, f3 b) { return rt_abs(dot(a, b)); }
RT_DEV f3 FresnelSchlick(float cosTheta, f3 f0)  // :36-41
{
  float cos4 = 1.0f - cosTheta;
  cos4 *= cos4;
  cos4 *= cos4;
  return mix(f0, mk3(1.0f), cos4 * (1.0f - cosTheta));
}
RT_DEV float SchlickG(float cosTheta, float alpha) { float a = alpha * 0.5f; return cosTheta / (cosTheta * (1.0f - a) + a); }
RT_DEV float SmithG(float cosWo, float cosWi, float alpha) { return SchlickG(rt_abs(cosWo), alpha) * SchlickG(rt_abs(cosWi), alpha); }
RT_DEV float GTR2Distrib(float cosTheta, float alpha)  // :52-61
{
  if(cosTheta < 1e-6f) return 0.0f;
  float aa = alpha * alpha;
  float nom = aa;
  float denom = cosTheta * cosTheta * (aa - 1.0f) + 1.0f;
  denom = denom * denom * Pi;
  return nom / denom;
}
RT_DEV float GTR2Pdf(f3 n, f3 m, f3 wo, float alpha)  // :63-65
{
  return GTR2Distrib(dot(n, m), alpha) * SchlickG(dot(n, wo), alpha) * absDot(m, wo) / absDot(n, wo);
}
RT_DEV f3 GTR2Sample(f3 n, f3 wo, float alpha, f2 r)  // :67-84
{
  m3 transMat = localRefMatrix(n);
  m3 transInv = inverse(transMat);
  f3 vh = normalize(mul(transInv, wo) * mk3(alpha, alpha, 1.0f));
  float lenSq = vh.x * vh.x + vh.y * vh.y;
  f3 t = lenSq > 0.0f ? mk3(-vh.y, vh.x, 0.0f) / rt_sqrt(lenSq) : mk3(1.0f, 0.0f, 0.0f);
  f3 b = cross(vh, t);
  f2 p = toConcentricDisk(r);
  float s = 0.5f * (vh.z + 1.0f);
  p.y = (1.0f - s) * rt_sqrt(1.0f - p.x * p.x) + s * p.y;
  f3 h = (t * p.x + b * p.y) + vh * rt_sqrt(rt_max(0.0f, 1.0f - dot(p, p)));
  h = mk3(h.x * alpha, h.y * alpha, rt_max(0.0f, h.z));
  return normalize(mul(transMat, h));
}
// BSDF value and pdf share everything but the last line; the two-in-one form evaluates exactly the expressions of
// metallicWorkflowBSDF (:86-105), metallicWorkflowPdf (:107-121) and metallicWorkflowEval (:123-144).
RT_DEV f3 metallicWorkflowBSDF(const Material& mat, f3 n, f3 wo, f3 wi)
{
  f3 baseColor = mat.albedo;
  float alpha = mat.roughness, metallic = mat.metallic;
  f3 h = normalize(wo + wi);
  float cosO = dot(n, wo), cosI = dot(n, wi);
  if(cosI * cosO < 1e-7f) return mk3(0.0f);
  f3 f = FresnelSchlick(dot(h, wo), mix(mk3(.08f), baseColor, metallic));
  float g = SmithG(cosO, cosI, alpha);
  float d = GTR2Distrib(dot(n, h), alpha);
  return mix(baseColor * PiInv * (1.0f - metallic), mk3(g * d / (4.0f * cosI * cosO)), f);
}
RT_DEV float metallicWorkflowPdf(const Material& mat, f3 n, f3 wo, f3 wi)
{
  float alpha = mat.roughness, metallic = mat.metallic;
  f3 h = normalize(wo + wi);
  return mixf(satDot(n, wi) * PiInv, GTR2Pdf(n, h, wo, alpha) / (4.0f * absDot(h, wo)), 1.0f / (2.0f - metallic));
}
RT_DEV float metallicWorkflowSample(const Material& mat, f3 n, f3 wo, f3 r, f3& bsdf, f3& dir)  // :146-167
{
  float alpha = mat.roughness, metallic = mat.metallic;
  if(r.z > (1.0f / (2.0f - metallic))) dir = sampleHemisphereCosine(n, mk2(r.x, r.y));
  else {
    f3 h = GTR2Sample(n, wo, alpha, mk2(r.x, r.y));
    dir = -reflect(wo, h);
  }
  if(dot(n, dir) < 0.0f) return InvalidPdf;
  bsdf = metallicWorkflowBSDF(mat, n, wo, dir);
  return metallicWorkflowPdf(mat, n, wo, dir);
}

// ------------------------------------------------------------------------------------------------------------
// Per-lane context: the GLSL globals prd / imageCoords plus the bound resources.
struct Ctx {
  const DevScene& S;
  const rt_state& rtx;
  const rt_scene_camera& cam;
  uint2* stack;
  uint32_t seed;  // prd.seed
  i2 imageCoords;
  RayHit hit;     // last ClosestHit payload
  uint32_t lastLightId;
  TravCounters tc;
  uint32_t nClosest, nAny, nShaded, nRis;

  RT_DEV Ctx(const DevScene& s, const rt_state& r, const rt_scene_camera& c, uint2* st)
      : S(s), rtx(r), cam(c), stack(st), seed(0), imageCoords{0, 0}, lastLightId(0xffffffffu), tc{0, 0}, nClosest(0), nAny(0), nShaded(0), nRis(0)
  {
    hit.t = RT_INFINITY; hit.gid = 0xffffffffu; hit.u = hit.v = 0.f;
  }

  // ------------------------------------------------------------------------------- traceray_rq.glsl:108-185
#if RT_WAVEPROF
  uint32_t cycClosest = 0, cycAny = 0;
  RT_DEV void ClosestHit(const Ray& r) { nClosest++; const uint64_t c0 = clock64(); traceRay<false>(S, r.origin, r.direction, RT_INFINITY, seed, stack, hit, tc); cycClosest += uint32_t(clock64() - c0); }
  RT_DEV bool AnyHit(const Ray& r, float maxDist) { nAny++; RayHit h; const uint64_t c0 = clock64(); const bool f = traceRay<true>(S, r.origin, r.direction, maxDist, seed, stack, h, tc); cycAny += uint32_t(clock64() - c0); return f; }
#else
  RT_DEV void ClosestHit(const Ray& r) { nClosest++; traceRay<false>(S, r.origin, r.direction, RT_INFINITY, seed, stack, hit, tc); }
  RT_DEV bool AnyHit(const Ray& r, float maxDist) { nAny++; RayHit h; return traceRay<true>(S, r.origin, r.direction, maxDist, seed, stack, h, tc); }
#endif

  // ----------------------------------------------------------------------------------- gltf_material.glsl
  RT_DEV static f4 SRGBtoLINEAR(f4 c) { return mk4(rt_pow(c.x, 2.2f), rt_pow(c.y, 2.2f), rt_pow(c.z, 2.2f), c.w); }  // :34-43
  RT_DEV void GetMaterials(State& state, const Ray& r) const  // :130-176 (+ GetMetallicRoughness :52-91)
  {
    const rt_material material = S.materials[state.matID];
    m3 TBN{state.tangent, state.bitangent, state.normal};
    if(material.normalTexture > -1) {
      f3 normalVector = xyz(sampleTexture(S, material.normalTexture, state.texCoord));
      normalVector = normalize(normalVector * 2.0f - mk3(1.0f));
      normalVector = normalVector * mk3(material.normalTextureScale, material.normalTextureScale, 1.0f);
      state.normal = normalize(mul(TBN, normalVector));
      state.ffnormal = dot(state.normal, r.direction) <= 0.0f ? state.normal : -state.normal;
      CreateCoordinateSystem(state.ffnormal, state.tangent, state.bitangent);
    }
    state.mat.emission = mk3(material.emissiveFactor);
    if(material.emissiveTexture > -1) state.mat.emission *= xyz(SRGBtoLINEAR(sampleTexture(S, material.emissiveTexture, state.texCoord)));
    state.isEmitter = ((state.mat.emission.x + state.mat.emission.y + state.mat.emission.z) > 1e-3f);
    {
      float perceptualRoughness = material.pbrRoughnessFactor;
      float metallic = material.pbrMetallicFactor;
      if(material.pbrMetallicRoughnessTexture > -1) {
        f4 mr = sampleTexture(S, material.pbrMetallicRoughnessTexture, state.texCoord);
        perceptualRoughness = mr.y * perceptualRoughness;
        metallic = mr.z * metallic;
      }
      f4 baseColor = mk4(material.pbrBaseColorFactor.x, material.pbrBaseColorFactor.y, material.pbrBaseColorFactor.z, material.pbrBaseColorFactor.w);
      if(material.pbrBaseColorTexture > -1) baseColor = baseColor * SRGBtoLINEAR(sampleTexture(S, material.pbrBaseColorTexture, state.texCoord));
      state.mat.albedo = xyz(baseColor);
      state.mat.metallic = metallic;
      state.mat.roughness = perceptualRoughness;
    }
    state.mat.roughness = rt_max(state.mat.roughness, 0.001f);
    state.mat.transmission = material.transmissionFactor;
    if(material.transmissionTexture > -1) state.mat.transmission *= sampleTexture(S, material.transmissionTexture, state.texCoord).x;
    state.mat.ior = material.ior;
    state.eta = dot(state.normal, state.ffnormal) > 0.0f ? (1.0f / state.mat.ior) : state.mat.ior;
  }

  // ------------------------------------------------------------------------------ shade_state.glsl:147-221
  RT_DEV State GetState(f3 rayDir)
  {
    nShaded++;
    State state = zeroState();
    const TriRef ref = S.triRef[hit.gid];
    const DevInstance* inst = &S.instances[ref.inst];
    const rt_prim_mesh geo = S.primMeshes[inst->primMesh];
    const float* o2w = inst->o2w;
    const float* w2o = inst->w2o;
    const f3 bary = mk3((1.0f - hit.u) - hit.v, hit.u, hit.v);
    const uint32_t* tri = &S.indices[geo.firstIndex + 3 * ref.prim];
    const rt_vertex attr0 = S.vertices[geo.vertexOffset + tri[0]];
    const rt_vertex attr1 = S.vertices[geo.vertexOffset + tri[1]];
    const rt_vertex attr2 = S.vertices[geo.vertexOffset + tri[2]];
    const uint32_t matIndex = uint32_t(geo.materialIndex > 0 ? geo.materialIndex : 0);

    const f3 pos0 = mk3(attr0.position), pos1 = mk3(attr1.position), pos2 = mk3(attr2.position);
    const f3 position = (pos0 * bary.x + pos1 * bary.y) + pos2 * bary.z;
    const f3 world_position = xformPoint(o2w, position);
    const f3 wpos0 = xformPoint(o2w, pos0), wpos1 = xformPoint(o2w, pos1), wpos2 = xformPoint(o2w, pos2);

    const f3 nrm0 = decompress_unit_vec(attr0.normal), nrm1 = decompress_unit_vec(attr1.normal), nrm2 = decompress_unit_vec(attr2.normal);
    const f3 normal = normalize((nrm0 * bary.x + nrm1 * bary.y) + nrm2 * bary.z);
    const f3 world_normal = normalize(xformNormal(w2o, normal));
    const f3 geom_normal = normalize(cross(pos1 - pos0, pos2 - pos0));
    const f3 wgeom_normal = normalize(xformNormal(w2o, geom_normal));

    const float h0 = (rt_f2u(attr0.texcoord.y) & 1u) == 1u ? 1.0f : -1.0f;
    const f3 tng0 = decompress_unit_vec(attr0.tangent), tng1 = decompress_unit_vec(attr1.tangent), tng2 = decompress_unit_vec(attr2.tangent);
    f3 tangent = (tng0 * bary.x + tng1 * bary.y) + tng2 * bary.z;
    tangent = normalize(tangent);
    f3 world_tangent = normalize(xformDir(o2w, tangent));
    world_tangent = normalize(world_tangent - dot(world_tangent, world_normal) * world_normal);
    const f3 world_binormal = cross(world_normal, world_tangent) * h0;

    const f2 uv0 = mk2(attr0.texcoord.x, rt_u2f(rt_f2u(attr0.texcoord.y) & ~1u));  // decode_texture :54-57
    const f2 uv1 = mk2(attr1.texcoord.x, rt_u2f(rt_f2u(attr1.texcoord.y) & ~1u));
    const f2 uv2 = mk2(attr2.texcoord.x, rt_u2f(rt_f2u(attr2.texcoord.y) & ~1u));
    const f2 texcoord0 = (uv0 * bary.x + uv1 * bary.y) + uv2 * bary.z;

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

  // ---------------------------------------------------------------------------- env_sampling.glsl:38-135
  RT_DEV f4 EnvSample(f3& radiance)
  {
    if(RT_SKY && S.sky) {  // :111-125
      const SkyPre& P = *S.sky;
      f3 T, B;
      CreateCoordinateSystem(P.rawSunDir, T, B);
      f3 d;
      d.x = rnd(seed) * P.sampleRadius;
      d.y = rnd(seed) * P.sampleRadius;
      d.z = rt_sqrt(rt_max(0.0f, (1.0f - d.x * d.x) - d.y * d.y));
      const f3 lightDir = normalize((T * d.x + B * d.y) + P.rawSunDir * d.z);
      radiance = skyfn::evaluate(P, lightDir);
      lastLightId = 0xBFFFFFFFu;
      radiance *= rtx.hdrMultiplier;
      return mk4(lightDir, 0.5f);
    }
    float r0 = rnd(seed), r1 = rnd(seed), r2 = rnd(seed);
    f3 xi = mk3(r0, r1, r2);
    const uint32_t width = uint32_t(S.envW), height = uint32_t(S.envH);
    const uint32_t size = width * height;
    const uint32_t idx = min(rt_ftou(xi.x * float(size)), size - 1);
    const rt_impt_samp sample_data = S.envAccel[idx];
    uint32_t env_idx; float pdf;
    if(xi.y < sample_data.q) { env_idx = idx; xi.y /= sample_data.q; pdf = sample_data.pdf; }
    else { env_idx = uint32_t(sample_data.alias); xi.y = (xi.y - sample_data.q) / (1.0f - sample_data.q); pdf = sample_data.aliasPdf; }
    lastLightId = 0x80000000u | env_idx;
    const uint32_t px = env_idx % width;
    const uint32_t py = env_idx / width;
    const float u = (float(px) + xi.y) / float(width);
    const float phi = u * (2.0f * M_PI_F) - M_PI_F;
    const float sin_phi = rt_sin(phi), cos_phi = rt_cos(phi);
    const float step_theta = M_PI_F / float(height);
    const float theta0 = float(py) * step_theta;
    const float cos_theta = rt_cos(theta0) * (1.0f - xi.z) + rt_cos(theta0 + step_theta) * xi.z;
    const float theta = rt_acos(cos_theta);
    const float sin_theta = rt_sin(theta);
    const float v = theta * M_1_OVER_PI_F;
    f3 lightDir = mk3(cos_phi * sin_theta, cos_theta, sin_phi * sin_theta);
    radiance = xyz(sampleEnv(S, mk2(u, v)));
    radiance *= rtx.hdrMultiplier;
    return mk4(lightDir, pdf);
  }

  // --------------------------------------------------------------------------------------- pathtrace.glsl
  RT_DEV static bool IsPdfInvalid(float p) { return p <= 1e-8f || rt_isnan(p); }  // :14-16
  RT_DEV bool Occlusion(const Ray& ray, f3 statePos, float dist)                  // :18-22
  {
    return AnyHit(ray, ((dist - rt_abs(ray.origin.x - statePos.x)) - rt_abs(ray.origin.y - statePos.y)) - rt_abs(ray.origin.z - statePos.z));
  }
  RT_DEV f3 Sample(const Material& mat, f3 V, f3 N, f3& L, float& pdf)  // :36-38
  {
    float r0 = rnd(seed), r1 = rnd(seed), r2 = rnd(seed);
    f3 bsdf = mk3(0.0f);
    pdf = metallicWorkflowSample(mat, N, V, mk3(r0, r1, r2), bsdf, L);
    return bsdf;
  }
  RT_DEV f3 EnvRadiance(f3 dir) const  // :40-47
  {
    if(RT_SKY && S.sky) return skyfn::evaluate(*S.sky, dir) * rtx.hdrMultiplier;
    return xyz(sampleEnv(S, GetSphericalUv(dir))) * rtx.hdrMultiplier;
  }
  RT_DEV f3 EnvEval(f3 dir, float& pdf) const  // :62-72
  {
    if(RT_SKY && S.sky) { pdf = 0.5f * rtx.environmentProb; return skyfn::evaluate(*S.sky, dir) * rtx.hdrMultiplier; }
    f3 radiance = xyz(sampleEnv(S, GetSphericalUv(dir)));
    pdf = luminance(radiance) * rtx.envMapLuminIntegInv * rtx.environmentProb;
    return radiance;
  }
  RT_DEV f3 LightEval(const State& state, float dist, f3 dir, float& pdf) const  // :74-88
  {
    float lightProb = (1.0f - rtx.environmentProb);
    const rt_material* mat = &S.materials[state.matID];
    f3 emission = mk3(mat->emissiveFactor);
    pdf = luminance(emission) * rtx.lightLuminIntegInv * lightProb;
    pdf *= dist * dist / absDot(state.ffnormal, dir);
    const int et = mat->emissiveTexture;
    if(et > -1) emission *= xyz(SRGBtoLINEAR(sampleTexture(S, et, state.texCoord)));
    return emission / state.area;
  }
  RT_DEV float SampleTriangleLight(f3 x, rt_light_sample& ls)  // :103-139
  {
    if(S.lightInfo.trigLightSize == 0) return InvalidPdf;
    int id = min(rt_ftoi(float(S.lightInfo.trigLightSize) * rnd(seed)), int(S.lightInfo.trigLightSize) - 1);
    if(rnd(seed) > S.trigLights[id].impSamp.q) id = S.trigLights[id].impSamp.alias;
    const rt_trig_light light = S.trigLights[id];
    lastLightId = 0x40000000u | uint32_t(id);
    f3 v0 = mk3(light.v0), v1 = mk3(light.v1), v2 = mk3(light.v2);
    f3 normal = cross(v1 - v0, v2 - v0);
    float area = length(normal) * 0.5f;
    normal = normalize(normal);
    float ru = rnd(seed), rv = rnd(seed);  // SampleTriangleUniform :90-97
    float r = rt_sqrt(rv);
    f2 baryCoord = mk2(1.0f - r, ru * r);
    f3 y = (baryCoord.x * v0 + baryCoord.y * v1) + ((1.0f - baryCoord.x) - baryCoord.y) * v2;
    const rt_material* mat = &S.materials[light.matIndex];
    f3 emission = mk3(mat->emissiveFactor);
    const int et = mat->emissiveTexture;
    if(et > -1) {
      f2 uv = (baryCoord.x * mk2(light.uv0.x, light.uv0.y) + baryCoord.y * mk2(light.uv1.x, light.uv1.y))
              + ((1.0f - baryCoord.x) - baryCoord.y) * mk2(light.uv2.x, light.uv2.y);
      emission *= xyz(SRGBtoLINEAR(sampleTexture(S, et, uv)));
    }
    f3 dir = y - x;
    float dist = length(dir);
    f3 wi = dir / dist;
    ls.Li = toR(emission / area);
    ls.wi = toR(wi);
    ls.dist = dist;
    return light.impSamp.pdf * (dist * dist) / (area * rt_abs(dot(wi, normal)));
  }
  RT_DEV float SamplePuncLight(f3 x, rt_light_sample& ls)  // :141-159
  {
    if(S.lightInfo.puncLightSize == 0) return InvalidPdf;
    int id = min(rt_ftoi(float(S.lightInfo.puncLightSize) * rnd(seed)), int(S.lightInfo.puncLightSize) - 1);
    if(rnd(seed) > S.puncLights[id].impSamp.q) id = S.puncLights[id].impSamp.alias;
    const rt_punc_light light = S.puncLights[id];
    lastLightId = 0x20000000u | uint32_t(id);
    f3 dir = mk3(light.position) - x;
    float dist = length(dir);
    ls.Li = toR(mk3(light.color) * light.intensity / (dist * dist));
    ls.wi = toR(dir / dist);
    ls.dist = dist;
    return light.impSamp.pdf;
  }
  RT_DEV float SampleDirectLightNoVisibility(f3 pos, rt_light_sample& ls)  // :161-183
  {
    nRis++;
    lastLightId = 0xffffffffu;
    ls.Li = rt_vec3{0.f, 0.f, 0.f}; ls.wi = rt_vec3{0.f, 0.f, 0.f}; ls.dist = 0.f;
    float r = rnd(seed);
    if(r < rtx.environmentProb) {
      f3 Li;
      f4 dirAndPdf = EnvSample(Li);
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
  RT_DEV float SampleDirectLight(const State& state, f3& radiance, f3& dir)  // :185-203
  {
    rt_light_sample ls;
    float pdf = SampleDirectLightNoVisibility(state.position, ls);
    if(IsPdfInvalid(pdf)) return InvalidPdf;
    Ray shadowRay{OffsetRay(state.position, state.ffnormal), mk3(ls.wi)};
    if(Occlusion(shadowRay, state.position, ls.dist)) return InvalidPdf;
    radiance = mk3(ls.Li);
    dir = mk3(ls.wi);
    return pdf;
  }
  RT_DEV f3 DirectLight(const State& state, f3 wo)  // :205-220
  {
    rt_light_sample ls;
    float pdf = SampleDirectLightNoVisibility(state.position, ls);
    if(IsPdfInvalid(pdf)) return mk3(0.0f);
    Ray shadowRay{OffsetRay(state.position, state.ffnormal), mk3(ls.wi)};
    if(Occlusion(shadowRay, state.position, ls.dist)) return mk3(0.0f);
    return mk3(ls.Li) * metallicWorkflowBSDF(state.mat, state.ffnormal, wo, mk3(ls.wi)) * rt_max(dot(state.ffnormal, mk3(ls.wi)), 0.0f) / pdf;
  }
  RT_DEV f3 clampRadiance(f3 radiance) const  // :222-232
  {
    if(rt_isnan(radiance.x) || rt_isnan(radiance.y) || rt_isnan(radiance.z)) return mk3(0.0f);
    float lum = luminance(radiance);
    if(lum > rtx.fireflyClampThreshold) radiance *= rtx.fireflyClampThreshold / lum;
    return radiance;
  }
  RT_DEV Ray raySpawn(i2 coord, i2 sizeImage) const  // :260-270
  {
    const f2 pixelCenter = mk2(float(coord.x), float(coord.y)) + 0.5f;
    const f2 inUV = pixelCenter / mk2(float(sizeImage.x), float(sizeImage.y));
    f2 d = inUV * 2.0f - 1.0f;
    f4 origin = mul(cam.viewInverse, mk4(0, 0, 0, 1));
    f4 target = mul(cam.projInverse, mk4(d.x, d.y, 1, 1));
    f4 direction = mul(cam.viewInverse, mk4(normalize(xyz(target)), 0));
    return Ray{xyz(origin), normalize(xyz(direction))};
  }
  RT_DEV f3 DebugInfo(const State& state) const  // :362-380
  {
    switch(rtx.debugging_mode) {
      case RT_DBG_METALLIC: return mk3(state.mat.metallic);
      case RT_DBG_NORMAL: return (state.normal + mk3(1.0f)) * .5f;
      case RT_DBG_DEPTH: return mk3(0.0f);
      case RT_DBG_BASECOLOR: return state.mat.albedo;
      case RT_DBG_EMISSIVE: return state.mat.emission;
      case RT_DBG_ROUGHNESS: return mk3(state.mat.roughness);
      case RT_DBG_TEXCOORD: return mk3(state.texCoord.x, state.texCoord.y, 0);
    }
    return mk3(1000, 0, 0);
  }
};

}  // namespace rt
