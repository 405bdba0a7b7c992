// kat_float.cpp — mints known-answer vectors for the FLOAT helpers of the hot path from the REFERENCE's own GLSL, compiled
// where it lies under /root/reference (authoring container only).  Nothing of the reference is copied into this repository:
// build_ref.sh cuts the excerpts into a temp dir (qualifiers rewritten to C++ references, float literals given an `f`
// suffix so that `1.0 - x` stays single precision as in GLSL, the two `.xy` swizzles spelled out) and this file only
// #includes them on top of a small GLSL vector shim.
//   ref_globals.inc   <- shaders/globals.glsl (M_PI, M_PI_2 constants)
//   ref_common2.inc   <- shaders/common.glsl (InvalidPdf, toConcentricDisk, powerHeuristic, HDRToLDR, LDRToHDR)
//   ref_common3.inc   <- shaders/common.glsl (GetSphericalUv, CreateCoordinateSystem)
//   ref_lum.inc       <- shaders/denoise_common.glsl (luminance)
//   ref_structs.inc   <- shaders/host_device.h (LightSample, GISample, DirectReservoir, IndirectReservoir)
//   ref_reservoir.inc <- shaders/reservoir.glsl (all of it below the #include)
//   ref_pbr.inc       <- shaders/pbr_metallicworkflow.glsl (all of it below the #includes)
//   ref_pcg3d.inc     <- shaders/random.glsl (pcg3d)
//   ref_tonemap.inc   <- shaders/tonemapping.glsl (GAMMA, linearTosRGB, sRGBToLinear(vec3), toneMapUncharted2Impl, toneMapUncharted)
//   ref_tmstruct.inc  <- shaders/host_device.h (Tonemapper)
//   ref_post.inc      <- shaders/post.frag (dither, RGB2XYZ, luminance, toneExposure; `tm` is the push constant)
//   ref_skystruct.inc <- shaders/host_device.h (SunAndSky)
//   ref_sky.inc       <- shaders/sun_and_sky.glsl (all of it inside the include guard)
// The shim's sqrt / sin / cos / acos are glibc's; the GLSL ones are the driver's and the oracle's are rt_detmath.h's: the
// transcendental-dependent vectors are therefore compared WITH A TOLERANCE (tests/test_kat_float.py states it), the
// reservoir arithmetic (+ * / and compares only) bit-exactly.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#undef M_PI
#undef M_PI_2
#undef M_PI_4
typedef unsigned int uint;

struct vec2 { float x, y; vec2() : x(0), y(0) {} vec2(float a, float b) : x(a), y(b) {} };
struct vec3 {
  float x, y, z;
  vec3() : x(0), y(0), z(0) {}
  vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  explicit vec3(float s) : x(s), y(s), z(s) {}
  vec3(vec2 a, float c) : x(a.x), y(a.y), z(c) {}
};
static inline vec2 operator*(vec2 a, float s) { return vec2(a.x * s, a.y * s); }
static inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
static inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator*(float s, vec3 a) { return vec3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
static inline vec3 operator/(vec3 a, vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline vec3 operator+(vec3 a, float s) { return vec3(a.x + s, a.y + s, a.z + s); }
static inline vec3 operator-(float s, vec3 a) { return vec3(s - a.x, s - a.y, s - a.z); }
static inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }
static inline vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
static inline vec3& operator/=(vec3& a, float s) { a = a / s; return a; }
static inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
static inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
static inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float sqrt(float v) { return std::sqrt(v); }
static inline float cos(float v) { return std::cos(v); }
static inline float sin(float v) { return std::sin(v); }
static inline float abs(float v) { return std::fabs(v); }
static inline float max(float a, float b) { return a > b ? a : b; }
static inline vec3 normalize(vec3 v) { float l = std::sqrt(dot(v, v)); return vec3(v.x / l, v.y / l, v.z / l); }
static inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline vec3 mix(vec3 a, vec3 b, float t) { return vec3(mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t)); }
static inline vec3 mix(vec3 a, vec3 b, vec3 t) { return vec3(mix(a.x, b.x, t.x), mix(a.y, b.y, t.y), mix(a.z, b.z, t.z)); }
static inline vec3 reflect(vec3 i, vec3 n) { return i - 2.0f * dot(n, i) * n; }
struct mat3 { vec3 c[3]; mat3(vec3 a, vec3 b, vec3 d) { c[0] = a; c[1] = b; c[2] = d; } };
static inline vec3 operator*(const mat3& m, vec3 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; }
static inline mat3 inverse(const mat3& m)
{
  const vec3 a = m.c[0], b = m.c[1], c = m.c[2];
  const vec3 r0 = cross(b, c), r1 = cross(c, a), r2 = cross(a, b);
  const float id = 1.0f / dot(a, r0);
  return mat3(vec3(r0.x, r1.x, r2.x) * id, vec3(r0.y, r1.y, r2.y) * id, vec3(r0.z, r1.z, r2.z) * id);
}
using std::isnan;
struct Material { vec3 albedo; float metallic, roughness; };
struct State { Material mat; };

static inline float asin(float v) { return std::asin(v); }
static inline float atan(float y, float x) { return std::atan2(y, x); }
static inline vec2 operator+(vec2 a, float s) { return vec2(a.x + s, a.y + s); }
namespace core {   // (a namespace so that sun_and_sky.glsl's own luminance() below is not ambiguous through ADL)
#include "ref_globals.inc"
#include "ref_common2.inc"
#include "ref_common3.inc"
#include "ref_lum.inc"
#include "ref_structs.inc"
#include "ref_reservoir.inc"
#include "ref_pbr.inc"
}
using namespace core;
// sun_and_sky.glsl brings its own luminance() and an M_PI macro: keep it in a namespace, included last
static inline float exp(float v) { return std::exp(v); }
static inline float pow(float a, float b) { return std::pow(a, b); }
static inline float acos(float v) { return std::acos(v); }
static inline vec3 exp(vec3 v) { return vec3(std::exp(v.x), std::exp(v.y), std::exp(v.z)); }
static inline vec3 pow(vec3 a, vec3 b) { return vec3(std::pow(a.x, b.x), std::pow(a.y, b.y), std::pow(a.z, b.z)); }
static inline float tan(float v) { return std::tan(v); }
static inline float length(vec3 v) { return std::sqrt(dot(v, v)); }
static inline float min(float a, float b) { return a < b ? a : b; }
static inline float clamp(float v, float a, float b) { return min(max(v, a), b); }
static inline float smoothstep(float e0, float e1, float x) { float t = clamp((x - e0) / (e1 - e0), 0.0f, 1.0f); return t * t * (3.0f - 2.0f * t); }
// display pass: post.frag + tonemapping.glsl + random.glsl pcg3d
struct uvec3 { uint x, y, z; uvec3(uint a, uint b, uint c) : x(a), y(b), z(c) {} explicit uvec3(uint s) : x(s), y(s), z(s) {} };
static inline uvec3 operator*(uvec3 a, uint s) { return uvec3(a.x * s, a.y * s, a.z * s); }
static inline uvec3 operator+(uvec3 a, uvec3 b) { return uvec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline uvec3 operator>>(uvec3 a, uvec3 b) { return uvec3(a.x >> b.x, a.y >> b.y, a.z >> b.z); }
static inline uvec3& operator^=(uvec3& a, uvec3 b) { a.x ^= b.x; a.y ^= b.y; a.z ^= b.z; return a; }
struct bvec3 { bool x, y, z; };
static inline bvec3 lessThan(vec3 a, vec3 b) { return bvec3{a.x < b.x, a.y < b.y, a.z < b.z}; }
static inline vec3 mix(vec3 a, vec3 b, bvec3 t) { return vec3(t.x ? b.x : a.x, t.y ? b.y : a.y, t.z ? b.z : a.z); }
static inline vec3 floor(vec3 v) { return vec3(std::floor(v.x), std::floor(v.y), std::floor(v.z)); }
static inline vec3 operator-(vec3 a, float s) { return vec3(a.x - s, a.y - s, a.z - s); }
static inline vec3 operator/(float s, vec3 a) { return vec3(s / a.x, s / a.y, s / a.z); }
struct mat3c : mat3 { mat3c(float a, float b, float c, float d, float e, float f, float g, float h, float i) : mat3(vec3(a, b, c), vec3(d, e, f), vec3(g, h, i)) {} };
namespace post {
#define mat3 mat3c
#include "ref_pcg3d.inc"
#include "ref_tonemap.inc"
#include "ref_tmstruct.inc"
static Tonemapper tm;
#include "ref_post.inc"
#undef mat3
}
namespace sky {
#include "ref_skystruct.inc"
#include "ref_sky.inc"
}

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
static float u01(uint32_t& s) { return float(lcg(s) >> 8) / 16777216.0f; }
static uint32_t fb(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static void p3(vec3 v) { printf("%u,%u,%u", fb(v.x), fb(v.y), fb(v.z)); }
static vec3 randDir(uint32_t& s) { vec3 v; do { v = vec3(u01(s) * 2 - 1, u01(s) * 2 - 1, u01(s) * 2 - 1); } while(dot(v, v) < 0.01f || dot(v, v) > 1.0f); return normalize(v); }

int main()
{
  uint32_t s = 777u;
  printf("{\n");
  // ---- toConcentricDisk / powerHeuristic / luminance / HDR<->LDR ----
  printf("\"concentric_disk\": [");
  for(int i = 0; i < 64; i++) { vec2 r(u01(s), u01(s)); if(i == 0) r = vec2(0, 0); if(i == 1) r = vec2(1, 0.25f); vec2 d = toConcentricDisk(r); printf("%s[%u,%u,%u,%u]", i ? "," : "", fb(r.x), fb(r.y), fb(d.x), fb(d.y)); }
  printf("],\n\"power_heuristic\": [");
  for(int i = 0; i < 32; i++) { float f = u01(s) * 10, g = u01(s) * (i % 2 ? 0.1f : 10); printf("%s[%u,%u,%u]", i ? "," : "", fb(f), fb(g), fb(powerHeuristic(f, g))); }
  printf("],\n\"luminance_ldr\": [");
  for(int i = 0; i < 32; i++) {
    vec3 c(u01(s) * 4, u01(s) * 4, u01(s) * 4); vec3 l = HDRToLDR(c), h = LDRToHDR(l);
    printf("%s[", i ? "," : ""); p3(c); printf(",%u,", fb(luminance(c))); p3(l); printf(","); p3(h); printf("]");
  }
  printf("],\n");
  printf("\"spherical_uv_coord_system\": [");
  for(int i = 0; i < 64; i++) {
    vec3 d = randDir(s); if(i == 0) d = vec3(0, 0, 1); if(i == 1) d = vec3(0, 0, -1); if(i == 2) d = vec3(0, 1, 0); if(i == 3) d = normalize(vec3(0.001f, 0.002f, -1.0f));
    vec2 uv = GetSphericalUv(d); vec3 t, b; CreateCoordinateSystem(d, t, b);
    printf("%s[", i ? "," : ""); p3(d); printf(",%u,%u,", fb(uv.x), fb(uv.y)); p3(t); printf(","); p3(b); printf("]");
  }
  printf("],\n");
  // ---- metallic workflow BSDF: Sample (dir, pdf, f) and Eval (f, pdf) ----
  printf("\"bsdf\": [");
  for(int i = 0; i < 400; i++) {
    State st;
    st.mat.albedo = vec3(0.05f + 0.95f * u01(s), 0.05f + 0.95f * u01(s), 0.05f + 0.95f * u01(s));
    st.mat.metallic = (i % 3 == 0) ? 0.0f : ((i % 3 == 1) ? 1.0f : u01(s));
    st.mat.roughness = 0.05f + 0.95f * u01(s);
    vec3 n = (i % 5 == 0) ? vec3(0, 1, 0) : randDir(s);
    vec3 wo; do { wo = randDir(s); } while(dot(wo, n) < 0.05f);
    vec3 r(u01(s), u01(s), u01(s));
    vec3 f(0.0f), dir(0.0f);
    float pdf = metallicWorkflowSample(st, n, wo, r, f, dir);
    vec3 wi; do { wi = randDir(s); } while(fabsf(dot(wi, n)) < 0.02f);
    float epdf = 0.0f;
    vec3 ef = metallicWorkflowEval(st, n, wo, wi, epdf);
    printf("%s[", i ? "," : "");
    p3(st.mat.albedo); printf(",%u,%u,", fb(st.mat.metallic), fb(st.mat.roughness)); p3(n); printf(","); p3(wo); printf(","); p3(r); printf(",");
    p3(dir); printf(",%u,", fb(pdf)); p3(f); printf(","); p3(wi); printf(","); p3(ef); printf(",%u]", fb(epdf));
  }
  printf("],\n");
  // ---- reservoirs: an op stream over one DirectReservoir and one IndirectReservoir; state logged after each op ----
  // ops: 0 update(w, r, tag)  1 merge(rhs{num, weight, tag}, r) [direct only]  2 clamp(c)  3 checkValidity  4 reset
  printf("\"reservoir_ops\": [");
  DirectReservoir d; IndirectReservoir g;
  memset(&d, 0, sizeof(d)); memset(&g, 0, sizeof(g));
  for(int i = 0; i < 600; i++) {
    int op = (i == 0) ? 4 : int(lcg(s) >> 8) % 11; op = op < 5 ? 0 : (op < 8 ? 1 : (op == 8 ? 2 : (op == 9 ? 3 : 4)));
    if(op == 4 && (lcg(s) >> 8) % 4) op = 0;
    float w = u01(s) * ((i % 7 == 0) ? 100.0f : 1.0f), r = u01(s), tag = float(i);
    if(i % 53 == 17) w = -1.0f;                       // drives resvInvalid
    if(i % 97 == 31) w = std::nanf("");
    uint rn = 1 + (lcg(s) >> 8) % 80; int c = 1 + int(lcg(s) >> 8) % 40;
    printf("%s[%d,%u,%u,%u,%u,%d", i ? "," : "", op, fb(w), fb(r), fb(tag), rn, c);
    switch(op) {
      case 0: { LightSample ls; ls.Li = vec3(tag); ls.wi = vec3(0, 0, 1); ls.dist = tag; resvUpdate(d, ls, w, r);
                GISample gs; memset(&gs, 0, sizeof(gs)); gs.L = vec3(tag); gs.pHat = tag; resvUpdate(g, gs, w, r); } break;
      case 1: { DirectReservoir rhs; rhs.lightSample.Li = vec3(tag); rhs.lightSample.wi = vec3(0, 0, 1); rhs.lightSample.dist = tag; rhs.num = rn; rhs.weight = w;
                if(!resvInvalid(rhs)) resvMerge(d, rhs, r); } break;
      case 2: resvClamp(d, c); resvClamp(g, c); break;
      case 3: resvCheckValidity(d); resvCheckValidity(g); break;
      default: resvReset(d); resvReset(g); break;
    }
    printf(",%u,%u,%u,%d,%u,%u,%u,%d]", d.num, fb(d.weight), fb(d.lightSample.dist), resvInvalid(d) ? 1 : 0, g.num, fb(g.weight), fb(g.giSample.pHat), resvInvalid(g) ? 1 : 0);
  }
  printf("],\n");
  // ---- sun_and_sky(): a few parameter sets x directions ----
  printf("\"sun_and_sky\": [");
  for(int k = 0; k < 6; k++) {
    sky::SunAndSky ss;
    ss.rgb_unit_conversion = vec3(1.0f / 80000.0f); ss.multiplier = 0.1f; ss.haze = 0.1f; ss.redblueshift = 0.1f; ss.saturation = 1.0f;
    ss.horizon_height = 0.0f; ss.ground_color = vec3(0.4f); ss.horizon_blur = 0.3f; ss.night_color = vec3(0.0f); ss.sun_disk_intensity = 1.0f;
    ss.sun_direction = vec3(0.0f, 0.7071f, 0.7071f); ss.sun_disk_scale = 1.0f; ss.sun_glow_intensity = 1.0f; ss.y_is_up = 1; ss.physically_scaled_sun = 0; ss.in_use = 1;
    if(k == 1) { ss.haze = 2.5f; ss.redblueshift = -0.3f; ss.saturation = 0.6f; ss.sun_direction = normalize(vec3(0.3f, 0.15f, -0.8f)); }
    if(k == 2) { ss.physically_scaled_sun = 1; ss.sun_disk_scale = 2.0f; ss.sun_glow_intensity = 0.5f; ss.horizon_height = 0.2f; ss.horizon_blur = 1.0f; }
    if(k == 3) { ss.sun_direction = normalize(vec3(0.5f, -0.2f, 0.4f)); ss.night_color = vec3(0.01f, 0.01f, 0.03f); }      // sun below the horizon
    if(k == 4) { ss.y_is_up = 0; ss.sun_direction = normalize(vec3(0.2f, 0.3f, 0.9f)); ss.ground_color = vec3(0.1f, 0.3f, 0.05f); ss.multiplier = 1.0f; }
    if(k == 5) { ss.haze = 10.0f; ss.saturation = 1.7f; ss.redblueshift = 0.8f; ss.sun_disk_intensity = 3.0f; ss.sun_direction = normalize(vec3(-0.6f, 0.05f, 0.2f)); }
    printf("%s{\"ss\":[", k ? "," : "");
    p3(ss.rgb_unit_conversion); printf(",%u,%u,%u,%u,%u,", fb(ss.multiplier), fb(ss.haze), fb(ss.redblueshift), fb(ss.saturation), fb(ss.horizon_height));
    p3(ss.ground_color); printf(",%u,", fb(ss.horizon_blur)); p3(ss.night_color); printf(",%u,", fb(ss.sun_disk_intensity)); p3(ss.sun_direction);
    printf(",%u,%u,%d,%d,%d],\"samples\":[", fb(ss.sun_disk_scale), fb(ss.sun_glow_intensity), ss.y_is_up, ss.physically_scaled_sun, ss.in_use);
    for(int i = 0; i < 48; i++) {
      vec3 d = randDir(s);
      if(i < 6) {  // towards / around the sun disk and straight up / down
        vec3 sd = ss.sun_direction; if(!ss.y_is_up) sd = vec3(sd.x, sd.z, sd.y);
        d = (i == 0) ? ss.sun_direction : (i == 1 ? normalize(ss.sun_direction + vec3(0.004f, 0.0f, 0.002f)) : (i == 2 ? normalize(ss.sun_direction + vec3(0.03f, -0.01f, 0.0f)) : (i == 3 ? vec3(0, 1, 0) : (i == 4 ? vec3(0, -1, 0) : normalize(vec3(1, 0.001f, 0))))));
        (void)sd;
      }
      vec3 c = sky::sun_and_sky(ss, d);
      printf("%s[", i ? "," : ""); p3(d); printf(","); p3(c); printf("]");
    }
    printf("]}");
  }
  printf("],\n");
  // ---- display pass helpers ----
  printf("\"pcg3d\": [");
  for(int i = 0; i < 32; i++) { uvec3 v(i < 4 ? uint(i) : lcg(s) % 4096u, i < 4 ? uint(3 - i) : lcg(s) % 4096u, 0u); uvec3 r = post::pcg3d(v); printf("%s[%u,%u,%u,%u,%u,%u]", i ? "," : "", v.x, v.y, v.z, r.x, r.y, r.z); }
  printf("],\n\"uncharted\": [");
  for(int i = 0; i < 64; i++) { float sc = (i % 4 == 0) ? 0.05f : ((i % 4 == 1) ? 1.0f : ((i % 4 == 2) ? 8.0f : 100.0f)); vec3 c(u01(s) * sc, u01(s) * sc, u01(s) * sc); if(i == 0) c = vec3(0.0f);
    vec3 t = post::toneMapUncharted(c); printf("%s[", i ? "," : ""); p3(c); printf(","); p3(t); printf("]"); }
  printf("],\n\"dither\": [");
  for(int i = 0; i < 64; i++) { vec3 c(u01(s), u01(s), u01(s)); vec3 nz(u01(s), u01(s), u01(s)); if(i % 8 == 0) c = c * 0.02f;
    vec3 t = post::dither(post::sRGBToLinear(c), nz, 1.0f / 255.0f); printf("%s[", i ? "," : ""); p3(c); printf(","); p3(nz); printf(","); p3(t); printf("]"); }
  printf("],\n\"tone_exposure\": [");
  for(int i = 0; i < 32; i++) { post::tm.key = 0.1f + u01(s); post::tm.Ywhite = 0.3f + u01(s); vec3 c(u01(s) * 4, u01(s) * 4, u01(s) * 4); float la = 0.05f + u01(s) * 2;
    vec3 t = post::toneExposure(c, la); printf("%s[%u,%u,", i ? "," : "", fb(post::tm.key), fb(post::tm.Ywhite)); p3(c); printf(",%u,", fb(la)); p3(t); printf("]"); }
  printf("]\n}\n");
  return 0;
}
