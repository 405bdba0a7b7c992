// kat_main.cpp — mints known-answer vectors from the REFERENCE's own source files, compiled where they lie under
// /root/reference (authoring container only).  Nothing of the reference is copied into this repository: the GLSL
// excerpts are cut + qualifier-rewritten into a temp dir by build_ref.sh, this file only #includes them.
//   ref_random.inc   <- shaders/random.glsl:34-48 (tea), :59-65 (pcg), :98-102 (rand)
//   ref_common.inc   <- shaders/common.glsl:98-115 (OffsetRay), :141-143 (hash8bit)
//   compress.glsl    <- shaders/compress.glsl (its own `#ifdef __cplusplus` branch)
//   alias_table.hpp  <- src/alias_table.hpp
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
using std::min; using std::max;
typedef unsigned int uint;
struct vec3 { float x, y, z; vec3() : x(0), y(0), z(0) {} vec3(float a, float b, float c) : x(a), y(b), z(c) {} explicit vec3(float s) : x(s), y(s), z(s) {} };
struct vec4 { float x, y, z, w; vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {} };
struct ivec3 { int x, y, z; ivec3(int a, int b, int c) : x(a), y(b), z(c) {} ivec3(float a, float b, float c) : x(int(a)), y(int(b)), z(int(c)) {} };
static inline vec3 normalize(vec3 v) { float l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); return vec3(v.x / l, v.y / l, v.z / l); }
static inline float abs(float v) { return std::fabs(v); }
static inline bool isinf(float v) { return std::isinf(v); }
static inline int floatBitsToInt(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float intBitsToFloat(int i) { float f; memcpy(&f, &i, 4); return f; }
#include "compress.glsl"      // -I/root/reference/shaders : C++ branch defines uintBitsToFloat, floatBitsToUint, packUnorm4x8, roundEven
#include "ref_random.inc"
#include "ref_common.inc"
#include "alias_table.hpp"    // -I/root/reference/src

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
static float u01(uint32_t& s) { return float(lcg(s) >> 8) / 16777216.0f; }
static uint32_t fb(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main()
{
  uint32_t s = 12345u;
  printf("{\n");
  // ---- tea / pcg / rand ----
  printf("\"tea\": [");
  for(int i = 0; i < 64; i++) { uint32_t a = (i < 4) ? uint32_t(i) : lcg(s), b = (i < 4) ? uint32_t(3 - i) : lcg(s); printf("%s[%u,%u,%u]", i ? "," : "", a, b, tea(a, b)); }
  printf("],\n\"tea_1920x540_960_t12345\": %u,\n", tea(1920u * 540u + 960u, 12345u));
  printf("\"pcg\": [");
  for(int i = 0; i < 16; i++) { uint st = lcg(s); uint st0 = st; uint o = pcg(st); printf("%s[%u,%u,%u]", i ? "," : "", st0, st, o); }
  printf("],\n\"rand\": [");
  for(int i = 0; i < 16; i++) { uint st = lcg(s); uint st0 = st; float r = rand(st); printf("%s[%u,%u,%u]", i ? "," : "", st0, st, fb(r)); }
  printf("],\n");
  // ---- compress / decompress / packUnorm4x8 ----
  printf("\"compress_unit_vec\": [");
  for(int i = 0; i < 256; i++) {
    vec3 v(u01(s) * 2 - 1, u01(s) * 2 - 1, u01(s) * 2 - 1);
    if(i == 0) v = vec3(0, 0, 1); if(i == 1) v = vec3(0, 0, -1); if(i == 2) v = vec3(1, 0, 0); if(i == 3) v = vec3(0, -1, 0); if(i == 4) v = vec3(0.267f, 0.535f, 0.802f);
    v = normalize(v);
    uint c = compress_unit_vec(v);
    vec3 d = decompress_unit_vec(c);
    printf("%s[%u,%u,%u,%u,%u,%u,%u]", i ? "," : "", fb(v.x), fb(v.y), fb(v.z), c, fb(d.x), fb(d.y), fb(d.z));
  }
  printf("],\n\"pack_unorm4x8\": [");
  for(int i = 0; i < 64; i++) {
    vec4 v(u01(s) * 1.2f - 0.1f, u01(s) * 1.2f - 0.1f, u01(s), u01(s));
    if(i == 0) v = vec4(.1f, .5f, .9f, 1.f); if(i == 1) v = vec4(0.5f / 255.f, 1.5f / 255.f, 2.5f / 255.f, 254.5f / 255.f);
    printf("%s[%u,%u,%u,%u,%u]", i ? "," : "", fb(v.x), fb(v.y), fb(v.z), fb(v.w), packUnorm4x8(v));
  }
  printf("],\n");
  // ---- hash8bit / OffsetRay ----
  printf("\"hash8bit\": [");
  for(int i = 0; i < 32; i++) { uint a = (i < 8) ? uint(i * 37) : (lcg(s) >> (i % 20)); printf("%s[%u,%u]", i ? "," : "", a, hash8bit(a)); }
  printf("],\n\"offset_ray\": [");
  for(int i = 0; i < 64; i++) {
    float sc = (i % 3 == 0) ? 0.01f : ((i % 3 == 1) ? 1.f : 100.f);
    vec3 p((u01(s) * 2 - 1) * sc, (u01(s) * 2 - 1) * sc, (u01(s) * 2 - 1) * sc), n = normalize(vec3(u01(s) * 2 - 1, u01(s) * 2 - 1, u01(s) * 2 - 1));
    vec3 o = OffsetRay(p, n);
    printf("%s[%u,%u,%u,%u,%u,%u,%u,%u,%u]", i ? "," : "", fb(p.x), fb(p.y), fb(p.z), fb(n.x), fb(n.y), fb(n.z), fb(o.x), fb(o.y), fb(o.z));
  }
  printf("],\n");
  // ---- alias table (DiscreteSampler1D) ----
  printf("\"alias_table\": [");
  for(int t = 0; t < 12; t++) {
    int n = (t == 0) ? 4 : (t == 1 ? 1 : 2 + int(lcg(s) % 40));
    std::vector<float> w(n);
    for(int i = 0; i < n; i++) w[i] = (t == 0) ? (i == 3 ? 10.f : float(i + 1)) : ((t % 4 == 2 && i % 3 == 0) ? 0.f : u01(s) * (i % 5 == 0 ? 20.f : 1.f) + 0.001f);
    if(t == 5) for(int i = 0; i < n; i++) w[i] = 2.5f;  // uniform
    DiscreteSampler1D<float> ds(w);
    printf("%s{\"w\":[", t ? "," : "");
    for(int i = 0; i < n; i++) printf("%s%u", i ? "," : "", fb(w[i]));
    printf("],\"prob\":[");
    for(int i = 0; i < n; i++) printf("%s%u", i ? "," : "", fb(ds.binomDistribs[i].prob));
    printf("],\"fail\":[");
    for(int i = 0; i < n; i++) printf("%s%d", i ? "," : "", ds.binomDistribs[i].failId);
    printf("]}");
  }
  printf("]\n}\n");
  return 0;
}
