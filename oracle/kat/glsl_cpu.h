// glsl_cpu.h — a GLSL 4.60 compute-shader runtime for g++ (TEST INFRASTRUCTURE, authoring container only).
//
// Purpose: let the REFERENCE's own stage shaders (shaders/*.comp + everything they #include) compile and run on the CPU from
// where they lie under /root/reference, so that the oracle's stage-level restatement (oracle/orc_stages.cpp) can be held to the
// reference source itself instead of to a second reading of it.  glsl2cpp.py rewrites the GLSL text mechanically into a temp
// dir (rules listed there); this header supplies what a GLSL compiler + Vulkan driver supply:
//   * vector / matrix types and the built-in functions the shaders call,
//   * storage images, combined image samplers, SSBO / UBO / buffer_reference bindings as plain pointers,
//   * GL_EXT_ray_query as a brute-force candidate iterator over the flattened triangle list,
//   * the compute built-ins (gl_GlobalInvocationID ...), `shared`, barrier().
//
// What the GLSL spec leaves to the implementation is bound to this repository's numerics contract (include/rt_detmath.h and
// DESIGN.md §2) so that the output can be compared BIT FOR BIT with the oracle:
//   * transcendentals (exp/pow/sin/cos/tan/asin/acos/atan), float->int conversion, min/max NaN rule: rt_detmath.h;
//   * per-operation vector semantics: dot = products summed left to right, normalize(v) = v * (1/sqrt(dot(v,v))),
//     length = sqrt(dot), mix(a,b,t) = a*(1-t) + b*t, reflect(I,N) = I - (2*dot(N,I))*N, M*v = column products summed left to
//     right, inverse(mat3) = adjugate/determinant (first-row cofactor expansion), packUnorm4x8 rounds half away from zero,
//     roundEven = rintf;
//   * the DRIVER side (ray/triangle arithmetic, hit ordering, instance inverse transforms, texture filtering) is the oracle's
//     own orc::Scene — that part of the reference lives in the NVIDIA driver and stays PARITY UNPINNED (oracle/README.md).
// Nothing here is product code and nothing of the reference is stored in the repository.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>
#include "../orc_scene.h"

#undef INFINITY
#undef M_PI
#undef M_PI_2
#undef M_PI_4
#undef M_1_PI
#undef M_2_PI

namespace glsl {

typedef uint32_t uint;
template <class S> constexpr bool is_num = std::is_arithmetic_v<S>;
// GLSL implicit conversions: int -> uint, int/uint -> float
template <class From, class To> constexpr bool glsl_implicit =
    std::is_same_v<From, To> || (std::is_same_v<From, int> && (std::is_same_v<To, uint> || std::is_same_v<To, float>)) ||
    (std::is_same_v<From, uint> && std::is_same_v<To, float>);

// scalar conversions with the contract's float->int rule
template <class T, class S> inline T conv(S s)
{
  if constexpr(std::is_same_v<S, float> && std::is_same_v<T, int>) return rt_ftoi(s);
  else if constexpr(std::is_same_v<S, float> && std::is_same_v<T, uint>) return rt_ftou(s);
  else if constexpr(std::is_same_v<S, double> && std::is_same_v<T, int>) return rt_ftoi(float(s));
  else if constexpr(std::is_same_v<S, double> && std::is_same_v<T, uint>) return rt_ftou(float(s));
  else return T(s);
}
// `int(x)` / `uint(x)` of the GLSL text (glsl2cpp.py rewrites the functional casts to these)
template <class S> inline int glsl_int(S s) { return conv<int>(s); }
template <class S> inline uint glsl_uint(S s) { return conv<uint>(s); }

template <class T> struct tvec2;
template <class T> struct tvec3;
template <class T> struct tvec4;

template <class T> struct tvec2 {
  union { struct { T x, y; }; struct { T r, g; }; };
  tvec2() : x(0), y(0) {}
  template <class A, class = std::enable_if_t<is_num<A>>> explicit tvec2(A s) : x(conv<T>(s)), y(conv<T>(s)) {}
  template <class A, class B, class = std::enable_if_t<is_num<A> && is_num<B>>> tvec2(A a, B b) : x(conv<T>(a)), y(conv<T>(b)) {}
  template <class U> explicit(!glsl_implicit<U, T>) tvec2(const tvec2<U>& o) : x(conv<T>(o.x)), y(conv<T>(o.y)) {}
  tvec2<T> xy() const { return *this; }
};
template <class T> struct tvec3 {
  union { struct { T x, y, z; }; struct { T r, g, b; }; };
  tvec3() : x(0), y(0), z(0) {}
  template <class A, class = std::enable_if_t<is_num<A>>> explicit tvec3(A s) : x(conv<T>(s)), y(conv<T>(s)), z(conv<T>(s)) {}
  template <class A, class B, class C, class = std::enable_if_t<is_num<A> && is_num<B> && is_num<C>>>
  tvec3(A a, B b, C c) : x(conv<T>(a)), y(conv<T>(b)), z(conv<T>(c)) {}
  template <class C, class = std::enable_if_t<is_num<C>>> tvec3(const tvec2<T>& a, C c) : x(a.x), y(a.y), z(conv<T>(c)) {}
  template <class U> explicit(!glsl_implicit<U, T>) tvec3(const tvec3<U>& o) : x(conv<T>(o.x)), y(conv<T>(o.y)), z(conv<T>(o.z)) {}
  explicit tvec3(const tvec4<T>& o);  // vec3(vec4) drops w
  tvec2<T> xy() const { return tvec2<T>(x, y); }
  tvec3<T> xyz() const { return *this; }
  tvec3<T> rgb() const { return *this; }
};
template <class T> struct tvec4 {
  union { struct { T x, y, z, w; }; struct { T r, g, b, a; }; };
  tvec4() : x(0), y(0), z(0), w(0) {}
  template <class A, class = std::enable_if_t<is_num<A>>> explicit tvec4(A s) : x(conv<T>(s)), y(conv<T>(s)), z(conv<T>(s)), w(conv<T>(s)) {}
  template <class A, class B, class C, class D, class = std::enable_if_t<is_num<A> && is_num<B> && is_num<C> && is_num<D>>>
  tvec4(A a, B b, C c, D d) : x(conv<T>(a)), y(conv<T>(b)), z(conv<T>(c)), w(conv<T>(d)) {}
  template <class D, class = std::enable_if_t<is_num<D>>> tvec4(const tvec3<T>& v, D d) : x(v.x), y(v.y), z(v.z), w(conv<T>(d)) {}
  template <class C, class D, class = std::enable_if_t<is_num<C> && is_num<D>>> tvec4(const tvec2<T>& v, C c, D d) : x(v.x), y(v.y), z(conv<T>(c)), w(conv<T>(d)) {}
  template <class U> explicit(!glsl_implicit<U, T>) tvec4(const tvec4<U>& o) : x(conv<T>(o.x)), y(conv<T>(o.y)), z(conv<T>(o.z)), w(conv<T>(o.w)) {}
  tvec2<T> xy() const { return tvec2<T>(x, y); }
  tvec3<T> xyz() const { return tvec3<T>(x, y, z); }
  tvec3<T> rgb() const { return tvec3<T>(x, y, z); }
};
template <class T> inline tvec3<T>::tvec3(const tvec4<T>& o) : x(o.x), y(o.y), z(o.z) {}

typedef tvec2<float> vec2; typedef tvec3<float> vec3; typedef tvec4<float> vec4;
typedef tvec2<int> ivec2;  typedef tvec3<int> ivec3;  typedef tvec4<int> ivec4;
typedef tvec2<uint> uvec2; typedef tvec3<uint> uvec3; typedef tvec4<uint> uvec4;
static_assert(sizeof(vec2) == 8 && sizeof(vec3) == 12 && sizeof(vec4) == 16 && sizeof(uvec3) == 12, "scalar block layout");

// ---- component-wise operators: vec op vec, vec op scalar, scalar op vec (the scalar is converted to the vector's type) ----
#define GLSL_BINOP(OP)                                                                                                        \
  template <class T> inline tvec2<T> operator OP(const tvec2<T>& a, const tvec2<T>& b) { return tvec2<T>(a.x OP b.x, a.y OP b.y); }      \
  template <class T> inline tvec3<T> operator OP(const tvec3<T>& a, const tvec3<T>& b) { return tvec3<T>(a.x OP b.x, a.y OP b.y, a.z OP b.z); } \
  template <class T> inline tvec4<T> operator OP(const tvec4<T>& a, const tvec4<T>& b) { return tvec4<T>(a.x OP b.x, a.y OP b.y, a.z OP b.z, a.w OP b.w); } \
  template <class T, class S, class = std::enable_if_t<is_num<S>>> inline tvec2<T> operator OP(const tvec2<T>& a, S s) { T t = conv<T>(s); return tvec2<T>(a.x OP t, a.y OP t); } \
  template <class T, class S, class = std::enable_if_t<is_num<S>>> inline tvec3<T> operator OP(const tvec3<T>& a, S s) { T t = conv<T>(s); return tvec3<T>(a.x OP t, a.y OP t, a.z OP t); } \
  template <class T, class S, class = std::enable_if_t<is_num<S>>> inline tvec4<T> operator OP(const tvec4<T>& a, S s) { T t = conv<T>(s); return tvec4<T>(a.x OP t, a.y OP t, a.z OP t, a.w OP t); } \
  template <class T, class S, class = std::enable_if_t<is_num<S>>> inline tvec2<T> operator OP(S s, const tvec2<T>& a) { T t = conv<T>(s); return tvec2<T>(t OP a.x, t OP a.y); } \
  template <class T, class S, class = std::enable_if_t<is_num<S>>> inline tvec3<T> operator OP(S s, const tvec3<T>& a) { T t = conv<T>(s); return tvec3<T>(t OP a.x, t OP a.y, t OP a.z); } \
  template <class T, class S, class = std::enable_if_t<is_num<S>>> inline tvec4<T> operator OP(S s, const tvec4<T>& a) { T t = conv<T>(s); return tvec4<T>(t OP a.x, t OP a.y, t OP a.z, t OP a.w); }
GLSL_BINOP(+) GLSL_BINOP(-) GLSL_BINOP(*) GLSL_BINOP(/)
GLSL_BINOP(^) GLSL_BINOP(>>) GLSL_BINOP(<<) GLSL_BINOP(&) GLSL_BINOP(|)
#undef GLSL_BINOP
#define GLSL_ASSIGNOP(OP, BIN)                                                                                                \
  template <class T, class R> inline tvec2<T>& operator OP(tvec2<T>& a, const R& b) { a = a BIN b; return a; }                \
  template <class T, class R> inline tvec3<T>& operator OP(tvec3<T>& a, const R& b) { a = a BIN b; return a; }                \
  template <class T, class R> inline tvec4<T>& operator OP(tvec4<T>& a, const R& b) { a = a BIN b; return a; }
GLSL_ASSIGNOP(+=, +) GLSL_ASSIGNOP(-=, -) GLSL_ASSIGNOP(*=, *) GLSL_ASSIGNOP(/=, /) GLSL_ASSIGNOP(^=, ^)
#undef GLSL_ASSIGNOP
template <class T> inline tvec2<T> operator-(const tvec2<T>& a) { return tvec2<T>(-a.x, -a.y); }
template <class T> inline tvec3<T> operator-(const tvec3<T>& a) { return tvec3<T>(-a.x, -a.y, -a.z); }
template <class T> inline tvec4<T> operator-(const tvec4<T>& a) { return tvec4<T>(-a.x, -a.y, -a.z, -a.w); }
// implicit int -> float promotion between vectors (ivec2 * vec2 does not occur in the sources; vec2 op ivec2 neither)

// ---- scalar built-ins ----
inline float abs(float x) { return rt_abs(x); }
inline int abs(int x) { return x < 0 ? -x : x; }
inline float sqrt(float x) { return rt_sqrt(x); }
inline float exp(float x) { return rt_exp(x); }
inline float pow(float a, float b) { return rt_pow(a, b); }
inline float sin(float x) { return rt_sin(x); }
inline float cos(float x) { return rt_cos(x); }
inline float tan(float x) { return rt_tan(x); }
inline float asin(float x) { return rt_asin(x); }
inline float acos(float x) { return rt_acos(x); }
inline float atan(float y, float x) { return rt_atan2(y, x); }
inline float floor(float x) { return rt_floor(x); }
inline float roundEven(float x) { return rintf(x); }
inline float round(float x) { float r = truncf(x); if(rt_abs(x - r) >= 0.5f) r += (x < 0 ? -1.0f : 1.0f); return r; }
inline bool isnan(float x) { return rt_isnan(x); }
inline bool isinf(float x) { return rt_isinf(x); }
template <class A, class B, class = std::enable_if_t<is_num<A> && is_num<B>>> inline auto min(A a, B b)
{
  using R = std::conditional_t<std::is_floating_point_v<A> || std::is_floating_point_v<B>, float, std::common_type_t<A, B>>;
  R x = R(a), y = R(b); return (x < y) ? x : y;
}
template <class A, class B, class = std::enable_if_t<is_num<A> && is_num<B>>> inline auto max(A a, B b)
{
  using R = std::conditional_t<std::is_floating_point_v<A> || std::is_floating_point_v<B>, float, std::common_type_t<A, B>>;
  R x = R(a), y = R(b); return (x > y) ? x : y;
}
template <class A, class B, class C, class = std::enable_if_t<is_num<A> && is_num<B> && is_num<C>>> inline auto clamp(A x, B lo, C hi) { return min(max(x, lo), hi); }
inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline float step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
inline float smoothstep(float e0, float e1, float x) { float t = clamp((x - e0) / (e1 - e0), 0.0f, 1.0f); return t * t * (3.0f - 2.0f * t); }
inline uint floatBitsToUint(float f) { return rt_f2u(f); }
inline int floatBitsToInt(float f) { return rt_f2ibits(f); }
inline float uintBitsToFloat(uint u) { return rt_u2f(u); }
inline float intBitsToFloat(int i) { return rt_ibits2f(i); }

// ---- vector built-ins ----
inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float dot(vec4 a, vec4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float length(vec3 a) { return rt_sqrt(dot(a, a)); }
inline float length(vec2 a) { return rt_sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { float inv = 1.0f / rt_sqrt(dot(a, a)); return a * inv; }
inline vec2 normalize(vec2 a) { float inv = 1.0f / rt_sqrt(dot(a, a)); return a * inv; }
inline vec3 mix(vec3 a, vec3 b, float t) { return a * (1.0f - t) + b * t; }
inline vec4 mix(vec4 a, vec4 b, float t) { return a * (1.0f - t) + b * t; }
inline vec3 mix(vec3 a, vec3 b, vec3 t) { return vec3(mix(a.x, b.x, t.x), mix(a.y, b.y, t.y), mix(a.z, b.z, t.z)); }
inline vec3 reflect(vec3 I, vec3 N) { return I - (2.0f * dot(N, I)) * N; }
#define GLSL_MAP1(F)                                                                                  \
  inline vec2 F(vec2 v) { return vec2(F(v.x), F(v.y)); }                                              \
  inline vec3 F(vec3 v) { return vec3(F(v.x), F(v.y), F(v.z)); }                                      \
  inline vec4 F(vec4 v) { return vec4(F(v.x), F(v.y), F(v.z), F(v.w)); }
GLSL_MAP1(abs) GLSL_MAP1(sqrt) GLSL_MAP1(exp) GLSL_MAP1(sin) GLSL_MAP1(cos) GLSL_MAP1(floor)
#undef GLSL_MAP1
inline vec3 pow(vec3 a, vec3 b) { return vec3(pow(a.x, b.x), pow(a.y, b.y), pow(a.z, b.z)); }
inline vec3 max(vec3 a, vec3 b) { return vec3(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
inline vec3 min(vec3 a, vec3 b) { return vec3(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)); }
inline vec3 max(vec3 a, float b) { return vec3(max(a.x, b), max(a.y, b), max(a.z, b)); }
inline vec3 min(vec3 a, float b) { return vec3(min(a.x, b), min(a.y, b), min(a.z, b)); }
inline vec3 clamp(vec3 v, float lo, float hi) { return vec3(clamp(v.x, lo, hi), clamp(v.y, lo, hi), clamp(v.z, lo, hi)); }
inline vec3 step(vec3 e, vec3 x) { return vec3(step(e.x, x.x), step(e.y, x.y), step(e.z, x.z)); }
// packUnorm4x8: round(clamp(c, 0, 1) * 255), round half away from zero (the contract's choice; GLSL leaves x.5 open)
inline uint packUnorm4x8(vec4 v)
{
  auto q = [](float c) { float s = clamp(c, 0.0f, 1.0f) * 255.0f; float r = truncf(s); if(s - r >= 0.5f) r += 1.0f; return uint(r); };
  return q(v.x) | (q(v.y) << 8) | (q(v.z) << 16) | (q(v.w) << 24);
}
inline vec4 unpackUnorm4x8(uint p) { return vec4(float(p & 0xffu) / 255.0f, float((p >> 8) & 0xffu) / 255.0f, float((p >> 16) & 0xffu) / 255.0f, float(p >> 24) / 255.0f); }

// ---- matrices (column vectors) ----
struct mat4x3;
struct mat3 {
  vec3 c[3];
  mat3() {}
  mat3(vec3 a, vec3 b, vec3 d) { c[0] = a; c[1] = b; c[2] = d; }
  mat3(float a, float b, float d, float e, float f, float g, float h, float i, float j) { c[0] = vec3(a, b, d); c[1] = vec3(e, f, g); c[2] = vec3(h, i, j); }
  vec3& operator[](int i) { return c[i]; }
  const vec3& operator[](int i) const { return c[i]; }
};
inline vec3 operator*(const mat3& M, vec3 v) { return (M.c[0] * v.x + M.c[1] * v.y) + M.c[2] * v.z; }
inline vec3 operator*(vec3 v, const mat3& M) { return vec3(dot(v, M.c[0]), dot(v, M.c[1]), dot(v, M.c[2])); }
inline mat3 inverse(const mat3& M)
{
  float a = M.c[0].x, b = M.c[1].x, c = M.c[2].x;
  float d = M.c[0].y, e = M.c[1].y, f = M.c[2].y;
  float g = M.c[0].z, h = M.c[1].z, i = M.c[2].z;
  float A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;
  float det = (a * A + b * B) + c * C;
  float inv = 1.0f / det;
  mat3 R;
  R.c[0] = vec3(A * inv, B * inv, C * inv);
  R.c[1] = vec3((c * h - b * i) * inv, (a * i - c * g) * inv, (b * g - a * h) * inv);
  R.c[2] = vec3((b * f - c * e) * inv, (c * d - a * f) * inv, (a * e - b * d) * inv);
  return R;
}
struct mat4 {
  vec4 c[4];
  mat4() {}
  explicit mat4(const mat4x3& m);
  vec4& operator[](int i) { return c[i]; }
  const vec4& operator[](int i) const { return c[i]; }
};
inline vec4 operator*(const mat4& M, vec4 v) { return ((M.c[0] * v.x + M.c[1] * v.y) + M.c[2] * v.z) + M.c[3] * v.w; }
struct mat4x3 {  // 4 columns x 3 rows
  vec3 c[4];
  mat4x3() {}
  vec3& operator[](int i) { return c[i]; }
  const vec3& operator[](int i) const { return c[i]; }
};
inline mat4::mat4(const mat4x3& m) { for(int i = 0; i < 4; i++) c[i] = vec4(m.c[i], i == 3 ? 1.0f : 0.0f); }
inline vec3 operator*(const mat4x3& M, vec4 v) { return ((M.c[0] * v.x + M.c[1] * v.y) + M.c[2] * v.z) + M.c[3] * v.w; }
inline vec4 operator*(vec3 v, const mat4x3& M) { return vec4(dot(v, M.c[0]), dot(v, M.c[1]), dot(v, M.c[2]), dot(v, M.c[3])); }
static_assert(sizeof(mat4) == 64, "mat4 is 16 floats, column major");

// ---- storage images (Vulkan robust access: out-of-bounds loads return 0, stores are dropped) ----
template <class T, int NC> struct timage {   // NC = stored components per texel
  T* data = nullptr; int w = 0, h = 0;
};
typedef timage<float, 4> image2D;      // RGBA32F
typedef timage<uint, 4> uimage2D;      // RGBA32UI
typedef timage<int16_t, 2> iimage2D;   // RG16_SINT
inline vec4 imageLoad(const image2D& im, ivec2 p)
{
  if(p.x < 0 || p.y < 0 || p.x >= im.w || p.y >= im.h) return vec4();
  const float* q = im.data + (size_t(p.y) * im.w + p.x) * 4; return vec4(q[0], q[1], q[2], q[3]);
}
inline uvec4 imageLoad(const uimage2D& im, ivec2 p)
{
  if(p.x < 0 || p.y < 0 || p.x >= im.w || p.y >= im.h) return uvec4();
  const uint* q = im.data + (size_t(p.y) * im.w + p.x) * 4; return uvec4(q[0], q[1], q[2], q[3]);
}
inline ivec4 imageLoad(const iimage2D& im, ivec2 p)
{
  if(p.x < 0 || p.y < 0 || p.x >= im.w || p.y >= im.h) return ivec4();
  const int16_t* q = im.data + (size_t(p.y) * im.w + p.x) * 2; return ivec4(q[0], q[1], 0, 1);
}
inline void imageStore(const image2D& im, ivec2 p, vec4 v)
{
  if(p.x < 0 || p.y < 0 || p.x >= im.w || p.y >= im.h) return;
  float* q = im.data + (size_t(p.y) * im.w + p.x) * 4; q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
}
inline void imageStore(const uimage2D& im, ivec2 p, uvec4 v)
{
  if(p.x < 0 || p.y < 0 || p.x >= im.w || p.y >= im.h) return;
  uint* q = im.data + (size_t(p.y) * im.w + p.x) * 4; q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
}
inline void imageStore(const iimage2D& im, ivec2 p, ivec4 v)   // RG16_SINT: the store saturates (Vulkan format conversion)
{
  if(p.x < 0 || p.y < 0 || p.x >= im.w || p.y >= im.h) return;
  auto sat = [](int c) { return int16_t(c < -32768 ? -32768 : (c > 32767 ? 32767 : c)); };
  int16_t* q = im.data + (size_t(p.y) * im.w + p.x) * 2; q[0] = sat(v.x); q[1] = sat(v.y);
}

// ---- combined image samplers: the oracle's Vulkan-sampler restatement (driver side, unpinned) ----
struct sampler2D { const orc::Scene* scene = nullptr; int tex = -1; /* -1 = the environment map */ };
inline vec4 texture(const sampler2D& s, vec2 uv)
{
  orc::vec4 r = (s.tex < 0) ? s.scene->sampleEnv(orc::V2(uv.x, uv.y)) : s.scene->sampleTexture(s.tex, orc::V2(uv.x, uv.y));
  return vec4(r.x, r.y, r.z, r.w);
}
template <class L> inline vec4 textureLod(const sampler2D& s, vec2 uv, L) { return texture(s, uv); }
inline ivec2 textureSize(const sampler2D& s, int) { return (s.tex < 0) ? ivec2(s.scene->envW, s.scene->envH) : ivec2(s.scene->textures[s.tex].w, s.scene->textures[s.tex].h); }
#define nonuniformEXT(x) (x)

// ---- GL_EXT_ray_query: brute-force candidate iterator over the oracle's flattened triangle list ----
struct accelerationStructureEXT { const orc::Scene* scene = nullptr; };
const uint gl_RayFlagsNoneEXT = 0u, gl_RayFlagsOpaqueEXT = 1u, gl_RayFlagsNoOpaqueEXT = 2u, gl_RayFlagsTerminateOnFirstHitEXT = 4u,
           gl_RayFlagsSkipClosestHitShaderEXT = 8u, gl_RayFlagsCullBackFacingTrianglesEXT = 16u, gl_RayFlagsCullFrontFacingTrianglesEXT = 32u;
const uint gl_RayQueryCommittedIntersectionNoneEXT = 0u, gl_RayQueryCommittedIntersectionTriangleEXT = 1u;
const uint gl_RayQueryCandidateIntersectionTriangleEXT = 0u, gl_RayQueryCandidateIntersectionAABBEXT = 1u;
// The reference's HitTest draws rand(prd.seed) for every non-opaque candidate, in the driver's candidate order.  This
// repository replaces that draw by one keyed on (ray seed, triangle) — DESIGN.md §6 deviation 1 — and the stand-in driver
// applies the same rule: while a candidate is exposed to the shader, *rq_seed holds the candidate's own seed, and the
// ray's seed is restored afterwards (so traversal does not advance prd.seed).  rq_seed is set by the harness to &prd.seed.
extern thread_local uint* rq_seed;
uint rq_candidate_seed(uint raySeed, uint tri);   // ref_frame.cpp: the contract's key
struct rayQueryEXT {
  const orc::Scene* scene = nullptr;
  orc::vec3 o, d; float tmin = 0, tmax = 0; uint flags = 0;
  uint next = 0;                 // next triangle to look at
  bool done = false, swapped = false; uint savedSeed = 0;
  struct Rec { float t = 0, u = 0, v = 0; uint tri = 0xffffffffu; } cand, committed;
  bool haveCommitted = false;
};
inline void rq_restore(rayQueryEXT& q) { if(q.swapped) { *rq_seed = q.savedSeed; q.swapped = false; } }
inline void rayQueryInitializeEXT(rayQueryEXT& q, const accelerationStructureEXT& as, uint flags, uint /*cullMask*/, vec3 o, float tmin, vec3 d, float tmax)
{
  q = rayQueryEXT(); q.scene = as.scene; q.o = orc::V3(o.x, o.y, o.z); q.d = orc::V3(d.x, d.y, d.z); q.tmin = tmin; q.tmax = tmax; q.flags = flags;
  // a ray with a NaN component or an empty interval finds nothing (same rule as orc::Scene::closestHit / anyHit)
  if(orc::hasNan(q.o) || orc::hasNan(q.d) || !(tmax > 0.0f)) q.done = true;
}
inline bool rq_closer(const rayQueryEXT& q, float t, uint tri)
{
  if(!q.haveCommitted) return t < q.tmax;
  return t < q.committed.t || (t == q.committed.t && tri < q.committed.tri);   // ties: lowest flattened triangle index
}
inline bool rayQueryProceedEXT(rayQueryEXT& q)
{
  rq_restore(q);
  if(q.done) return false;
  const orc::Scene& S = *q.scene;
  const bool terminateOnFirst = (q.flags & gl_RayFlagsTerminateOnFirstHitEXT) != 0;
  while(q.next < S.tris.size()) {
    if(terminateOnFirst && q.haveCommitted) break;
    const uint ti = q.next++;
    const orc::Tri& T = S.tris[ti];
    float t, u, v;
    if(!S.intersectTri(T, q.o, q.d, t, u, v)) continue;   // includes back-face culling per instance flags
    if(!(t > q.tmin)) continue;
    if(!rq_closer(q, t, ti)) continue;
    if(T.flags & orc::TRI_OPAQUE) { q.committed = {t, u, v, ti}; q.haveCommitted = true; continue; }
    q.cand = {t, u, v, ti};
    if(rq_seed) { q.savedSeed = *rq_seed; *rq_seed = rq_candidate_seed(q.savedSeed, ti); q.swapped = true; }
    return true;   // non-opaque candidate: the shader decides (HitTest)
  }
  q.done = true;
  return false;
}
inline void rayQueryConfirmIntersectionEXT(rayQueryEXT& q) { q.committed = q.cand; q.haveCommitted = true; }
inline uint rayQueryGetIntersectionTypeEXT(const rayQueryEXT& q, bool committed)
{
  if(committed) return q.haveCommitted ? gl_RayQueryCommittedIntersectionTriangleEXT : gl_RayQueryCommittedIntersectionNoneEXT;
  return gl_RayQueryCandidateIntersectionTriangleEXT;
}
inline const rayQueryEXT::Rec& rq_rec(const rayQueryEXT& q, bool committed) { return committed ? q.committed : q.cand; }
inline float rayQueryGetIntersectionTEXT(const rayQueryEXT& q, bool c) { return rq_rec(q, c).t; }
inline int rayQueryGetIntersectionPrimitiveIndexEXT(const rayQueryEXT& q, bool c) { return int(q.scene->tris[rq_rec(q, c).tri].prim); }
inline int rayQueryGetIntersectionInstanceIdEXT(const rayQueryEXT& q, bool c) { return int(q.scene->tris[rq_rec(q, c).tri].inst); }
inline int rayQueryGetIntersectionInstanceCustomIndexEXT(const rayQueryEXT& q, bool c) { return int(q.scene->instances[q.scene->tris[rq_rec(q, c).tri].inst].primMesh); }
inline vec2 rayQueryGetIntersectionBarycentricsEXT(const rayQueryEXT& q, bool c) { return vec2(rq_rec(q, c).u, rq_rec(q, c).v); }
inline mat4x3 rq_affine(const orc::affine& A) { mat4x3 M; for(int c = 0; c < 4; c++) M.c[c] = vec3(A.a[c], A.a[4 + c], A.a[8 + c]); return M; }
inline mat4x3 rayQueryGetIntersectionObjectToWorldEXT(const rayQueryEXT& q, bool c) { return rq_affine(q.scene->objectToWorld[q.scene->tris[rq_rec(q, c).tri].inst]); }
inline mat4x3 rayQueryGetIntersectionWorldToObjectEXT(const rayQueryEXT& q, bool c) { return rq_affine(q.scene->worldToObject[q.scene->tris[rq_rec(q, c).tri].inst]); }

// ---- compute built-ins ----
extern thread_local uvec3 gl_GlobalInvocationID, gl_LocalInvocationID, gl_WorkGroupID;
extern thread_local uint gl_LocalInvocationIndex;
inline void barrier() {}   // invocations of a workgroup run to completion one after another, in gl_LocalInvocationIndex order
#define shared static

}  // namespace glsl
