// orc_math.h — GLSL-style vector types for the CPU oracle (TEST INFRASTRUCTURE, not product code).
//
// The oracle restates the reference's GLSL in scalar C++.  GLSL leaves evaluation order of vector
// expressions to the compiler; here every operation is fixed (component-wise, sums left-to-right) so
// that the result is a function of the inputs only.  The product's HIP kernels are written
// independently but obey the same per-operation definitions, which are listed in DESIGN.md §Numerics.
#pragma once
#include <cstdint>
#include <cstring>
#include "../include/rt_detmath.h"

namespace orc {

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct ivec2 { int x, y; };
struct uvec4 { uint32_t x, y, z, w; };

inline vec2 V2(float x, float y) { return vec2{x, y}; }
inline vec3 V3(float x, float y, float z) { return vec3{x, y, z}; }
inline vec3 V3(float s) { return vec3{s, s, s}; }
inline vec4 V4(float x, float y, float z, float w) { return vec4{x, y, z, w}; }
inline vec4 V4(vec3 v, float w) { return vec4{v.x, v.y, v.z, w}; }

inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline vec2 operator*(vec2 a, float s) { return {a.x * s, a.y * s}; }
inline vec2 operator*(float s, vec2 a) { return {s * a.x, s * a.y}; }
inline vec2 operator*(vec2 a, vec2 b) { return {a.x * b.x, a.y * b.y}; }
inline vec2 operator/(vec2 a, vec2 b) { return {a.x / b.x, a.y / b.y}; }
inline vec2 operator+(vec2 a, float s) { return {a.x + s, a.y + s}; }
inline vec2 operator-(vec2 a, float s) { return {a.x - s, a.y - s}; }

inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3 operator+(vec3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline vec3 operator-(float s, vec3 a) { return {s - a.x, s - a.y, s - a.z}; }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
inline vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }

inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline vec4 operator*(vec4 a, vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }

inline vec3 xyz(vec4 v) { return {v.x, v.y, v.z}; }

// dot: products summed left to right
inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(vec3 a) { return rt_sqrt(dot(a, a)); }
// normalize(v) := v * (1 / sqrt(dot(v,v)))
inline vec3 normalize(vec3 a) { float inv = 1.0f / rt_sqrt(dot(a, a)); return a * inv; }
// mix(a,b,t) := a*(1-t) + b*t   (GLSL definition)
inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline vec3 mix(vec3 a, vec3 b, float t) { return a * (1.0f - t) + b * t; }
inline vec3 mix(vec3 a, vec3 b, vec3 t) { return {mix(a.x, b.x, t.x), mix(a.y, b.y, t.y), mix(a.z, b.z, t.z)}; }
inline vec4 mix(vec4 a, vec4 b, float t) { return a * (1.0f - t) + b * t; }
// reflect(I,N) := I - 2*dot(N,I)*N
inline vec3 reflect(vec3 I, vec3 N) { return I - (2.0f * dot(N, I)) * N; }
inline float luminance(vec3 c) { return (0.2126f * c.x + 0.7152f * c.y) + 0.0722f * c.z; }  // sun_and_sky.glsl:31-34
inline bool hasNan(vec3 v) { return rt_isnan(v.x) || rt_isnan(v.y) || rt_isnan(v.z); }     // common.glsl:181-183

// Column-major 4x4 (nvmath::mat4f / GLSL mat4): m[c*4+r]
struct mat4 { float m[16]; };
// M * v : r_i = ((M_i0*v.x + M_i1*v.y) + M_i2*v.z) + M_i3*v.w
inline vec4 mul(const mat4& M, vec4 v)
{
  vec4 r;
  r.x = ((M.m[0] * v.x + M.m[4] * v.y) + M.m[8] * v.z) + M.m[12] * v.w;
  r.y = ((M.m[1] * v.x + M.m[5] * v.y) + M.m[9] * v.z) + M.m[13] * v.w;
  r.z = ((M.m[2] * v.x + M.m[6] * v.y) + M.m[10] * v.z) + M.m[14] * v.w;
  r.w = ((M.m[3] * v.x + M.m[7] * v.y) + M.m[11] * v.z) + M.m[15] * v.w;
  return r;
}

// 3x3, column vectors c0,c1,c2 (GLSL mat3(c0,c1,c2))
struct mat3 { vec3 c0, c1, c2; };
inline vec3 mul(const mat3& M, vec3 v) { return (M.c0 * v.x + M.c1 * v.y) + M.c2 * v.z; }
// inverse(mat3) := adjugate / determinant, determinant expanded along the first column
inline mat3 inverse(const mat3& M)
{
  float a = M.c0.x, b = M.c1.x, c = M.c2.x;
  float d = M.c0.y, e = M.c1.y, f = M.c2.y;
  float g = M.c0.z, h = M.c1.z, i = M.c2.z;
  float A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;
  float det = (a * A + b * B) + c * C;
  float inv = 1.0f / det;
  mat3 R;
  R.c0 = V3(A * inv, B * inv, C * inv);
  R.c1 = V3((c * h - b * i) * inv, (a * i - c * g) * inv, (b * g - a * h) * inv);
  R.c2 = V3((b * f - c * e) * inv, (c * d - a * f) * inv, (a * e - b * d) * inv);
  return R;
}

// 3x4 affine (rows), as VkTransformMatrixKHR / GLSL mat4x3 transposed: a[r*4+c]
struct affine { float a[12]; };
// objectToWorld * vec4(p,1)
inline vec3 xformPoint(const affine& M, vec3 p)
{
  return {((M.a[0] * p.x + M.a[1] * p.y) + M.a[2] * p.z) + M.a[3],
          ((M.a[4] * p.x + M.a[5] * p.y) + M.a[6] * p.z) + M.a[7],
          ((M.a[8] * p.x + M.a[9] * p.y) + M.a[10] * p.z) + M.a[11]};
}
// mat4(objectToWorld) * vec4(v,0)
inline vec3 xformDir(const affine& M, vec3 v)
{
  return {(M.a[0] * v.x + M.a[1] * v.y) + M.a[2] * v.z, (M.a[4] * v.x + M.a[5] * v.y) + M.a[6] * v.z,
          (M.a[8] * v.x + M.a[9] * v.y) + M.a[10] * v.z};
}
// vec3(v * worldToObject)  (row vector times mat4x3 => transpose(W3x3) * v)
inline vec3 xformNormal(const affine& W, vec3 v)
{
  return {(v.x * W.a[0] + v.y * W.a[4]) + v.z * W.a[8], (v.x * W.a[1] + v.y * W.a[5]) + v.z * W.a[9],
          (v.x * W.a[2] + v.y * W.a[6]) + v.z * W.a[10]};
}
// inverse of an affine 3x4: adjugate/determinant for the 3x3 block, then -inv3 * t
inline affine inverseAffine(const affine& M, float* detOut = nullptr)
{
  float a = M.a[0], b = M.a[1], c = M.a[2];
  float d = M.a[4], e = M.a[5], f = M.a[6];
  float g = M.a[8], h = M.a[9], i = M.a[10];
  float A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;
  float det = (a * A + b * B) + c * C;
  if(detOut) *detOut = det;
  float inv = 1.0f / det;
  affine R;
  R.a[0] = A * inv;               R.a[1] = (c * h - b * i) * inv; R.a[2] = (b * f - c * e) * inv;
  R.a[4] = B * inv;               R.a[5] = (a * i - c * g) * inv; R.a[6] = (c * d - a * f) * inv;
  R.a[8] = C * inv;               R.a[9] = (b * g - a * h) * inv; R.a[10] = (a * e - b * d) * inv;
  float tx = M.a[3], ty = M.a[7], tz = M.a[11];
  R.a[3]  = -((R.a[0] * tx + R.a[1] * ty) + R.a[2] * tz);
  R.a[7]  = -((R.a[4] * tx + R.a[5] * ty) + R.a[6] * tz);
  R.a[11] = -((R.a[8] * tx + R.a[9] * ty) + R.a[10] * tz);
  return R;
}

}  // namespace orc
