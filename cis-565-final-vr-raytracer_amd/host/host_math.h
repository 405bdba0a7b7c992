// host_math.h — minimal vec/mat helpers for the host side (nvmath stand-in; nvpro_core is not vendored by the
// reference, CMakeLists.txt:29-44).  Only camera/scene preparation uses these: results are INPUTS of the hot
// path (rt_scene_camera, instance transforms), so they carry no bit-exactness contract.
#pragma once
#include <cmath>
#include <cstring>
#include "../../include/rt_abi.h"

namespace rth {

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { float l = length(a); return l > 0 ? a * (1.0f / l) : a; }
inline rt_vec3 R3(V3 v) { return rt_vec3{v.x, v.y, v.z}; }

// column-major 4x4, m[c*4+r] (nvmath::mat4f memory layout)
struct M4 {
  float m[16];
  float& at(int r, int c) { return m[c * 4 + r]; }
  float at(int r, int c) const { return m[c * 4 + r]; }
  static M4 identity() { M4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1; return r; }
};
inline M4 operator*(const M4& a, const M4& b)
{
  M4 r{};
  for(int c = 0; c < 4; c++)
    for(int rr = 0; rr < 4; rr++) {
      double s = 0;
      for(int k = 0; k < 4; k++) s += double(a.at(rr, k)) * double(b.at(k, c));
      r.at(rr, c) = float(s);
    }
  return r;
}
inline V3 xformPoint(const M4& m, V3 p)
{
  return {m.at(0, 0) * p.x + m.at(0, 1) * p.y + m.at(0, 2) * p.z + m.at(0, 3), m.at(1, 0) * p.x + m.at(1, 1) * p.y + m.at(1, 2) * p.z + m.at(1, 3),
          m.at(2, 0) * p.x + m.at(2, 1) * p.y + m.at(2, 2) * p.z + m.at(2, 3)};
}
inline V3 xformDir(const M4& m, V3 p)
{
  return {m.at(0, 0) * p.x + m.at(0, 1) * p.y + m.at(0, 2) * p.z, m.at(1, 0) * p.x + m.at(1, 1) * p.y + m.at(1, 2) * p.z,
          m.at(2, 0) * p.x + m.at(2, 1) * p.y + m.at(2, 2) * p.z};
}
// general 4x4 inverse (cofactors, double accumulation) — nvmath::invert stand-in
inline M4 invert(const M4& a)
{
  double inv[16], m[16];
  for(int i = 0; i < 16; i++) m[i] = a.m[i];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  M4 r{};
  if(det == 0) return r;
  det = 1.0 / det;
  for(int i = 0; i < 16; i++) r.m[i] = float(inv[i] * det);
  return r;
}
// right-handed look-at view matrix (nvh::CameraManip::getMatrix stand-in)
inline M4 lookAt(V3 eye, V3 center, V3 up)
{
  V3 f = normalize(center - eye), s = normalize(cross(f, up)), u = cross(s, f);
  M4 r = M4::identity();
  r.at(0, 0) = s.x; r.at(0, 1) = s.y; r.at(0, 2) = s.z; r.at(0, 3) = -dot(s, eye);
  r.at(1, 0) = u.x; r.at(1, 1) = u.y; r.at(1, 2) = u.z; r.at(1, 3) = -dot(u, eye);
  r.at(2, 0) = -f.x; r.at(2, 1) = -f.y; r.at(2, 2) = -f.z; r.at(2, 3) = dot(f, eye);
  return r;
}
// Vulkan-convention perspective: depth in [0,1], Y flipped (nvmath::perspectiveVK stand-in; scene.cpp:785)
inline M4 perspectiveVK(float fovyDeg, float aspect, float n, float f)
{
  M4 r{};
  float t = n * std::tan(fovyDeg * 3.14159265358979323846f / 180.0f * 0.5f);
  float b = -t, l = b * aspect, rr = t * aspect;
  r.at(0, 0) = (2 * n) / (rr - l);
  r.at(1, 1) = -(2 * n) / (t - b);
  r.at(0, 2) = (rr + l) / (rr - l);
  r.at(1, 2) = (t + b) / (t - b);
  r.at(2, 2) = -f / (f - n);
  r.at(3, 2) = -1;
  r.at(2, 3) = (f * n) / (n - f);
  return r;
}
inline rt_mat4 toRt(const M4& m) { rt_mat4 r; memcpy(r.m, m.m, sizeof(r.m)); return r; }
inline float luminance(const float* c) { return c[0] * 0.2126f + c[1] * 0.7152f + c[2] * 0.0722f; }  // tools.hpp:57-61

}  // namespace rth
