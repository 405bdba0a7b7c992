// pack.h — host-side vertex attribute packing used by Scene::createVertexBuffer.
// The reference packs on the host with the C++ branch of shaders/compress.glsl (:31-139): octahedral 16+16-bit
// unit vectors with round-half-even, and unorm8x4 colours with std::round.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace rth {

inline uint32_t floatBits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float bitsFloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

inline uint32_t packUnorm4x8(float r, float g, float b, float a)
{
  auto q = [](float v) { v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v); return uint32_t((unsigned char)std::round(v * 255.f)); };
  return q(r) | (q(g) << 8) | (q(b) << 16) | (q(a) << 24);
}

// octahedral encode: project on |x|+|y|+|z| = 1 scaled to 32767, fold the lower hemisphere, bias to unsigned
inline uint32_t compressUnitVec(float nx, float ny, float nz)
{
  if(!(nx < 3.402823466e+38f) || std::isinf(nx)) return ~0u;
  const float d = 32767.0f / (std::fabs(nx) + std::fabs(ny) + std::fabs(nz));
  int x = int(std::nearbyint(nx * d));  // default rounding mode = round-half-even
  int y = int(std::nearbyint(ny * d));
  if(nz < 0.0f) {
    const int mx = x >> 31, my = y >> 31;
    const int t = 32767 + mx + my;
    const int ox = x;
    x = (t - (y ^ my)) ^ mx;
    y = (t - (ox ^ mx)) ^ my;
  }
  uint32_t packed = (uint32_t(y + 32767) << 16) | uint32_t(x + 32767);
  return packed == ~0u ? ~0x1u : packed;
}

}  // namespace rth
