#include "hdr_sampling.hpp"
#include "host_math.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>

namespace rth {

// ---- Radiance RGBE reader (stands in for stbi_loadf, hdr_sampling.cpp:64) --------------------------------
static bool readLine(FILE* f, std::string& s)
{
  s.clear();
  int c;
  while((c = fgetc(f)) != EOF) { if(c == '\n') return true; s.push_back(char(c)); }
  return !s.empty();
}
static void rgbe2float(const unsigned char* p, float* out)
{
  if(p[3] == 0) { out[0] = out[1] = out[2] = 0.f; }
  else {
    float f = std::ldexp(1.0f, int(p[3]) - (128 + 8));
    out[0] = p[0] * f; out[1] = p[1] * f; out[2] = p[2] * f;
  }
  out[3] = 1.f;
}
bool HdrSampling::loadEnvironment(const std::string& path)
{
  FILE* f = fopen(path.c_str(), "rb");
  if(!f) return false;
  std::string line;
  if(!readLine(f, line) || (line.rfind("#?RADIANCE", 0) != 0 && line.rfind("#?RGBE", 0) != 0)) { fclose(f); return false; }
  bool fmtOk = false;
  while(readLine(f, line) && !line.empty()) if(line.find("FORMAT=32-bit_rle_rgbe") != std::string::npos) fmtOk = true;
  int w = 0, h = 0;
  if(!fmtOk || !readLine(f, line) || sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) { fclose(f); return false; }
  std::vector<float> px(size_t(w) * h * 4);
  std::vector<unsigned char> scan(size_t(w) * 4);
  for(int y = 0; y < h; y++) {
    unsigned char hd[4];
    if(fread(hd, 1, 4, f) != 4) { fclose(f); return false; }
    if(w >= 8 && w < 32768 && hd[0] == 2 && hd[1] == 2 && ((hd[2] << 8) | hd[3]) == w) {
      for(int ch = 0; ch < 4; ch++) {  // new-style RLE, one channel at a time
        int x = 0;
        while(x < w) {
          int cnt = fgetc(f);
          if(cnt == EOF) { fclose(f); return false; }
          if(cnt > 128) { cnt -= 128; int v = fgetc(f); if(x + cnt > w) { fclose(f); return false; } for(int k = 0; k < cnt; k++) scan[size_t(x++) * 4 + ch] = (unsigned char)v; }
          else { if(cnt == 0 || x + cnt > w) { fclose(f); return false; } for(int k = 0; k < cnt; k++) scan[size_t(x++) * 4 + ch] = (unsigned char)fgetc(f); }
        }
      }
    } else {  // flat
      memcpy(scan.data(), hd, 4);
      if(fread(scan.data() + 4, 1, size_t(w - 1) * 4, f) != size_t(w - 1) * 4) { fclose(f); return false; }
    }
    for(int x = 0; x < w; x++) rgbe2float(&scan[size_t(x) * 4], &px[(size_t(y) * w + x) * 4]);
  }
  fclose(f);
  setEnvironment(px.data(), w, h);
  return true;
}

void HdrSampling::setEnvironment(const float* rgba, int w, int h)
{
  m_w = w; m_h = h;
  m_pixels.assign(rgba, rgba + size_t(w) * h * 4);
  createEnvironmentAccel();
}

void HdrSampling::makeSyntheticSky(int w, int h, float sunPeak, uint32_t seed)
{
  std::vector<float> px(size_t(w) * h * 4);
  // sun direction fixed in the upper hemisphere; small seed-dependent azimuth so fixtures differ by seed
  const float sunTheta = 0.9f, sunPhi = 0.6f + 0.37f * float(seed % 17);
  const V3 sun{std::sin(sunTheta) * std::cos(sunPhi), std::cos(sunTheta), std::sin(sunTheta) * std::sin(sunPhi)};
  for(int y = 0; y < h; y++) {
    float theta = (y + 0.5f) / h * 3.14159265f;
    for(int x = 0; x < w; x++) {
      float phi = (x + 0.5f) / w * 6.2831853f - 3.14159265f;
      V3 d{std::cos(phi) * std::sin(theta), std::cos(theta), std::sin(phi) * std::sin(theta)};
      float up = std::max(d.y, 0.f);
      float sky[3] = {0.25f + 0.35f * (1 - up), 0.45f + 0.35f * (1 - up), 0.9f - 0.1f * (1 - up)};
      if(d.y < 0) { sky[0] = 0.18f; sky[1] = 0.16f; sky[2] = 0.14f; }
      float c = std::max(dot(d, sun), 0.f);
      float lobe = sunPeak * std::pow(c, 2048.f) + 4.f * std::pow(c, 32.f);
      float* p = &px[(size_t(y) * w + x) * 4];
      p[0] = sky[0] + lobe; p[1] = sky[1] + lobe * 0.95f; p[2] = sky[2] + lobe * 0.85f; p[3] = 1.f;
    }
  }
  setEnvironment(px.data(), w, h);
}

// hdr_sampling.cpp:107-176: q = importance / mean; partition below/above-average texels from both ends of one
// table; each below-average texel takes the current above-average texel as alias, which gives up (1-q).
float HdrSampling::buildAliasmap(const std::vector<float>& data, std::vector<rt_impt_samp>& accel)
{
  const uint32_t n = uint32_t(data.size());
  const float sum = std::accumulate(data.begin(), data.end(), 0.f);
  const float invAvg = float(n) / sum;
  for(uint32_t i = 0; i < n; i++) { accel[i].q = data[i] * invAvg; accel[i].alias = int32_t(i); }
  std::vector<uint32_t> part(n);
  uint32_t lo = 0, hi = n;
  for(uint32_t i = 0; i < n; i++) { if(accel[i].q < 1.f) part[lo++] = i; else part[--hi] = i; }
  for(lo = 0; lo < hi && hi < n; ++lo) {
    const uint32_t small = part[lo], big = part[hi];
    accel[small].alias = int32_t(big);
    accel[big].q -= 1.f - accel[small].q;
    if(accel[big].q < 1.0f) hi++;
  }
  return sum;
}

// hdr_sampling.cpp:181-242
void HdrSampling::createEnvironmentAccel()
{
  const uint32_t rx = uint32_t(m_w), ry = uint32_t(m_h);
  m_accel.assign(size_t(rx) * ry, rt_impt_samp{});
  std::vector<float> importance(size_t(rx) * ry);
  float cosTheta0 = 1.0f;
  const float stepPhi = float(2.0 * M_PI) / float(rx);
  const float stepTheta = float(M_PI) / float(ry);
  double total = 0;
  const float* px = m_pixels.data();
  for(uint32_t y = 0; y < ry; ++y) {
    const float cosTheta1 = std::cos(float(y + 1) * stepTheta);
    const float area = (cosTheta0 - cosTheta1) * stepPhi;
    cosTheta0 = cosTheta1;
    for(uint32_t x = 0; x < rx; ++x) {
      const size_t i = size_t(y) * rx + x;
      total += luminance(&px[i * 4]);
      importance[i] = area * std::max(px[i * 4], std::max(px[i * 4 + 1], px[i * 4 + 2]));
    }
  }
  m_average = float(total) / float(rx * ry);
  m_integral = buildAliasmap(importance, m_accel);
  const float invInt = 1.0f / m_integral;
  for(size_t i = 0; i < size_t(rx) * ry; ++i) m_accel[i].pdf = std::max(px[i * 4], std::max(px[i * 4 + 1], px[i * 4 + 2])) * invInt;
  for(size_t i = 0; i < size_t(rx) * ry; ++i) m_accel[i].aliasPdf = m_accel[size_t(m_accel[i].alias)].pdf;
}

}  // namespace rth
