// hdr_sampling.hpp — environment map + importance-sampling acceleration data.
// Mirrors the data-producing half of the reference's HdrSampling (src/hdr_sampling.hpp:38-66,
// src/hdr_sampling.cpp:56-242); the Vulkan upload half is replaced by rt_upload_scene (rt_scene_desc.env*).
#pragma once
#include <string>
#include <vector>
#include "../../include/rt_abi.h"

namespace rth {

class HdrSampling {
 public:
  // hdr_sampling.cpp:56-99 — Radiance .hdr (RGBE, RLE or flat) -> RGBA32F + accel.  Returns false on I/O or format errors.
  bool loadEnvironment(const std::string& hdrImage);
  // same products from pixels already in memory (RGBA32F, w*h*4 floats)
  void setEnvironment(const float* rgba, int w, int h);
  // procedural stand-in for the absent daytime.hdr: sky gradient + sun lobe
  void makeSyntheticSky(int w, int h, float sunPeak, uint32_t seed);

  float getIntegral() const { return m_integral; }  // hdr_sampling.hpp:52
  float getAverage() const { return m_average; }    // hdr_sampling.hpp:53
  int width() const { return m_w; }
  int height() const { return m_h; }
  const std::vector<float>& pixels() const { return m_pixels; }
  const std::vector<rt_impt_samp>& accel() const { return m_accel; }

 private:
  void createEnvironmentAccel();                                                     // hdr_sampling.cpp:181-242
  static float buildAliasmap(const std::vector<float>& data, std::vector<rt_impt_samp>& accel);  // hdr_sampling.cpp:107-176
  int m_w = 0, m_h = 0;
  std::vector<float> m_pixels;
  std::vector<rt_impt_samp> m_accel;
  float m_integral = 1.f, m_average = 1.f;
};

}  // namespace rth
