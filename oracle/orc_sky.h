// orc_sky.h — CPU restatement of shaders/sun_and_sky.glsl (TEST INFRASTRUCTURE; see oracle/README.md).
//   helpers          sun_and_sky.glsl:31-136     luminance, xyz2dir, mi_lib_square_to_disk, mi_reflection_dir_diffuse_x
//   calc_sun_color   :139-162      sky_color_xyz :165-223      sky_luminance :226-254      calc_env_color :257-272
//   calc_irrad       :274-294      tweak_saturation :297-314   arch_vectortweak :317-330   arch_colortweak :333-362
//   calc_physical_scale :365-436   night_brightness_adjustment :439-450                    sun_and_sky :453-601
// PINNED WITHIN TOLERANCE: tests/test_kat_float.py compares sun_and_sky() with vectors minted from sun_and_sky.glsl itself
// compiled on a vector shim (oracle/kat/kat_float.cpp): <= 7e-5 relative over 6 parameter sets x 48 directions.  Not bit-exact
// by construction: GLSL leaves exp/pow/acos/sin/cos/tan precision to the driver; here they are the contract functions of
// include/rt_detmath.h and every expression is evaluated left to right as written in the GLSL (all literals are floats there).
// tests/test_sky.py additionally checks this file against an independent float64 numpy statement of the same model.
#pragma once
#include "orc_math.h"
#include "../include/rt_abi.h"

namespace orc {
namespace sky {

inline vec3 rv(const rt_vec3& v) { return vec3{v.x, v.y, v.z}; }
constexpr float PI_SKY = 3.1415926535f;  // sun_and_sky.glsl:26 (same float as globals.glsl's M_PI)

inline float lum(vec3 rgb) { return (0.2126f * rgb.x + 0.7152f * rgb.y) + 0.0722f * rgb.z; }
inline vec3 vexp(vec3 v) { return {rt_exp(v.x), rt_exp(v.y), rt_exp(v.z)}; }
inline vec3 vpow(vec3 v, float e) { return {rt_pow(v.x, e), rt_pow(v.y, e), rt_pow(v.z, e)}; }
inline float smoothstep(float e0, float e1, float x)
{
  float t = rt_clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
  return (t * t) * (3.0f - 2.0f * t);
}

inline vec3 xyz2dir(vec3 in_main, float x, float y, float z)
{
  vec3 u;
  if(rt_abs(in_main.x) < rt_abs(in_main.y)) u = V3(0.0f, -in_main.z, in_main.y);
  else u = V3(in_main.z, 0.0f, -in_main.x);
  // (the "degenerate transform" branch recomputes the same u)
  u = normalize(u);
  vec3 v = cross(in_main, u);
  return (x * u + y * v) + z * in_main;
}

inline void square_to_disk(float in_x, float in_y, float& r, float& phi)
{
  const float lx = 2.0f * in_x - 1.0f, ly = 2.0f * in_y - 1.0f;
  if(lx == 0.0f && ly == 0.0f) { phi = 0.0f; r = 0.0f; return; }
  if(lx > -ly) {
    if(lx > ly) { r = lx; phi = (PI_SKY / 4.0f) * (1.0f + ly / lx); }
    else { r = ly; phi = (PI_SKY / 4.0f) * (3.0f - lx / ly); }
  } else {
    if(lx < ly) { r = -lx; phi = (PI_SKY / 4.0f) * (5.0f + ly / lx); }
    else { r = -ly; phi = (PI_SKY / 4.0f) * (7.0f - lx / ly); }
  }
}

inline vec3 reflection_dir_diffuse_x(vec3 n, float sx, float sy)
{
  float r, phi;
  square_to_disk(sx, sy, r, phi);
  const float x = r * rt_cos(phi), y = r * rt_sin(phi);
  const float z2 = (1.0f - x * x) - y * y;
  const float z = z2 > 0.0f ? rt_sqrt(z2) : 0.0f;
  return xyz2dir(n, x, y, z);
}

inline vec3 calc_sun_color(vec3 sun_dir, float turbidity)
{
  vec3 sun_color = V3(0.0f);
  const vec3 ko = V3(12.0f, 8.5f, 0.9f), wavelength = V3(0.610f, 0.550f, 0.470f);
  const vec3 solRad = V3(1.0f * 127500.0f / 0.9878f, 0.992f * 127500.0f / 0.9878f, 0.911f * 127500.0f / 0.9878f);
  if(sun_dir.z > 0.0f) {
    const float m = 1.0f / (sun_dir.z + 0.15f * rt_pow(93.885f - rt_acos(sun_dir.z) * 180.0f / PI_SKY, -1.253f));
    const float beta = 0.04608f * turbidity - 0.04586f;
    const float alpha = 1.3f;
    vec3 ta = vexp((-m * beta) * vpow(wavelength, -alpha));
    const float l = 0.0035f;
    vec3 to = vexp(((-m) * ko) * l);
    vec3 tr = vexp((-m * 0.008735f) * vpow(wavelength, -4.08f));
    sun_color = ((tr * ta) * to) * solRad;
  }
  return sun_color;
}

inline float perez(float A, float B, float C, float D, float E, float cos_theta, float gamma, float cos_gamma, float theta_sun, float cos_theta_sun)
{
  const float n1 = 1.0f + A * rt_exp(B / cos_theta);
  const float n2 = (1.0f + C * rt_exp(D * gamma)) + (E * cos_gamma) * cos_gamma;
  const float d1 = 1.0f + A * rt_exp(B / 1.0f);
  const float d2 = (1.0f + C * rt_exp(D * theta_sun)) + (E * cos_theta_sun) * cos_theta_sun;
  return (n1 * n2) / (d1 * d2);
}

inline vec3 sky_color_xyz(vec3 in_dir, vec3 in_sun_pos, float T, float in_luminance)
{
  float cos_gamma = dot(in_sun_pos, in_dir);
  if(cos_gamma > 1.0f) cos_gamma = 2.0f - cos_gamma;
  const float gamma = rt_acos(cos_gamma);
  const float cos_theta = in_dir.z, cos_theta_sun = in_sun_pos.z;
  const float theta_sun = rt_acos(cos_theta_sun);
  const float t2 = T * T, ts2 = theta_sun * theta_sun, ts3 = ts2 * theta_sun;
  const float zenith_x = ((((+0.001650f * ts3 - 0.003742f * ts2) + 0.002088f * theta_sun) + 0.0f) * t2
                          + (((-0.029028f * ts3 + 0.063773f * ts2) - 0.032020f * theta_sun) + 0.003948f) * T)
                         + (((+0.116936f * ts3 - 0.211960f * ts2) + 0.060523f * theta_sun) + 0.258852f);
  const float zenith_y = ((((+0.002759f * ts3 - 0.006105f * ts2) + 0.003162f * theta_sun) + 0.0f) * t2
                          + (((-0.042149f * ts3 + 0.089701f * ts2) - 0.041536f * theta_sun) + 0.005158f) * T)
                         + (((+0.153467f * ts3 - 0.267568f * ts2) + 0.066698f * theta_sun) + 0.266881f);
  vec3 xyz;
  xyz.y = in_luminance;
  float A = -0.019257f * T - (0.29f - rt_pow(cos_theta_sun, 0.5f) * 0.09f);
  float B = -0.066513f * T + 0.000818f, C = -0.000417f * T + 0.212479f, D = -0.064097f * T - 0.898875f, E = -0.003251f * T + 0.045178f;
  float x = perez(A, B, C, D, E, cos_theta, gamma, cos_gamma, theta_sun, cos_theta_sun);
  A = -0.016698f * T - 0.260787f; B = -0.094958f * T + 0.009213f; C = -0.007928f * T + 0.210230f; D = -0.044050f * T - 1.653694f; E = -0.010922f * T + 0.052919f;
  float y = perez(A, B, C, D, E, cos_theta, gamma, cos_gamma, theta_sun, cos_theta_sun);
  const float local_saturation = 1.0f;
  x = zenith_x * ((x * local_saturation) + (1.0f - local_saturation));
  y = zenith_y * ((y * local_saturation) + (1.0f - local_saturation));
  xyz.x = (x / y) * xyz.y;
  xyz.z = (((1.0f - x) - y) / y) * xyz.y;
  return xyz;
}

inline float sky_luminance(vec3 in_dir, vec3 in_sun_pos, float T)
{
  float cos_gamma = dot(in_sun_pos, in_dir);
  if(cos_gamma < 0.0f) cos_gamma = 0.0f;
  if(cos_gamma > 1.0f) cos_gamma = 2.0f - cos_gamma;
  const float gamma = rt_acos(cos_gamma);
  const float cos_theta = in_dir.z, cos_theta_sun = in_sun_pos.z;
  const float theta_sun = rt_acos(cos_theta_sun);
  const float A = 0.178721f * T - 1.463037f, B = -0.355402f * T + 0.427494f, C = -0.022669f * T + 5.325056f, D = 0.120647f * T - 2.577052f,
              E = -0.066967f * T + 0.370275f;
  return perez(A, B, C, D, E, cos_theta, gamma, cos_gamma, theta_sun, cos_theta_sun);
}

inline vec3 calc_env_color(vec3 in_sun_dir, vec3 in_dir, float T)
{
  const float theta_sun = rt_acos(in_sun_dir.z);
  const float chi = (4.0f / 9.0f - T / 120.0f) * (PI_SKY - 2.0f * theta_sun);
  float luminance = 1000.0f * ((((4.0453f * T - 4.9710f) * rt_tan(chi)) - 0.2155f * T) + 2.4192f);
  luminance *= sky_luminance(in_dir, in_sun_dir, T);
  const vec3 XYZ = sky_color_xyz(in_dir, in_sun_dir, T, luminance);
  vec3 env_color = V3((3.241f * XYZ.x - 1.537f * XYZ.y) - 0.499f * XYZ.z, (-0.969f * XYZ.x + 1.876f * XYZ.y) + 0.042f * XYZ.z,
                      (0.056f * XYZ.x - 0.204f * XYZ.y) + 1.057f * XYZ.z);
  env_color *= PI_SKY;
  return env_color;
}

inline vec3 calc_irrad(vec3 sun_dir, float haze)
{
  vec3 colaccu = V3(0.0f);
  const vec3 n = V3(0.0f, 0.0f, 1.0f);
  for(float u = 1.0f / 10.0f; u < 1.0f; u += 1.0f / 5.0f)
    for(float v = 1.0f / 10.0f; v < 1.0f; v += 1.0f / 5.0f) {
      const vec3 diff = reflection_dir_diffuse_x(n, u, v);
      colaccu += calc_env_color(sun_dir, diff, haze);
    }
  return colaccu / 25.0f;
}

inline float tweak_saturation(float saturation, float haze)
{
  const float lowsat = rt_pow(saturation, 3.0f);
  if(saturation <= 1.0f) {
    float h = haze;
    h -= 2.0f;
    h /= 15.0f;
    if(h < 0.0f) h = 0.0f;
    if(h > 1.0f) h = 1.0f;
    h = rt_pow(h, 3.0f);
    return (saturation * (1.0f - h)) + lowsat * h;
  }
  return 1.0f;
}

inline vec3 arch_vectortweak(vec3 dir, int y_is_up, float horiz_height)
{
  vec3 o = dir;
  if(y_is_up == 1) o = V3(dir.x, dir.z, dir.y);
  if(horiz_height != 0.0f) { o.z -= horiz_height; o = normalize(o); }
  return o;
}

inline vec3 arch_colortweak(vec3 tint, float saturation, float redness)
{
  const float intensity = lum(tint);
  vec3 out_tint;
  if(saturation <= 0.0f) out_tint = V3(intensity);
  else out_tint = tint * saturation + intensity * (1.0f - saturation);  // (the clamp that follows in the GLSL writes a dead variable)
  out_tint *= V3(1.0f + redness, 1.0f, 1.0f - redness);
  return out_tint;
}

inline vec2 calc_physical_scale(float sun_disk_scale, float sun_glow_intensity, float sun_disk_intensity)
{
  const float sun_angular_radius = 0.00465f;
  const float sun_disk_radius = sun_angular_radius * sun_disk_scale;
  const float sun_glow_radius = sun_disk_radius * 10.0f;
  const float glow_func_integral =
      sun_glow_intensity * (((4.0f * PI_SKY) - (24.0f * PI_SKY) / (sun_glow_radius * sun_glow_radius))
                            + (24.0f * PI_SKY) * rt_sin(sun_glow_radius) / ((sun_glow_radius * sun_glow_radius) * sun_glow_radius));
  float target_sundisk_integral = sun_disk_intensity * PI_SKY;
  float sky_sunglow_scale = 1.0f;
  const float max_glow_integral = 0.5f * target_sundisk_integral;
  if(glow_func_integral > max_glow_integral) { sky_sunglow_scale *= max_glow_integral / glow_func_integral; target_sundisk_integral -= max_glow_integral; }
  else target_sundisk_integral -= glow_func_integral;
  const float sundisk_area = (2.0f * PI_SKY) * (1.0f - rt_cos(sun_disk_radius));
  const float target_sundisk_intensity = target_sundisk_integral / sundisk_area;
  const float actual_sundisk_integral = 1.0f * sundisk_area;
  const float actual_sundisk_intensity = ((sun_disk_intensity * 100.0f) * actual_sundisk_integral) / sundisk_area;
  return V2((target_sundisk_intensity == 0.0f) ? 0.0f : target_sundisk_intensity / actual_sundisk_intensity, sky_sunglow_scale);
}

inline float night_brightness_adjustment(vec3 sun_dir)
{
  const float lmt = 0.30901699437494742410229341718282f;
  if(sun_dir.z <= -lmt) return 0.0f;
  float factor = (sun_dir.z + lmt) / lmt;
  factor *= factor;
  factor *= factor;
  return factor;
}

inline vec3 sun_and_sky(const rt_sun_and_sky& ss, vec3 in_direction)
{
  float factor = 1.0f, night_factor = 1.0f;
  vec3 rgb_scale = rv(ss.rgb_unit_conversion);
  const float horiz_height = ss.horizon_height / 10.0f;
  vec3 dir = arch_vectortweak(in_direction, ss.y_is_up, horiz_height);
  float local_haze = 2.0f + ss.haze;
  if(local_haze < 2.0f) local_haze = 2.0f;
  const float local_saturation = tweak_saturation(ss.saturation, local_haze);
  if(lum(rgb_scale) < 0.0f) rgb_scale = V3(1.0f / 80000.0f);
  rgb_scale *= ss.multiplier;
  if(ss.multiplier <= 0.0f) return V3(0.0f);

  const float downness = dir.z;
  const vec3 real_dir = dir;
  if(dir.z < 0.001f) { dir.z = 0.001f; dir = normalize(dir); }

  vec3 sun_dir = normalize(rv(ss.sun_direction));
  sun_dir = arch_vectortweak(sun_dir, ss.y_is_up, horiz_height);
  const vec3 real_sun_dir = sun_dir;
  if(sun_dir.z < 0.001f) {
    if(sun_dir.z < 0.0f) factor = night_brightness_adjustment(sun_dir);
    sun_dir.z = 0.001f;
    sun_dir = normalize(sun_dir);
  }

  vec3 tint;
  if(factor > 0.0f) {
    tint = calc_env_color(sun_dir, dir, local_haze);
    if(factor < 1.0f) tint *= factor;
  } else tint = V3(0.0f);
  const vec3 data_sun_color = calc_sun_color(sun_dir, downness > 0.0f ? local_haze : 2.0f);
  if(ss.sun_disk_intensity > 0.0f && ss.sun_disk_scale > 0.0f) {
    const float sun_angle = rt_acos(dot(real_dir, real_sun_dir));
    const float sun_radius = (0.00465f * ss.sun_disk_scale) * 10.0f;
    if(sun_angle < sun_radius) {
      float sky_sundisk_scale = 1.0f, sky_sunglow_scale = 1.0f;
      if(ss.physically_scaled_sun == 1) {
        const vec2 rv = calc_physical_scale(ss.sun_disk_scale, ss.sun_glow_intensity, ss.sun_disk_intensity);
        sky_sundisk_scale = rv.x; sky_sunglow_scale = rv.y;
      }
      float sun_factor = (1.0f - sun_angle / sun_radius) * 10.0f;
      sun_factor = ((rt_pow(sun_factor / 10.0f, 3.0f) * 2.0f) * ss.sun_glow_intensity) * sky_sunglow_scale
                   + ((smoothstep(8.5f, 9.5f + (local_haze / 50.0f), sun_factor) * 100.0f) * ss.sun_disk_intensity) * sky_sundisk_scale;
      tint += data_sun_color * sun_factor;
    }
  }
  vec3 out_color = tint * rgb_scale;
  if(downness <= 0.0f) {
    vec3 downcolor = rv(ss.ground_color);
    const vec3 irrad = calc_irrad(sun_dir, 2.0f);
    downcolor *= (irrad + data_sun_color * sun_dir.z) * rgb_scale;
    if(factor < 1.0f) downcolor *= factor;
    const float hor_blur = ss.horizon_blur / 10.0f;
    if(hor_blur > 0.0f) {
      float dness = -downness;
      dness /= hor_blur;
      if(dness > 1.0f) dness = 1.0f;
      dness = smoothstep(0.0f, 1.0f, dness);
      out_color = out_color * (1.0f - dness) + downcolor * dness;
      night_factor = 1.0f - dness;
    } else {
      out_color = downcolor;
      night_factor = 0.0f;
    }
  }
  out_color = arch_colortweak(out_color, local_saturation, ss.redblueshift);
  vec3 result = out_color;
  if(night_factor > 0.0f) {
    const vec3 night = rv(ss.night_color) * night_factor;
    if(result.x < night.x) result.x = night.x;
    if(result.y < night.y) result.y = night.y;
    if(result.z < night.z) result.z = night.z;
  }
  result *= PI_SKY;
  return result;
}

}  // namespace sky
}  // namespace orc
