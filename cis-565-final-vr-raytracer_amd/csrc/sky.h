// sky.h — the procedural sun & sky environment of shaders/sun_and_sky.glsl:453-601 on the device.
// Everything in sun_and_sky() that depends only on the uniform block (sun direction after the horizon tweak, night factor,
// haze / saturation, both sun colours, the physically-scaled-sun factors and the 25-direction ground irradiance of
// calc_irrad, sun_and_sky.glsl:274-294) is evaluated ONCE per rt_set_sun_and_sky by a one-thread kernel (skyPrepare) with
// the same expressions in the same order, so per-ray work is one calc_env_color + the disk / horizon / colour tweaks.
// Arithmetic: include/rt_detmath.h functions, left-to-right evaluation of the GLSL expressions (all GLSL literals are floats).
#pragma once
#include "dev_math.h"

namespace rt {

struct SkyPre {  // uniform-only terms of sun_and_sky()
  f3 sunDir, realSunDir, rgbScale, sunColorUp, sunColorDown, downBase, night;
  f3 rawSunDir; float sampleRadius;   // EnvSample's cone around the un-tweaked direction (env_sampling.glsl:114-122)
  float horizHeight, localHaze, localSaturation, factor, diskScale, glowScale, sunRadius, horBlur;
  float sunDiskIntensity, sunGlowIntensity, redblueshift;
  int yIsUp, diskOn, black;
};

namespace skyfn {

constexpr float PI_S = 3.1415926535f;  // sun_and_sky.glsl:26

RT_DEV float lum(f3 c) { return (0.2126f * c.x + 0.7152f * c.y) + 0.0722f * c.z; }  // :31-34
RT_DEV f3 exp3(f3 v) { return mk3(rt_exp(v.x), rt_exp(v.y), rt_exp(v.z)); }
RT_DEV f3 pow3(f3 v, float e) { return mk3(rt_pow(v.x, e), rt_pow(v.y, e), rt_pow(v.z, e)); }
RT_DEV float smooth(float e0, float e1, float x) { const float t = rt_clamp((x - e0) / (e1 - e0), 0.0f, 1.0f); return (t * t) * (3.0f - 2.0f * t); }

RT_DEV f3 vectortweak(f3 d, int yIsUp, float horizHeight)  // :317-330
{
  f3 o = (yIsUp == 1) ? mk3(d.x, d.z, d.y) : d;
  if(horizHeight != 0.0f) { o.z -= horizHeight; o = normalize(o); }
  return o;
}

// Perez term shared by sky_luminance (:226-254) and the two chromaticity fits of sky_color_xyz (:165-223)
RT_DEV float perezRatio(float A, float B, float C, float D, float E, float cosTheta, float gamma, float cosGamma, float thetaSun, float cosThetaSun)
{
  const float num = (1.0f + A * rt_exp(B / cosTheta)) * ((1.0f + C * rt_exp(D * gamma)) + (E * cosGamma) * cosGamma);
  const float den = (1.0f + A * rt_exp(B / 1.0f)) * ((1.0f + C * rt_exp(D * thetaSun)) + (E * cosThetaSun) * cosThetaSun);
  return num / den;
}

RT_DEV f3 envColor(f3 sun, f3 dir, float T)  // calc_env_color :257-272
{
  const float thetaSun = rt_acos(sun.z);
  const float chi = (4.0f / 9.0f - T / 120.0f) * (PI_S - 2.0f * thetaSun);
  float Y = 1000.0f * ((((4.0453f * T - 4.9710f) * rt_tan(chi)) - 0.2155f * T) + 2.4192f);
  const float cg0 = dot(sun, dir);
  {  // sky_luminance
    float cg = cg0;
    if(cg < 0.0f) cg = 0.0f;
    if(cg > 1.0f) cg = 2.0f - cg;
    Y *= perezRatio(0.178721f * T - 1.463037f, -0.355402f * T + 0.427494f, -0.022669f * T + 5.325056f, 0.120647f * T - 2.577052f, -0.066967f * T + 0.370275f,
                    dir.z, rt_acos(cg), cg, thetaSun, sun.z);
  }
  // sky_color_xyz
  float cg = cg0;
  if(cg > 1.0f) cg = 2.0f - cg;
  const float gamma = rt_acos(cg);
  const float t2 = T * T, s2 = thetaSun * thetaSun, s3 = s2 * thetaSun;
  const float zx = ((((0.001650f * s3 - 0.003742f * s2) + 0.002088f * thetaSun) + 0.0f) * t2 + (((-0.029028f * s3 + 0.063773f * s2) - 0.032020f * thetaSun) + 0.003948f) * T)
                   + (((0.116936f * s3 - 0.211960f * s2) + 0.060523f * thetaSun) + 0.258852f);
  const float zy = ((((0.002759f * s3 - 0.006105f * s2) + 0.003162f * thetaSun) + 0.0f) * t2 + (((-0.042149f * s3 + 0.089701f * s2) - 0.041536f * thetaSun) + 0.005158f) * T)
                   + (((0.153467f * s3 - 0.267568f * s2) + 0.066698f * thetaSun) + 0.266881f);
  float x = perezRatio(-0.019257f * T - (0.29f - rt_pow(sun.z, 0.5f) * 0.09f), -0.066513f * T + 0.000818f, -0.000417f * T + 0.212479f, -0.064097f * T - 0.898875f,
                       -0.003251f * T + 0.045178f, dir.z, gamma, cg, thetaSun, sun.z);
  float y = perezRatio(-0.016698f * T - 0.260787f, -0.094958f * T + 0.009213f, -0.007928f * T + 0.210230f, -0.044050f * T - 1.653694f, -0.010922f * T + 0.052919f, dir.z,
                       gamma, cg, thetaSun, sun.z);
  const float sat = 1.0f;
  x = zx * ((x * sat) + (1.0f - sat));
  y = zy * ((y * sat) + (1.0f - sat));
  const float X = (x / y) * Y, Z = (((1.0f - x) - y) / y) * Y;
  f3 c = mk3((3.241f * X - 1.537f * Y) - 0.499f * Z, (-0.969f * X + 1.876f * Y) + 0.042f * Z, (0.056f * X - 0.204f * Y) + 1.057f * Z);
  c *= PI_S;
  return c;
}

RT_DEV f3 sunColor(f3 sun, float T)  // calc_sun_color :139-162
{
  if(!(sun.z > 0.0f)) return mk3(0.0f);
  const f3 ko = mk3(12.0f, 8.5f, 0.9f), wl = mk3(0.610f, 0.550f, 0.470f);
  const f3 solRad = mk3(1.0f * 127500.0f / 0.9878f, 0.992f * 127500.0f / 0.9878f, 0.911f * 127500.0f / 0.9878f);
  const float m = 1.0f / (sun.z + 0.15f * rt_pow(93.885f - rt_acos(sun.z) * 180.0f / PI_S, -1.253f));
  const float beta = 0.04608f * T - 0.04586f;
  const f3 ta = exp3(pow3(wl, -1.3f) * (-m * beta));
  const f3 to = exp3((ko * (-m)) * 0.0035f);
  const f3 tr = exp3(pow3(wl, -4.08f) * (-m * 0.008735f));
  return ((tr * ta) * to) * solRad;
}

RT_DEV f3 diffuseDir(float su, float sv)  // mi_reflection_dir_diffuse_x around +z (:118-136) with mi_lib_square_to_disk (:73-115) and xyz2dir (:37-70)
{
  const float lx = 2.0f * su - 1.0f, ly = 2.0f * sv - 1.0f;
  float r = 0.0f, phi = 0.0f;
  if(!(lx == 0.0f && ly == 0.0f)) {
    if(lx > -ly) {
      if(lx > ly) { r = lx; phi = (PI_S / 4.0f) * (1.0f + ly / lx); }
      else { r = ly; phi = (PI_S / 4.0f) * (3.0f - lx / ly); }
    } else {
      if(lx < ly) { r = -lx; phi = (PI_S / 4.0f) * (5.0f + ly / lx); }
      else { r = -ly; phi = (PI_S / 4.0f) * (7.0f - lx / ly); }
    }
  }
  const float x = r * rt_cos(phi), y = r * rt_sin(phi);
  const float z2 = (1.0f - x * x) - y * y;
  const float z = z2 > 0.0f ? rt_sqrt(z2) : 0.0f;
  // xyz2dir with main = (0,0,1): |main.x| < |main.y| is false => u = normalize((main.z, 0, -main.x)), v = cross(main, u)
  const f3 main = mk3(0.0f, 0.0f, 1.0f);
  const f3 u = normalize(mk3(main.z, 0.0f, -main.x));
  const f3 v = cross(main, u);
  return (u * x + v * y) + main * z;
}

// one thread: the uniform-only part of sun_and_sky(), :453-601
RT_DEV void prepare(const rt_sun_and_sky& ss, SkyPre& P)
{
  P.yIsUp = ss.y_is_up;
  P.rawSunDir = mk3(ss.sun_direction.x, ss.sun_direction.y, ss.sun_direction.z);
  P.sampleRadius = (0.00465f * 10.0f) * ss.sun_disk_scale;
  P.horizHeight = ss.horizon_height / 10.0f;
  float haze = 2.0f + ss.haze;
  if(haze < 2.0f) haze = 2.0f;
  P.localHaze = haze;
  {  // tweak_saturation :297-314
    const float s = ss.saturation, lowsat = rt_pow(s, 3.0f);
    if(s <= 1.0f) {
      float h = haze;
      h -= 2.0f; h /= 15.0f;
      if(h < 0.0f) h = 0.0f;
      if(h > 1.0f) h = 1.0f;
      h = rt_pow(h, 3.0f);
      P.localSaturation = (s * (1.0f - h)) + lowsat * h;
    } else P.localSaturation = 1.0f;
  }
  f3 scale = mk3(ss.rgb_unit_conversion.x, ss.rgb_unit_conversion.y, ss.rgb_unit_conversion.z);
  if(lum(scale) < 0.0f) scale = mk3(1.0f / 80000.0f);
  scale *= ss.multiplier;
  P.rgbScale = scale;
  P.black = ss.multiplier <= 0.0f ? 1 : 0;
  f3 sun = normalize(mk3(ss.sun_direction.x, ss.sun_direction.y, ss.sun_direction.z));
  sun = vectortweak(sun, ss.y_is_up, P.horizHeight);
  P.realSunDir = sun;
  float factor = 1.0f;
  if(sun.z < 0.001f) {
    if(sun.z < 0.0f) {  // night_brightness_adjustment :439-450
      const float lmt = 0.30901699437494742410229341718282f;
      if(sun.z <= -lmt) factor = 0.0f;
      else { factor = (sun.z + lmt) / lmt; factor *= factor; factor *= factor; }
    }
    sun.z = 0.001f;
    sun = normalize(sun);
  }
  P.sunDir = sun; P.factor = factor;
  P.sunColorUp = sunColor(sun, haze);
  P.sunColorDown = sunColor(sun, 2.0f);
  P.diskOn = (ss.sun_disk_intensity > 0.0f && ss.sun_disk_scale > 0.0f) ? 1 : 0;
  P.sunRadius = (0.00465f * ss.sun_disk_scale) * 10.0f;
  P.sunDiskIntensity = ss.sun_disk_intensity; P.sunGlowIntensity = ss.sun_glow_intensity; P.redblueshift = ss.redblueshift;
  P.diskScale = 1.0f; P.glowScale = 1.0f;
  if(ss.physically_scaled_sun == 1) {  // calc_physical_scale :365-436
    const float diskR = 0.00465f * ss.sun_disk_scale, glowR = diskR * 10.0f;
    const float glowInt = ss.sun_glow_intensity * (((4.0f * PI_S) - (24.0f * PI_S) / (glowR * glowR)) + (24.0f * PI_S) * rt_sin(glowR) / ((glowR * glowR) * glowR));
    float target = ss.sun_disk_intensity * PI_S;
    const float maxGlow = 0.5f * target;
    if(glowInt > maxGlow) { P.glowScale *= maxGlow / glowInt; target -= maxGlow; }
    else target -= glowInt;
    const float area = (2.0f * PI_S) * (1.0f - rt_cos(diskR));
    const float targetIntensity = target / area;
    const float actualIntegral = 1.0f * area;
    const float actualIntensity = ((ss.sun_disk_intensity * 100.0f) * actualIntegral) / area;
    P.diskScale = (targetIntensity == 0.0f) ? 0.0f : targetIntensity / actualIntensity;
  }
  // ground: downcolor = ground_color * (irrad + sun_color(2.0) * sun.z) * rgb_scale [* factor]   (:547-556)
  f3 acc = mk3(0.0f);
  for(float u = 1.0f / 10.0f; u < 1.0f; u += 1.0f / 5.0f)
    for(float v = 1.0f / 10.0f; v < 1.0f; v += 1.0f / 5.0f) acc += envColor(sun, diffuseDir(u, v), 2.0f);
  const f3 irrad = acc / 25.0f;
  f3 down = mk3(ss.ground_color.x, ss.ground_color.y, ss.ground_color.z);
  down *= (irrad + P.sunColorDown * sun.z) * scale;
  if(factor < 1.0f) down *= factor;
  P.downBase = down;
  P.horBlur = ss.horizon_blur / 10.0f;
  P.night = mk3(ss.night_color.x, ss.night_color.y, ss.night_color.z);
}

RT_DEV f3 evaluate(const SkyPre& P, f3 inDir)  // sun_and_sky(), the per-direction part
{
  if(P.black) return mk3(0.0f);
  f3 dir = vectortweak(inDir, P.yIsUp, P.horizHeight);
  const float downness = dir.z;
  const f3 realDir = dir;
  if(dir.z < 0.001f) { dir.z = 0.001f; dir = normalize(dir); }
  f3 tint = mk3(0.0f);
  if(P.factor > 0.0f) {
    tint = envColor(P.sunDir, dir, P.localHaze);
    if(P.factor < 1.0f) tint *= P.factor;
  }
  const f3 sunCol = downness > 0.0f ? P.sunColorUp : P.sunColorDown;
  if(P.diskOn) {
    const float angle = rt_acos(dot(realDir, P.realSunDir));
    if(angle < P.sunRadius) {
      float f = (1.0f - angle / P.sunRadius) * 10.0f;
      f = ((rt_pow(f / 10.0f, 3.0f) * 2.0f) * P.sunGlowIntensity) * P.glowScale + ((smooth(8.5f, 9.5f + (P.localHaze / 50.0f), f) * 100.0f) * P.sunDiskIntensity) * P.diskScale;
      tint += sunCol * f;
    }
  }
  f3 out = tint * P.rgbScale;
  float nightFactor = 1.0f;
  if(downness <= 0.0f) {
    if(P.horBlur > 0.0f) {
      float d = -downness;
      d /= P.horBlur;
      if(d > 1.0f) d = 1.0f;
      d = smooth(0.0f, 1.0f, d);
      out = out * (1.0f - d) + P.downBase * d;
      nightFactor = 1.0f - d;
    } else {
      out = P.downBase;
      nightFactor = 0.0f;
    }
  }
  {  // arch_colortweak :333-362
    const float intensity = lum(out);
    f3 t = (P.localSaturation <= 0.0f) ? mk3(intensity) : (out * P.localSaturation + intensity * (1.0f - P.localSaturation));
    t *= mk3(1.0f + P.redblueshift, 1.0f, 1.0f - P.redblueshift);
    out = t;
  }
  if(nightFactor > 0.0f) {
    const f3 n = P.night * nightFactor;
    if(out.x < n.x) out.x = n.x;
    if(out.y < n.y) out.y = n.y;
    if(out.z < n.z) out.z = n.z;
  }
  out *= PI_S;
  return out;
}

}  // namespace skyfn
}  // namespace rt
