/*
 * rt_detmath.h — the numerics contract of the ReSTIR frame path.
 *
 * GLSL leaves exp/log/pow/sin/cos/asin/acos/atan precision implementation-defined, so the reference's
 * float outputs differ between drivers, and ReSTIR turns a 1-ulp difference into a different reservoir
 * sample.  To make "same scene + same seed => same frame" a checkable statement, every transcendental on
 * the path is defined here from IEEE-754 binary32 +,-,*,/ ,sqrt and integer bit operations only (Cephes
 * single-precision kernels, Cody–Waite range reduction).  Any implementation of the path — the CPU oracle
 * under oracle/ and the gfx950 HIP kernels under csrc/ — that evaluates these expressions in the written
 * order, without FMA contraction or fast-math, produces bit-identical results.
 *
 * Build rules for bit-reproducibility (both g++ and hipcc): -ffp-contract=off, no -ffast-math,
 * correctly rounded f32 divide/sqrt (hipcc: -fhip-fp32-correctly-rounded-divide-sqrt, the default),
 * denormals kept (hipcc default for gfx9+).
 *
 * Accuracy (checked in tests/test_detmath.py against float64): exp/log/sin/cos/asin/acos/atan2 <= 4 ulp
 * on the ranges the path uses; pow(x, 2.2) relative error <= 4e-6.
 */
#ifndef RT_DETMATH_H
#define RT_DETMATH_H

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ inline
#elif defined(__cplusplus)
#define RT_HD inline
#else
#include <stdbool.h>
#define RT_HD static inline
#endif

RT_HD uint32_t rt_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
RT_HD float rt_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
RT_HD int32_t rt_f2ibits(float f) { int32_t u; memcpy(&u, &f, 4); return u; }
RT_HD float rt_ibits2f(int32_t u) { float f; memcpy(&f, &u, 4); return f; }

RT_HD bool rt_isnan(float x) { return !(x == x); }
RT_HD bool rt_isinf(float x) { return (rt_f2u(x) & 0x7fffffffu) == 0x7f800000u; }
RT_HD float rt_abs(float x) { return rt_u2f(rt_f2u(x) & 0x7fffffffu); }
/* GLSL min/max/clamp with a defined NaN rule: the comparison decides, NaN in `a` falls through to `b`. */
RT_HD float rt_min(float a, float b) { return (a < b) ? a : b; }
RT_HD float rt_max(float a, float b) { return (a > b) ? a : b; }
RT_HD float rt_clamp(float x, float lo, float hi) { return rt_min(rt_max(x, lo), hi); }
RT_HD float rt_sqrt(float x) { return sqrtf(x); }   /* IEEE correctly rounded on both targets */
RT_HD float rt_floor(float x) { return floorf(x); } /* exact */

/* float -> int32 with defined behaviour everywhere (C leaves NaN / out-of-range undefined; GLSL too):
 * NaN -> 0, x >= 2147483520 -> 2147483520, x <= -2^31 -> INT32_MIN, else truncation toward zero.
 * Written with clamps + one select instead of early returns: the same function, but three fewer branches per call on the GPU
 * (rt_max(NaN, lo) = lo by its definition as a select; the last select restores the NaN case). */
RT_HD int32_t rt_ftoi(float x)
{
  const float c = rt_min(rt_max(x, -2147483648.0f), 2147483520.0f);
  const int32_t r = (int32_t)c; /* in range: truncation toward zero */
  return rt_isnan(x) ? 0 : r;
}
/* float -> uint32 (GLSL uint(x)): NaN or x <= 0 -> 0, x >= 4294967040 -> 4294967040, else truncation */
RT_HD uint32_t rt_ftou(float x)
{
  const float c = rt_min(rt_max(x, 0.0f), 4294967040.0f); /* rt_max(NaN, 0) = 0 */
  return (uint32_t)c;
}

/* 2^n for n in [-126, 127] */
RT_HD float rt_pow2i(int n) { return rt_u2f((uint32_t)(n + 127) << 23); }

/* a*b + c with ONE rounding (IEEE fusedMultiplyAdd) on both targets: v_fma_f32 on the GPU, fmaf() — a hardware FMA when the host
 * code is built with -mfma, glibc's exact software fma otherwise — on the CPU.  The compilers never form it on their own
 * (-ffp-contract=off); where the contract wants a fused step it says so with rt_fma. */
#if defined(__HIP_DEVICE_COMPILE__)
RT_HD float rt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#else
RT_HD float rt_fma(float a, float b, float c) { return fmaf(a, b, c); }
#endif

/* exp: Cody-Waite reduction by ln2 in two pieces, degree-5 polynomial (Cephes expf coefficients), every multiply-add step fused
 * (rt_fma): one instruction per step on the GPU, where exp is the hot function of the A-Trous filters (75 calls per pixel and
 * level), and one rounding instead of two per step. */
RT_HD float rt_exp(float x)
{
  if(rt_isnan(x)) return x;
  if(x > 88.72283905206835f) return rt_u2f(0x7f800000u);
  if(x < -87.33654475055310f) return 0.0f;
  float z = rt_floor(rt_fma(x, 1.44269504088896341f, 0.5f));
  int   n = (int)z;
  x = rt_fma(z, -0.693359375f, x);
  x = rt_fma(z, 2.12194440e-4f, x);
  float zz = x * x;
  float p = 1.9875691500E-4f;
  p = rt_fma(p, x, 1.3981999507E-3f);
  p = rt_fma(p, x, 8.3334519073E-3f);
  p = rt_fma(p, x, 4.1665795894E-2f);
  p = rt_fma(p, x, 1.6666665459E-1f);
  p = rt_fma(p, x, 5.0000001201E-1f);
  p = rt_fma(p, zz, x);
  p = p + 1.0f;
  int a = n >> 1;
  int b = n - a;
  return (p * rt_pow2i(a)) * rt_pow2i(b);
}

RT_HD float rt_log(float x)
{
  if(rt_isnan(x)) return x;
  if(x < 0.0f) return rt_u2f(0x7fc00000u);
  if(x == 0.0f) return rt_u2f(0xff800000u);
  if(rt_isinf(x)) return x;
  int e = 0;
  if(x < 1.17549435e-38f) { x = x * 8388608.0f; e = -23; }
  uint32_t u = rt_f2u(x);
  e += (int)((u >> 23) & 0xffu) - 126;           /* x = m * 2^e, m in [0.5, 1) */
  float m = rt_u2f((u & 0x007fffffu) | 0x3f000000u);
  if(m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
  else { m = m - 1.0f; }
  float z = m * m;
  float y = 7.0376836292E-2f;
  y = y * m + -1.1514610310E-1f;
  y = y * m + 1.1676998740E-1f;
  y = y * m + -1.2420140846E-1f;
  y = y * m + 1.4249322787E-1f;
  y = y * m + -1.6668057665E-1f;
  y = y * m + 2.0000714765E-1f;
  y = y * m + -2.4999993993E-1f;
  y = y * m + 3.3333331174E-1f;
  y = y * m * z;
  float fe = (float)e;
  y = y + -2.12194440e-4f * fe;
  y = y + -0.5f * z;
  float r = m + y;
  r = r + 0.693359375f * fe;
  return r;
}

/* GLSL pow(x, y) for x >= 0 (undefined for x < 0 in GLSL; defined as 0 here) */
RT_HD float rt_pow(float x, float y)
{
  if(rt_isnan(x) || rt_isnan(y)) return rt_u2f(0x7fc00000u);
  if(x <= 0.0f) return 0.0f;
  return rt_exp(y * rt_log(x));
}

/* shared octant reduction for sin/cos: |x| -> (r in [-pi/4, pi/4], octant j in 0..7 (even)) */
RT_HD float rt_trig_reduce(float ax, int* jo)
{
  int   j = (int)(ax * 1.27323954473516f);
  float y = (float)j;
  if(j & 1) { j += 1; y += 1.0f; }
  *jo = j & 7;
  return ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
}
RT_HD float rt_sin_poly(float x)
{
  float z = x * x;
  float y = -1.9515295891E-4f;
  y = y * z + 8.3321608736E-3f;
  y = y * z + -1.6666654611E-1f;
  y = y * z * x;
  return y + x;
}
RT_HD float rt_cos_poly(float x)
{
  float z = x * x;
  float y = 2.443315711809948E-005f;
  y = y * z + -1.388731625493765E-003f;
  y = y * z + 4.166664568298827E-002f;
  y = y * z * z;
  y = y - 0.5f * z;
  return y + 1.0f;
}
/* valid for |x| <= 8192 (the path only uses |x| <= 2*pi); larger or NaN => 0 */
RT_HD float rt_sin(float x)
{
  float ax = rt_abs(x);
  if(!(ax <= 8192.0f)) return 0.0f;
  bool neg = x < 0.0f;
  int  j;
  float r = rt_trig_reduce(ax, &j);
  if(j > 3) { neg = !neg; j -= 4; }
  float y = (j == 1 || j == 2) ? rt_cos_poly(r) : rt_sin_poly(r);
  return neg ? -y : y;
}
RT_HD float rt_cos(float x)
{
  float ax = rt_abs(x);
  if(!(ax <= 8192.0f)) return 0.0f;
  bool neg = false;
  int  j;
  float r = rt_trig_reduce(ax, &j);
  if(j > 3) { neg = !neg; j -= 4; }
  if(j > 1) neg = !neg;
  float y = (j == 1 || j == 2) ? rt_sin_poly(r) : rt_cos_poly(r);
  return neg ? -y : y;
}

/* tan as the quotient of the two contract functions (GLSL leaves tan's precision undefined) */
RT_HD float rt_tan(float x) { return rt_sin(x) / rt_cos(x); }

/* asin on [-1,1]; |x| > 1 or NaN => NaN */
RT_HD float rt_asin(float x)
{
  float a = rt_abs(x);
  if(!(a <= 1.0f)) return rt_u2f(0x7fc00000u);
  bool big = a > 0.5f;
  float z, s;
  if(big) { z = 0.5f * (1.0f - a); s = rt_sqrt(z); }
  else { s = a; z = a * a; }
  float p = 4.2163199048E-2f;
  p = p * z + 2.4181311049E-2f;
  p = p * z + 4.5470025998E-2f;
  p = p * z + 7.4953002686E-2f;
  p = p * z + 1.6666752422E-1f;
  p = p * z * s;
  p = p + s;
  if(big) { p = p + p; p = 1.5707963267948966192f - p; }
  return (x < 0.0f) ? -p : p;
}
RT_HD float rt_acos(float x)
{
  if(!(rt_abs(x) <= 1.0f)) return rt_u2f(0x7fc00000u);
  if(x < -0.5f) return 3.14159265358979323846f - 2.0f * rt_asin(rt_sqrt(0.5f * (1.0f + x)));
  if(x > 0.5f) return 2.0f * rt_asin(rt_sqrt(0.5f * (1.0f - x)));
  return 1.5707963267948966192f - rt_asin(x);
}

/* atan for x >= 0 */
RT_HD float rt_atan_pos(float x)
{
  float y;
  if(x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.0f / x); }
  else if(x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.0f) / (x + 1.0f); }
  else { y = 0.0f; }
  float z = x * x;
  float p = 8.05374449538e-2f;
  p = p * z + -1.38776856032E-1f;
  p = p * z + 1.99777106478E-1f;
  p = p * z + -3.33329491539E-1f;
  p = p * z * x;
  p = p + x;
  return y + p;
}
/* GLSL atan(y, x); atan2(0,0) := 0, NaN in => NaN */
RT_HD float rt_atan2(float y, float x)
{
  if(rt_isnan(x) || rt_isnan(y)) return rt_u2f(0x7fc00000u);
  if(x == 0.0f)
  {
    if(y == 0.0f) return 0.0f;
    return (y > 0.0f) ? 1.5707963267948966192f : -1.5707963267948966192f;
  }
  float a = rt_atan_pos(rt_abs(y / x));
  if(x < 0.0f) a = 3.14159265358979323846f - a;
  return (y < 0.0f) ? -a : a;
}

#endif /* RT_DETMATH_H */
