// exp_ldexp_probe.hip — is v_ldexp_f32 a bit-exact stand-in for the two power-of-two multiplies that end rt_exp (include/rt_detmath.h)?
// Every one of the 2^32 float bit patterns goes through rt_exp with either ending on the GPU AND through the host's rt_exp for a sample;
// prints the mismatch counts.  (measurement tool, round 4; on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -I include scripts/probe/exp_ldexp_probe.hip -o /tmp/exp_probe && /tmp/exp_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#define RT_EXP_PROBE 1
#include "rt_detmath.h"

__device__ float expMul(float x)     // the contract as written: (p * 2^a) * 2^b
{
  if(rt_isnan(x)) return x;
  if(x > 88.72283905206835f) return rt_u2f(0x7f800000u);
  if(x < -87.33654475055310f) return 0.0f;
  float z = rt_floor(rt_fma(x, 1.44269504088896341f, 0.5f));
  int n = (int)z;
  x = rt_fma(z, -0.693359375f, x); x = rt_fma(z, 2.12194440e-4f, x);
  float zz = x * x, p = 1.9875691500E-4f;
  p = rt_fma(p, x, 1.3981999507E-3f); p = rt_fma(p, x, 8.3334519073E-3f); p = rt_fma(p, x, 4.1665795894E-2f);
  p = rt_fma(p, x, 1.6666665459E-1f); p = rt_fma(p, x, 5.0000001201E-1f); p = rt_fma(p, zz, x); p = p + 1.0f;
  int a = n >> 1, b = n - a;
  return (p * rt_pow2i(a)) * rt_pow2i(b);
}
__device__ float expLdexp(float x)
{
  if(rt_isnan(x)) return x;
  if(x > 88.72283905206835f) return rt_u2f(0x7f800000u);
  if(x < -87.33654475055310f) return 0.0f;
  float z = rt_floor(rt_fma(x, 1.44269504088896341f, 0.5f));
  int n = (int)z;
  x = rt_fma(z, -0.693359375f, x); x = rt_fma(z, 2.12194440e-4f, x);
  float zz = x * x, p = 1.9875691500E-4f;
  p = rt_fma(p, x, 1.3981999507E-3f); p = rt_fma(p, x, 8.3334519073E-3f); p = rt_fma(p, x, 4.1665795894E-2f);
  p = rt_fma(p, x, 1.6666665459E-1f); p = rt_fma(p, x, 5.0000001201E-1f); p = rt_fma(p, zz, x); p = p + 1.0f;
  return __builtin_ldexpf(p, n);
}
__global__ void k_all(unsigned long long* bad, uint32_t* firstBad)
{
  const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
  unsigned long long mine = 0;
  for(uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const float x = rt_u2f(uint32_t(i));
    const uint32_t a = rt_f2u(expMul(x)), b = rt_f2u(expLdexp(x));
    if(a != b) { mine++; atomicMin(firstBad, uint32_t(i)); }
  }
  if(mine) atomicAdd(bad, mine);
}
__global__ void k_sample(const uint32_t* in, uint32_t* out, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) out[i] = rt_f2u(expLdexp(rt_u2f(in[i])));
}
int main()
{
  unsigned long long* dBad; uint32_t* dFirst; unsigned long long bad = 0; uint32_t first = 0xffffffffu;
  hipMalloc(&dBad, 8); hipMalloc(&dFirst, 4); hipMemcpy(dBad, &bad, 8, hipMemcpyHostToDevice); hipMemcpy(dFirst, &first, 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_all, dim3(256 * 16), dim3(256), 0, 0, dBad, dFirst);
  hipDeviceSynchronize();
  hipMemcpy(&bad, dBad, 8, hipMemcpyDeviceToHost); hipMemcpy(&first, dFirst, 4, hipMemcpyDeviceToHost);
  printf("device: multiplies vs ldexp over all 2^32 inputs: %llu mismatching (first 0x%08x)\n", bad, first);
  // host's rt_exp against the device's ldexp ending: every 251st pattern + the neighbourhood of the denormal results
  std::vector<uint32_t> in;
  for(uint64_t i = 0; i < (1ull << 32); i += 251) in.push_back(uint32_t(i));
  for(uint32_t i = 0; i < (1u << 22); i++) { float x = -87.4f + (103.3f - 87.4f) * -1.0f * float(i) / float(1u << 22); uint32_t u; memcpy(&u, &x, 4); in.push_back(u); }   // results 2^-126 .. 2^-149
  const int n = int(in.size());
  uint32_t *dIn, *dOut; hipMalloc(&dIn, size_t(n) * 4); hipMalloc(&dOut, size_t(n) * 4);
  hipMemcpy(dIn, in.data(), size_t(n) * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_sample, dim3((n + 255) / 256), dim3(256), 0, 0, dIn, dOut, n);
  std::vector<uint32_t> out(static_cast<size_t>(n)); hipMemcpy(out.data(), dOut, size_t(n) * 4, hipMemcpyDeviceToHost);
  size_t hb = 0;
  for(int i = 0; i < n; i++) { float x; memcpy(&x, &in[size_t(i)], 4); const float r = rt_exp(x); uint32_t u; memcpy(&u, &r, 4); if(u != out[size_t(i)] && !(r != r)) hb++; }
  printf("host rt_exp vs device ldexp ending: %d inputs, %zu mismatching\n", n, hb);
  return (bad || hb) ? 1 : 0;
}
