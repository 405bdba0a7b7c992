// microbench.hip — measured VALU issue ceiling of the chip the library runs on (rt_measure_valu_peak).
//
// The traced kernels of the frame are bound by VALU instruction issue, not by bytes, so the roofline that says
// something about them is "wave-level VALU instructions per second".  The ceiling is MEASURED, not assumed: a kernel whose
// inner loop is a chain-free block of the instruction kinds the traversal and shading code executes — v_mul_f32 / v_fma_f32 /
// v_add_f32 / v_sub_f32, v_cndmask_b32, v_cmp_*, v_and_b32 / v_add_u32 / v_sub_u32, v_cvt_f32_ubyte0, v_min3_f32 / v_max3_f32 —
// roughly in the proportions of k_direct_stage (static histogram of the code object), every result written to a
// register no instruction of the same block reads.  The launch puts `wavesPerSimd` waves on every SIMD of the chip.  Variant 1 is
// the pure v_fma_f32 loop (the usual FLOP/s peak), for comparison.
#include <hip/hip_runtime.h>
#include <cstdint>

namespace rt {

constexpr int MB_BLOCK = 32;   // VALU instructions per asm block

// One asm block = 32 independent VALU instructions.  Inputs a,b,c,m (never written); outputs r0..r15 (never read inside the block;
// the v_cndmask pair reads VCC written by a v_cmp four instructions earlier, as in compiled code).
#define MB_MIX_BLOCK                                                                                                        \
  asm volatile(                                                                                                             \
      "v_mul_f32 %0, %16, %17\n v_fma_f32 %1, %16, %17, %18\n v_add_f32 %2, %17, %18\n v_and_b32 %3, %19, %16\n"            \
      "v_cmp_gt_f32 vcc, %16, %17\n v_mul_f32 %4, %17, %18\n v_sub_f32 %5, %16, %18\n v_add_u32 %6, %19, %19\n"             \
      "v_fma_f32 %7, %17, %18, %16\n v_cndmask_b32 %8, %16, %17, vcc\n v_mul_f32 %9, %16, %18\n v_cvt_f32_ubyte0 %10, %19\n" \
      "v_min3_f32 %11, %16, %17, %18\n v_mul_f32 %12, %18, %18\n v_sub_u32 %13, %19, %16\n v_fma_f32 %14, %18, %16, %17\n"  \
      "v_mul_f32 %15, %16, %16\n v_add_f32 %0, %16, %18\n v_cndmask_b32 %1, %17, %18, vcc\n v_mul_f32 %2, %17, %17\n"       \
      "v_cmp_lt_f32 vcc, %17, %18\n v_fma_f32 %3, %16, %16, %18\n v_and_b32 %4, %19, %17\n v_mul_f32 %5, %18, %17\n"        \
      "v_max3_f32 %6, %16, %17, %18\n v_add_u32 %7, %19, %16\n v_mul_f32 %8, %16, %17\n v_cndmask_b32 %9, %18, %16, vcc\n"  \
      "v_fma_f32 %10, %17, %17, %16\n v_add_f32 %11, %16, %17\n v_mul_f32 %12, %17, %18\n v_sub_f32 %13, %18, %17\n"        \
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8), "=&v"(r9), "=&v"(r10),      \
        "=&v"(r11), "=&v"(r12), "=&v"(r13), "=&v"(r14), "=&v"(r15)                                                               \
      : "v"(a), "v"(b), "v"(c), "v"(m)                                                                                      \
      : "vcc")

#define MB_FMA_BLOCK                                                                                                        \
  asm volatile(                                                                                                             \
      "v_fma_f32 %0, %16, %17, %18\n v_fma_f32 %1, %16, %17, %18\n v_fma_f32 %2, %16, %17, %18\n v_fma_f32 %3, %16, %17, %18\n"     \
      "v_fma_f32 %4, %16, %17, %18\n v_fma_f32 %5, %16, %17, %18\n v_fma_f32 %6, %16, %17, %18\n v_fma_f32 %7, %16, %17, %18\n"     \
      "v_fma_f32 %8, %16, %17, %18\n v_fma_f32 %9, %16, %17, %18\n v_fma_f32 %10, %16, %17, %18\n v_fma_f32 %11, %16, %17, %18\n"   \
      "v_fma_f32 %12, %16, %17, %18\n v_fma_f32 %13, %16, %17, %18\n v_fma_f32 %14, %16, %17, %18\n v_fma_f32 %15, %16, %17, %18\n" \
      "v_fma_f32 %0, %17, %16, %18\n v_fma_f32 %1, %17, %16, %18\n v_fma_f32 %2, %17, %16, %18\n v_fma_f32 %3, %17, %16, %18\n"     \
      "v_fma_f32 %4, %17, %16, %18\n v_fma_f32 %5, %17, %16, %18\n v_fma_f32 %6, %17, %16, %18\n v_fma_f32 %7, %17, %16, %18\n"     \
      "v_fma_f32 %8, %17, %16, %18\n v_fma_f32 %9, %17, %16, %18\n v_fma_f32 %10, %17, %16, %18\n v_fma_f32 %11, %17, %16, %18\n"   \
      "v_fma_f32 %12, %17, %16, %18\n v_fma_f32 %13, %17, %16, %18\n v_fma_f32 %14, %17, %16, %18\n v_fma_f32 %15, %17, %16, %18\n" \
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8), "=&v"(r9), "=&v"(r10),      \
        "=&v"(r11), "=&v"(r12), "=&v"(r13), "=&v"(r14), "=&v"(r15)                                                               \
      : "v"(a), "v"(b), "v"(c), "v"(m))

template <int VARIANT>
__global__ void __launch_bounds__(256) k_valu_issue(float* out, int iters)
{
  float a = 1.0f + float(threadIdx.x) * 1e-3f, b = 0.999f, c = 1e-3f;
  uint32_t m = threadIdx.x * 2654435761u;
  float r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
  for(int i = 0; i < iters; i++) {
    if(VARIANT == 0) { MB_MIX_BLOCK; MB_MIX_BLOCK; MB_MIX_BLOCK; MB_MIX_BLOCK; }
    else { MB_FMA_BLOCK; MB_FMA_BLOCK; MB_FMA_BLOCK; MB_FMA_BLOCK; }
  }
  // keep every result alive (never true at run time)
  const float s = ((((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))) + (((r8 + r9) + (r10 + r11)) + ((r12 + r13) + (r14 + r15))));
  if(s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// wave-level VALU instructions per second with `wavesPerSimd` resident waves on every SIMD of the chip
hipError_t measureValuIssue(hipStream_t stream, int variant, int wavesPerSimd, double* waveInstPerSec, double* seconds)
{
  hipDeviceProp_t prop;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev); if(e != hipSuccess) return e;
  e = hipGetDeviceProperties(&prop, dev); if(e != hipSuccess) return e;
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * wavesPerSimd;           // 256 threads = 4 waves = one wave per SIMD of a CU
  const int iters = 4096;
  float* out = nullptr;
  e = hipMalloc(&out, size_t(blocks) * 256 * sizeof(float)); if(e != hipSuccess) return e;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto launch = [&](int it) {
    if(variant == 0) hipLaunchKernelGGL(k_valu_issue<0>, dim3(blocks), dim3(256), 0, stream, out, it);
    else hipLaunchKernelGGL(k_valu_issue<1>, dim3(blocks), dim3(256), 0, stream, out, it);
  };
  launch(64);                                       // warm-up (code load, clocks)
  launch(iters);
  (void)hipEventRecord(e0, stream);
  launch(iters);
  (void)hipEventRecord(e1, stream);
  e = hipEventSynchronize(e1);
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(out);
  if(e != hipSuccess) return e;
  const double insts = double(blocks) * 4.0 * double(iters) * 4.0 * MB_BLOCK;   // waves x iterations x blocks x instructions
  *seconds = ms * 1e-3;
  *waveInstPerSec = insts / (ms * 1e-3);
  return hipGetLastError();
}

}  // namespace rt
