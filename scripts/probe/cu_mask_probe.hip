// cu_mask_probe.hip — which compute units does bit i of a hipExtStreamCreateWithCUMask mask select on this part?
// (measurement tool, round 4: CU-partitioned streams; build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 scripts/probe/cu_mask_probe.hip -o /tmp/cu_probe && /tmp/cu_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
#include <map>

__global__ void k_where(uint32_t* out, int spin)
{
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the workgroup resident for a while so that the launch spreads over every CU the mask allows
  uint64_t t0 = wall_clock64();
  while(wall_clock64() - t0 < uint64_t(spin)) {}
  if(threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
}

static std::set<uint32_t> run(hipStream_t s, uint32_t* d, int blocks)
{
  std::vector<uint32_t> h(size_t(blocks) * 2);
  hipLaunchKernelGGL(k_where, dim3(blocks), dim3(64), 0, s, d, 20000);
  hipStreamSynchronize(s);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::set<uint32_t> cus;
  for(int b = 0; b < blocks; b++) {
    const uint32_t hw = h[size_t(b) * 2], xcc = h[size_t(b) * 2 + 1] & 0xf;
    const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
  }
  return cus;
}

int main()
{
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
  const int blocks = 8192;
  uint32_t* d; hipMalloc(&d, size_t(blocks) * 8);
  hipStream_t s0; hipStreamCreate(&s0);
  auto all = run(s0, d, blocks);
  printf("no mask: %zu distinct (xcc,se,sh,cu)\n", all.size());
  std::map<uint32_t, int> perXcc;
  for(uint32_t c : all) perXcc[c >> 12]++;
  for(auto& kv : perXcc) printf("  xcc %u: %d CUs\n", kv.first, kv.second);
  const int words = (p.multiProcessorCount + 31) / 32;
  // single bits 0..63, then strides
  for(int bit = 0; bit < 40; bit++) {
    std::vector<uint32_t> m(size_t(words), 0u); m[size_t(bit / 32)] |= 1u << (bit % 32);
    hipStream_t s; if(hipExtStreamCreateWithCUMask(&s, uint32_t(words), m.data()) != hipSuccess) { printf("bit %d: create failed\n", bit); continue; }
    auto c = run(s, d, 512);
    printf("bit %3d ->", bit);
    for(uint32_t x : c) printf(" xcc%u.se%u.sh%u.cu%u", x >> 12, (x >> 8) & 0xf, (x >> 4) & 0xf, x & 0xf);
    printf("\n");
    hipStreamDestroy(s);
  }
  for(int nb : {8, 16, 32, 64, 128, 192, 256}) {
    std::vector<uint32_t> m(size_t(words), 0u);
    for(int b = 0; b < nb && b < words * 32; b++) m[size_t(b / 32)] |= 1u << (b % 32);
    hipStream_t s; if(hipExtStreamCreateWithCUMask(&s, uint32_t(words), m.data()) != hipSuccess) { printf("first %d bits: create failed\n", nb); continue; }
    auto c = run(s, d, blocks);
    std::map<uint32_t, int> px; for(uint32_t x : c) px[x >> 12]++;
    printf("first %3d bits -> %zu CUs:", nb, c.size());
    for(auto& kv : px) printf(" xcc%u:%d", kv.first, kv.second);
    printf("\n");
    hipStreamDestroy(s);
  }
  return 0;
}
