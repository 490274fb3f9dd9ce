// Where do the waves of a 512-thread block land?  Prints, for a few blocks, the SIMD of every wave (HW_ID register).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/hwid_probe.hip -o /tmp/hwid_probe && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(unsigned* out, int spin) {
  extern __shared__ unsigned char lds[];
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = hw;
  // keep the block resident for a while so that all blocks coexist
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (spin < 0) lds[threadIdx.x] = 1;
}
__global__ void k_other(float* x, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] * 1.0001f + 1.0f;
}
int main() {
  const int blocks = 245, threads = 512, waves = threads / 64;
  unsigned* d;
  hipMalloc(&d, blocks * waves * 4);
  float* x;
  hipMalloc(&x, 1 << 24);
  hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int trial = 0; trial < 6; ++trial) {
    if (trial & 1) hipLaunchKernelGGL(k_other, dim3((1 << 22) / 256), dim3(256), 0, 0, x, 1 << 22);
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(threads), 136 * 1024, 0, d, 20000);
    std::vector<unsigned> h(blocks * waves);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    int paired = 0;
    for (int b = 0; b < blocks; ++b) {
      int cnt[4] = {};
      bool ok = true;
      for (int w = 0; w < waves; ++w) {
        const int simd = (h[b * waves + w] >> 4) & 3;
        hist[w][simd]++;
        cnt[simd]++;
      }
      for (int w = 0; w < 4; ++w)
        if (((h[b * waves + w] >> 4) & 3) != ((h[b * waves + w + 4] >> 4) & 3)) ok = false;
      paired += ok;
    }
    printf("trial %d (%s): blocks with wave w and w+4 on one SIMD: %d / %d\n", trial, (trial & 1) ? "after another kernel" : "alone", paired, blocks);
    for (int w = 0; w < waves; ++w) printf("  wave %d: simd histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("  block 0:");
    for (int w = 0; w < waves; ++w) printf(" %u", (h[w] >> 4) & 3);
    printf("   block 100:");
    for (int w = 0; w < waves; ++w) printf(" %u", (h[100 * waves + w] >> 4) & 3);
    printf("\n");
  }
  return 0;
}
