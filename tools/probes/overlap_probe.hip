// Do bf16 MFMAs of one wave overlap with vector work of ANOTHER wave on the same SIMD?  (and of the same wave?)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/overlap_probe.hip -o tools/probes/overlap_probe && tools/probes/overlap_probe
// One 512-thread block per CU: waves 0-3 (one per SIMD) run MFMAs, waves 4-7 (same SIMDs) run FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  float r = 0.0f;
  if (mf && (mode & 1)) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
      }
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  }
  if (!mf && (mode & 2)) {
    float v0 = threadIdx.x, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f, v4 = 0.5f, v5 = 0.25f, v6 = 4.0f, v7 = 5.0f;
    const float c = 1.0001f, d = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {  // 128 independent-ish FMAs per iteration (16 MFMAs x 8 passes x 4 cycles = 512 cycles)
        v0 = fmaf(v0, c, d); v1 = fmaf(v1, c, d); v2 = fmaf(v2, c, d); v3 = fmaf(v3, c, d);
        v4 = fmaf(v4, c, d); v5 = fmaf(v5, c, d); v6 = fmaf(v6, c, d); v7 = fmaf(v7, c, d);
      }
    }
    r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  if ((mode & 4) && mf) {  // same-wave interleave: MFMAs and FMAs in ONE wave (waves 4-7 idle)
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f + i); }
    float v0 = threadIdx.x, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f, v4 = 0.5f, v5 = 0.25f, v6 = 4.0f, v7 = 5.0f;
    const float c = 1.0001f, d = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        v0 = fmaf(v0, c, d); v1 = fmaf(v1, c, d); v2 = fmaf(v2, c, d); v3 = fmaf(v3, c, d);
        v4 = fmaf(v4, c, d); v5 = fmaf(v5, c, d); v6 = fmaf(v6, c, d); v7 = fmaf(v7, c, d);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        v0 = fmaf(v0, c, d); v1 = fmaf(v1, c, d); v2 = fmaf(v2, c, d); v3 = fmaf(v3, c, d);
        v4 = fmaf(v4, c, d); v5 = fmaf(v5, c, d); v6 = fmaf(v6, c, d); v7 = fmaf(v7, c, d);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        v0 = fmaf(v0, c, d); v1 = fmaf(v1, c, d); v2 = fmaf(v2, c, d); v3 = fmaf(v3, c, d);
        v4 = fmaf(v4, c, d); v5 = fmaf(v5, c, d); v6 = fmaf(v6, c, d); v7 = fmaf(v7, c, d);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
        v0 = fmaf(v0, c, d); v1 = fmaf(v1, c, d); v2 = fmaf(v2, c, d); v3 = fmaf(v3, c, d);
        v4 = fmaf(v4, c, d); v5 = fmaf(v5, c, d); v6 = fmaf(v6, c, d); v7 = fmaf(v7, c, d);
      }
    }
    r = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}
int main() {
  float* d;
  hipMalloc(&d, 4096);
  const int iters = 20000;
  const char* names[] = {"", "MFMA waves only (16 MFMA/iter)", "FMA waves only (128 FMA/iter)", "both, different waves of a SIMD", "one wave: 16 MFMA + 128 FMA interleaved"};
  const int modes[] = {1, 2, 3, 4};
  for (int rep = 0; rep < 2; ++rep)
    for (int mi = 0; mi < 4; ++mi) {
      const int mode = modes[mi];
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, 100, d);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, d);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%-45s %8.3f ms  (%.1f cycles/iter at 2.4 GHz)\n", names[mode == 4 ? 4 : mode], ms, ms * 1e-3 * 2.4e9 / iters);
    }
  return 0;
}
