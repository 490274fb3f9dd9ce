// Which vector instructions of ANOTHER wave on the same SIMD overlap with bf16 MFMAs?  (follow-up of overlap_probe.hip,
// whose FMA stream the compiler had SLP-packed into v_pk_fma_f32.)  Instruction kinds are pinned with inline asm.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/overlap_probe2.hip -o tools/probes/overlap_probe2 && tools/probes/overlap_probe2
// One 512-thread block per CU: waves 0-3 (one per SIMD) run MFMAs, waves 4-7 (same SIMDs) run the vector stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void vec8(float (&v)[8], f32x2 (&p)[4], float c, float d) {
  if (KIND == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
  } else if (KIND == 1) {  // 4 packed = 8 flops-equivalents
    f32x2 cc = {c, c}, dd = {d, d};
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(cc), "v"(dd));
  } else if (KIND == 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
  } else if (KIND == 3) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned r;
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v[i]), "v"(c));
      asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(r));
    }
  } else if (KIND == 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
  }
}

template <int KIND, int PER_GAP>
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  float r = 0.0f;
  float v[8];
  f32x2 p[4];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  for (int i = 0; i < 4; ++i) p[i] = f32x2{v[i], v[i + 4]};
  const float c = 1.0001f, d = 0.5f;
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f + i); }
  if (mode == 1 && mf) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
      }
    }
  }
  if (mode == 2 && !mf) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) vec8<KIND>(v, p, c, d);
  }
  if (mode == 3) {
    if (mf) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
        }
      }
    } else {
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) vec8<KIND>(v, p, c, d);
    }
  }
  if (mode == 4 && mf) {  // same wave: PER_GAP groups of 8 per 8 MFMAs -> PER_GAP instructions per MFMA gap
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          f32x16& a = (g & 3) == 0 ? a0 : (g & 3) == 1 ? a1 : (g & 3) == 2 ? a2 : a3;
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
          if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < PER_GAP; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(c), "v"(d));
          } else if (KIND == 1) {
            f32x2 cc = {c, c}, dd = {d, d};
#pragma unroll
            for (int i = 0; i < PER_GAP; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 3]) : "v"(cc), "v"(dd));
          } else if (KIND == 2) {
#pragma unroll
            for (int i = 0; i < PER_GAP; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
          } else {
#pragma unroll
            for (int i = 0; i < PER_GAP; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(c));
          }
        }
      }
    }
  }
  r = a0[0] + a1[1] + a2[2] + a3[3];
  for (int i = 0; i < 8; ++i) r += v[i];
  for (int i = 0; i < 4; ++i) r += p[i][0] + p[i][1];
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KIND, int PER_GAP>
float run(int mode, float* d) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, PER_GAP>), dim3(256), dim3(512), 0, 0, mode, 100, d);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, PER_GAP>), dim3(256), dim3(512), 0, 0, mode, iters, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best * 1e-3f * 2.4e9f / iters;  // cycles per iteration at a nominal 2.4 GHz
}

template <int KIND>
void kind(const char* name, int per_iter, float* d) {
  const float m = run<KIND, 1>(1, d), v = run<KIND, 1>(2, d), b = run<KIND, 1>(3, d);
  printf("%-22s 16 MFMA alone %6.0f | %3d vec alone %6.0f | two waves of a SIMD %6.0f (sum %6.0f, max %6.0f)\n", name, m,
         per_iter, v, b, m + v, m > v ? m : v);
  const float g2 = run<KIND, 2>(4, d), g4 = run<KIND, 4>(4, d), g5 = run<KIND, 5>(4, d), g6 = run<KIND, 6>(4, d),
              g8 = run<KIND, 8>(4, d);
  printf("%-22s one wave, 16 MFMA with k per gap: k=2 %6.0f  k=4 %6.0f  k=5 %6.0f  k=6 %6.0f  k=8 %6.0f\n", name, g2, g4, g5, g6, g8);
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  kind<0>("v_fma_f32", 128, d);
  kind<1>("v_pk_fma_f32", 64, d);
  kind<2>("v_exp_f32", 128, d);
  kind<3>("v_cvt_pk_bf16 + v_and", 256, d);
  kind<4>("v_add_u32", 128, d);
  return 0;
}
