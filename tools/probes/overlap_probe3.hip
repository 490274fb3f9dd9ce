// Follow-up of overlap_probe2: is the missing overlap between the MFMAs of one wave and the vector instructions of the
// OTHER wave of the SIMD head-of-line blocking by an MFMA that waits at the issue stage for the busy matrix pipe?
// The MFMA wave pads every MFMA with s_nop wait states / independent vector work of its own, or the vector wave runs at a
// raised priority.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/overlap_probe3.hip -o tools/probes/overlap_probe3 && tools/probes/overlap_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// PAD: PAD / 10 times `s_nop 7` (8 wait states each) plus one `s_nop PAD % 10` after each MFMA in the MFMA wave; PRIO: priority of the vector wave;
// OWN: plain vector instructions of the MFMA wave itself after each MFMA
template <int PAD, int PRIO, int OWN>
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  float v[8], w[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i, w[i] = v[i] + 1.0f;
  const float c = 1.0001f, d = 0.5f;
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f + i); }
  if (mf && (mode & 1)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        f32x16& a = (g & 3) == 0 ? a0 : (g & 3) == 1 ? a1 : (g & 3) == 2 ? a2 : a3;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#pragma unroll
        for (int i = 0; i < OWN; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(w[i & 7]) : "v"(c), "v"(d));
#pragma unroll
        for (int i = 0; i < PAD / 10; ++i) asm volatile("s_nop 7");
        if (PAD % 10 == 3) asm volatile("s_nop 3");
        if (PAD % 10 == 4) asm volatile("s_nop 4");
        if (PAD % 10 == 5) asm volatile("s_nop 5");
        if (PAD % 10 == 6) asm volatile("s_nop 6");
      }
    }
  }
  if (!mf && (mode & 2)) {
    if (PRIO > 0) asm volatile("s_setprio 3");
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
  }
  float r = a0[0] + a1[1] + a2[2] + a3[3];
  for (int i = 0; i < 8; ++i) r += v[i] + w[i];
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int PAD, int PRIO, int OWN>
float run(int mode, float* d) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PAD, PRIO, OWN>), dim3(256), dim3(512), 0, 0, mode, 100, d);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<PAD, PRIO, OWN>), dim3(256), dim3(512), 0, 0, mode, iters, d);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best * 1e-3f * 2.4e9f / iters;
}

template <int PAD, int PRIO, int OWN>
void line(const char* name, float* d) {
  const float m = run<PAD, PRIO, OWN>(1, d), v = run<PAD, PRIO, OWN>(2, d), b = run<PAD, PRIO, OWN>(3, d);
  printf("%-44s MFMA wave alone %6.0f | 128 v_fma wave alone %6.0f | both %6.0f (sum %6.0f)\n", name, m, v, b, m + v);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 4096);
  line<0, 0, 0>("16 MFMA back to back", d);
  line<0, 1, 0>("16 MFMA back to back, vector wave prio 3", d);
  line<3, 0, 0>("MFMA + s_nop 3", d);
  line<4, 0, 0>("MFMA + s_nop 4", d);
  line<5, 0, 0>("MFMA + s_nop 5", d);
  line<6, 0, 0>("MFMA + s_nop 6", d);
  line<6, 1, 0>("MFMA + s_nop 6, vector wave prio 3", d);
  line<10, 0, 0>("MFMA + s_nop 7", d);
  line<10, 1, 0>("MFMA + s_nop 7, vector wave prio 3", d);
  line<20, 0, 0>("MFMA + 2 x s_nop 7", d);
  line<0, 0, 4>("MFMA + 4 own v_fma", d);
  line<0, 1, 4>("MFMA + 4 own v_fma, vector wave prio 3", d);
  line<4, 0, 4>("MFMA + 4 own v_fma + s_nop 4", d);
  line<4, 1, 4>("MFMA + 4 own v_fma + s_nop 4, vector wave prio 3", d);
  line<0, 1, 2>("MFMA + 2 own v_fma, vector wave prio 3", d);
  line<5, 0, 2>("MFMA + 2 own v_fma + s_nop 5", d);
  return 0;
}
