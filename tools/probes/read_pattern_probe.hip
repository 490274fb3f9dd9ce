// HBM read rate of two access patterns over the same [R][128] fp32 matrices:
//   (a) the weight-gradient kernel's: lane = column, eight 4-byte loads down a column (a half-wave reads 128 contiguous bytes)
//   (b) 16-byte row pieces, fully coalesced (8 lanes per 128-byte row segment)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/read_pattern_probe.hip -o tools/probes/read_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void k_col(const float* __restrict__ A, int R, int nmat, float* out) {
  // block = 4 waves; wave w reads column block w (32 columns) of 32-row tiles
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, kh = lane >> 5;
  float s = 0.0f;
  const int ntiles = R / 32;
  for (int z = 0; z < nmat; ++z) {
    const float* M = A + (size_t)z * R * 128;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e)
        v[e] = __builtin_nontemporal_load(M + (size_t)(32 * t + 16 * (e >> 3) + 8 * kh + (e & 7)) * 128 + 32 * wave + j);
#pragma unroll
      for (int e = 0; e < 16; ++e) s += v[e];
    }
  }
  if (s == 1234.5f) out[0] = s;
}
__global__ __launch_bounds__(256, 2) void k_row(const float* __restrict__ A, int R, int nmat, float* out) {
  const int tid = threadIdx.x;
  float s = 0.0f;
  const int ntiles = R / 32;
  for (int z = 0; z < nmat; ++z) {
    const f32x4* M = reinterpret_cast<const f32x4*>(A + (size_t)z * R * 128);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      f32x4 v[4];  // 32 rows x 128 cols = 1024 float4 per tile / 256 threads
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __builtin_nontemporal_load(M + (size_t)t * 1024 + 256 * e + tid);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += v[e].x + v[e].y + v[e].z + v[e].w;
    }
  }
  if (s == 1234.5f) out[0] = s;
}
// (c) the aggregation's: one wave per contiguous run of 153 rows, one 512-byte row (8 bytes per lane) per request, RING in flight
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int RING>
__global__ __launch_bounds__(64, 2) void k_agg(const float* __restrict__ A, int nruns, float* out) {
  const int lane = threadIdx.x;
  const float* base = A + (size_t)blockIdx.x * 153 * 128 + 2 * lane;
  f32x2 ring[RING], acc = {0.0f, 0.0f};
#pragma unroll
  for (int q = 0; q < RING; ++q) ring[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(base + (size_t)q * 128));
#pragma unroll
  for (int q = 0; q < 153; ++q) {
    const f32x2 w = ring[q % RING];
    if (q + RING < 153) ring[q % RING] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(base + (size_t)(q + RING) * 128));
    acc += w;
    asm volatile("" : "+v"(acc.x), "+v"(acc.y) : : "memory");
  }
  if (acc.x == 1234.5f) out[0] = acc.y;
}
int main() {
  {
    const int nruns = 2048 * 6;  // six layers' worth: 963 MB
    float *A, *out;
    hipMalloc(&A, (size_t)nruns * 153 * 128 * 4);
    hipMalloc(&out, 16);
    hipMemset(A, 0, (size_t)nruns * 153 * 128 * 4);
    for (int which = 0; which < 3; ++which) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 6; ++l) {  // six launches of 2048 waves, as in a step
          const float* Al = A + (size_t)l * 2048 * 153 * 128;
          if (which == 0) hipLaunchKernelGGL(k_agg<16>, dim3(2048), dim3(64), 0, 0, Al, 2048, out);
          else if (which == 1) hipLaunchKernelGGL(k_agg<32>, dim3(2048), dim3(64), 0, 0, Al, 2048, out);
          else hipLaunchKernelGGL(k_agg<48>, dim3(2048), dim3(64), 0, 0, Al, 2048, out);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("aggregation pattern, ring %d: %.1f us per launch of 2048 waves x 153 rows  %.2f TB/s\n", which == 0 ? 16 : (which == 1 ? 32 : 48),
             best / 6 * 1e3, (double)nruns * 153 * 128 * 4 / best / 1e9);
    }
    hipFree(A);
  }
  const int R = 36864, nmat = 40;  // 40 matrices of 18.9 MB = 755 MB
  float *A, *out;
  hipMalloc(&A, (size_t)nmat * R * 128 * 4);
  hipMalloc(&out, 16);
  hipMemset(A, 0, (size_t)nmat * R * 128 * 4);
  for (int grid : {512, 1024, 2048}) {
    for (int which = 0; which < 2; ++which) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k_col, dim3(grid), dim3(256), 0, 0, A, R, nmat, out);
        else hipLaunchKernelGGL(k_row, dim3(grid), dim3(256), 0, 0, A, R, nmat, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("grid %4d  %s  %.3f ms  %.2f TB/s\n", grid, which == 0 ? "4-byte column loads" : "16-byte row pieces ", ms, (double)nmat * R * 128 * 4 / ms / 1e9);
    }
  }
  return 0;
}
