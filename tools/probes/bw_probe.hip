// Streaming rates of this chip as simple kernels see them: read-only (sum), write-only (fill), copy - 16-byte accesses,
// grid-stride, several sizes.  hipcc --offload-arch=gfx950 -O3 -o /tmp/bw_probe tools/probes/bw_probe.hip && /tmp/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_read(const f32x4* __restrict__ p, size_t n, float* out) {
  f32x4 s = {0, 0, 0, 0};
  const size_t step = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * step < n; i += 4 * step) {   // four requests in flight per lane
    const f32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + step),
                c = __builtin_nontemporal_load(p + i + 2 * step), d = __builtin_nontemporal_load(p + i + 3 * step);
    s += (a + b) + (c + d);
  }
  for (; i < n; i += step) s += p[i];
  if (s.x + s.y + s.z + s.w == 123.456f) *out = 1.0f;
}
__global__ void k_write(f32x4* __restrict__ p, size_t n, float v) {
  const f32x4 x = {v, v, v, v};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = x;
}
__global__ void k_copy(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main() {
  const size_t sizes[] = {160u << 20, 963u << 20, 2048u << 20};
  float* out;
  hipMalloc(&out, 4);
  for (size_t bytes : sizes) {
    f32x4 *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    const size_t n = bytes / 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks : {1024, 2048, 4096, 8192}) {
      float ms[3];
      for (int which = 0; which < 3; ++which) {
        for (int rep = 0; rep < 3; ++rep) {
          if (rep == 1) hipEventRecord(e0);
          for (int it = 0; it < (rep == 0 ? 2 : 10); ++it) {
            if (which == 0) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, out);
            if (which == 1) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n, 1.0f);
            if (which == 2) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n);
          }
          if (rep == 1) { hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[which], e0, e1); ms[which] /= 10; break; }
        }
      }
      printf("%5zu MB, %4d blocks: read %.2f TB/s, write %.2f TB/s, copy %.2f TB/s (read + write bytes)\n", bytes >> 20, blocks,
             bytes / ms[0] / 1e9, bytes / ms[1] / 1e9, 2.0 * bytes / ms[2] / 1e9);
    }
    hipFree(a);
    hipFree(b);
  }
  return 0;
}
