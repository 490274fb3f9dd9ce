// Which arithmetic form reproduces torch.optim.Adam's default (foreach) CUDA path bit for bit?  Variants of the three
// element-wise updates with and without fused multiply-adds; tools/probes/adam_probe.py compares each against torch.
#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
// mode bits: 0 lerp as fma, 1 addcmul as fma, 2 addcdiv as fma, 3 addcdiv as (value*m)/denom, 4 addcmul as value*(g*g)
// w1 = float(1 - beta1), w2 = float(1 - beta2) formed in DOUBLE on the host, as torch forms them (1.0f - 0.999f is
// 4.7e-5 away from float(0.001))
__global__ void k_adam_probe(float* p, const float* g, float* m, float* v, int64_t n, float lr_over_bc1, float w1,
                             float beta2, float eps, float bc2_sqrt, int mode, float w2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i], pi = p[i];
    const float w = w1;
    const float d = add_rn(gi, -m[i]);
    const float mi = (mode & 1) ? fmaf(w, d, m[i]) : add_rn(m[i], mul_rn(w, d));
    const float vb = mul_rn(v[i], beta2);
    const float t = mul_rn(w2, gi);
    float vi;
    if (mode & 16) vi = (mode & 2) ? fmaf(w2, mul_rn(gi, gi), vb) : add_rn(vb, mul_rn(w2, mul_rn(gi, gi)));  // value * (g * g)
    else vi = (mode & 2) ? fmaf(t, gi, vb) : add_rn(vb, mul_rn(t, gi));                                       // (value * g) * g
    const float denom = add_rn(sqrtf(vi) / bc2_sqrt, eps);
    float pn;
    if (mode & 8) pn = add_rn(pi, mul_rn(-lr_over_bc1, mi) / denom);
    else if (mode & 4) pn = fmaf(-lr_over_bc1, mi / denom, pi);
    else pn = add_rn(pi, mul_rn(-lr_over_bc1, mi / denom));
    m[i] = mi; v[i] = vi; p[i] = pn;
  }
}
extern "C" int adam_probe(float* p, const float* g, float* m, float* v, int64_t n, float lr_over_bc1, float w1,
                          float beta2, float eps, float bc2_sqrt, int mode, float w2, hipStream_t s) {
  hipLaunchKernelGGL(k_adam_probe, dim3(512), dim3(256), 0, s, p, g, m, v, n, lr_over_bc1, w1, beta2, eps, bc2_sqrt, mode, w2);
  return (int)hipGetLastError();
}
