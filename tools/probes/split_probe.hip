// Bit-identity of the fused-multiply-add form of the two-piece fp16 split (split.h: split8h / split8h_scaled, on
// v_fma_mixlo/hi_f16) against the plain form (convert, convert back, subtract, convert), over random values of every
// binade fp16 reaches after scaling, values in fp16's subnormal range, rounding ties, zeros and the largest magnitudes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I geossl_amd/csrc tools/probes/split_probe.hip -o scratch/split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "split.h"
using namespace geossl;

__global__ void k(const float* x, const float* sc, uint32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v[8], w[8];
  const float s = sc[i];
  for (int e = 0; e < 8; ++e) { v[e] = x[(size_t)i * 8 + e]; w[e] = v[e] * s; }
  const Frag2 a = split8h_plain(w), b = split8h(w), c = split8h_scaled(v, s);
  // the scaled form adds +0 inside the instruction: a negative zero comes out positive - equal as an MFMA operand
  auto z = [](uint32_t w) { return ((w & 0x7FFFu) ? (w & 0xFFFFu) : 0u) | ((w & 0x7FFF0000u) ? (w & 0xFFFF0000u) : 0u); };
  uint32_t bad = 0;
  for (int q = 0; q < 4; ++q) {
    bad |= (a.h[q] != b.h[q]) | ((a.l[q] != b.l[q]) << 1) | ((z(a.h[q]) != z(c.h[q])) << 2) | ((z(a.l[q]) != z(c.l[q])) << 3);
  }
  out[i] = bad;
}

int main() {
  const int n = 1 << 22;
  std::vector<float> x((size_t)n * 8), s(n);
  srand(7);
  auto rnd = [] { return (float)rand() / (float)RAND_MAX; };
  for (int i = 0; i < n; ++i) {
    const int kind = i % 8;
    s[i] = ldexpf(1.0f, (rand() % 41) - 20);
    for (int e = 0; e < 8; ++e) {
      float v;
      if (kind < 4) v = (rnd() * 2 - 1) * ldexpf(1.0f, (rand() % 40) - 24) / s[i] * 16384.0f;      // all binades up to 2^14
      else if (kind == 4) v = (rnd() * 2 - 1) * ldexpf(1.0f, -(rand() % 30) - 10) / s[i];       // fp16 subnormals and below
      else if (kind == 5) {  // exact ties of the first rounding: (2m + 1) * 2^(e - 11)
        const int m = rand() % 1024, ex = (rand() % 20) - 6;
        v = ldexpf((float)(2 * (1024 + m) + 1), ex - 11) / s[i];
      } else if (kind == 6) v = (rand() % 3 == 0) ? 0.0f : ((rand() & 1) ? -0.0f : 65000.0f * rnd() / s[i]);
      else v = ldexpf(1.0f + ldexpf((float)(rand() % (1 << 23)), -23), (rand() % 30) - 15) * ((rand() & 1) ? -1.f : 1.f) / s[i];
      x[(size_t)i * 8 + e] = v;
    }
  }
  float *dx, *ds; uint32_t* dout;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&ds, s.size() * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(ds, s.data(), s.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dout, n);
  std::vector<uint32_t> out(n);
  if (hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAILED to run\n"); return 2; }
  long bad[4] = {0, 0, 0, 0};
  int first = -1;
  for (int i = 0; i < n; ++i) {
    for (int b = 0; b < 4; ++b) bad[b] += (out[i] >> b) & 1;
    if (out[i] && first < 0) first = i;
  }
  printf("groups %d: mismatches  mix.h %ld  mix.l %ld  scaled.h %ld  scaled.l %ld\n", n, bad[0], bad[1], bad[2], bad[3]);
  if (first >= 0) {
    printf("first bad group %d (kind %d) scale %g values:", first, first % 8, s[first]);
    for (int e = 0; e < 8; ++e) printf(" %a", x[(size_t)first * 8 + e]);
    printf("\n");
  }
  return (bad[0] | bad[1] | bad[2] | bad[3]) ? 1 : 0;
}
