// Does `v_pk_mul_f32 vD, vA, vB op_sel:[0,1]` (both results read the HIGH half of src1) always return lo = A.lo * B.hi?
// Round 6: the mu-zero form of k_painn_fwd_mma lost such low results in lanes 48-63 now and then (a term of a sum missing,
// as if the product were 0) when two waves shared a SIMD; never with one block per CU, never built without packed ops.
//   hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize tools/probes/pk_opsel_probe.hip -o tools/probes/pk_opsel_probe
//   tools/probes/pk_opsel_probe
// One 512-thread block per CU.  Waves 0-3 (one per SIMD) issue the packed multiply in a loop and compare both halves
// with scalar products; waves 4-7 (the same SIMDs) run a disturber: nothing / MFMAs / scalar FMAs / LDS reads / packed
// multiplies / v_permlane32_swap / v_readlane.  test form 1 feeds the packed multiply the way the kernel did: its src0 pair is [a register written by
// v_mov just before, the second dword of a ds_read_b128].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k(int test, int disturb, int iters, unsigned long long* bad_lo, unsigned long long* bad_hi,
                                         unsigned long long* lanes, float* sink) {
  __shared__ __attribute__((aligned(16))) float tab[8][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8 * 256; i += 512) (&tab[0][0])[i] = 0.25f + 0.001f * (i % 251);
  __syncthreads();
  float r = 0.0f;
  if (wave < 4) {
    unsigned long long nlo = 0, nhi = 0, mask = 0;
    float a0 = 1.0f + 0.01f * lane, a1 = 2.0f - 0.01f * lane, b0 = 0.5f + 0.003f * lane, b1 = 1.5f - 0.002f * lane;
    for (int it = 0; it < iters; ++it) {
      f32x2 d;
      if (test == 0) {
        const f32x2 a = {a0, a1}, b = {b0, b1};
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        if (d.x != a0 * b1) { ++nlo; mask |= 1ull << lane; }
        if (d.y != a1 * b1) ++nhi;
      } else {  // the kernel's shape: two b128 table reads (uniform per half), d1[e] moved next to d2[e], then the product
        const float* p = &tab[wave][(it & 7) * 16 + 8 * (lane >> 5)];
        f32x2 b = {b0, b1};
        float lo_a, d2_1;
        asm volatile(
            "ds_read_b128 v[40:43], %4\n"
            "ds_read_b128 v[44:47], %4 offset:16\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_mov_b32 %1, v40\n"                           // d1[0] (copy for the check)
            "v_mov_b32 %2, v45\n"                           // d2[1] (copy for the check)
            "v_mov_b32 v44, v40\n"                          // the pair [d1[0], d2[1]] formed IN the second read's registers
            "v_pk_mul_f32 %0, v[44:45], %3 op_sel:[0,1]\n"  // (dword 0 overwritten by the v_mov, dword 1 as loaded)
            : "=&v"(d), "=&v"(lo_a), "=&v"(d2_1)
            : "v"(b), "v"((unsigned)(size_t)p)
            : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        if (d.x != lo_a * b1) { ++nlo; mask |= 1ull << lane; }
        if (d.y != d2_1 * b1) ++nhi;
      }
      a0 += 0.125f; b1 += 0.0625f;
      r += d.x + d.y;
    }
    atomicAdd(bad_lo, nlo);
    atomicAdd(bad_hi, nhi);
    atomicOr(lanes, mask);
  } else {
    if (disturb == 1) {
      f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
      f16x8 x, y;
      for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(lane * 0.001f + i); y[i] = (_Float16)(1.0f + i); }
      for (int it = 0; it < iters / 8; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c3, 0, 0, 0);
      }
      r = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (disturb == 2) {
      float v0 = lane, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f;
      for (int it = 0; it < iters * 2; ++it) {
        v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
      }
      r = v0 + v1 + v2 + v3;
    } else if (disturb == 3) {
      for (int it = 0; it < iters; ++it) {
        const f32x4 q = *reinterpret_cast<const volatile f32x4*>(&tab[wave][((it * 7 + lane) & 63) * 4]);
        r += q[0] + q[3];
      }
    } else if (disturb == 5) {  // the kernel's half exchange
      unsigned u = lane * 2654435761u, w = u ^ 0x55555555u;
      for (int it = 0; it < iters * 2; ++it) {
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(w));
        u += 1;
      }
      r = (float)(u ^ w);
    } else if (disturb == 6) {  // lane reads into SGPRs and selects under SGPR masks (the kernel's group codes)
      int code = lane * 37 + 1, acc = 0;
      for (int it = 0; it < iters; ++it) {
        const int c0 = __builtin_amdgcn_readlane(code, it & 7), c1 = __builtin_amdgcn_readlane(code, (it + 3) & 7);
        acc += (c0 & 1) ? c1 : -c1;
        code = code * 3 + acc;
      }
      r = (float)acc;
    } else if (disturb == 4) {
      f32x2 a = {1.0f + lane, 2.0f}, b = {0.5f, 1.0001f}, d;
      for (int it = 0; it < iters * 2; ++it) {
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        a.x = d.y;
      }
      r = a.x;
    }
  }
  if (r == 123.456f) sink[0] = r;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200000;
  unsigned long long *d, h[3];
  float* sink;
  if (hipMalloc(&d, 3 * sizeof(*d)) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  const char* dn[] = {"idle", "mfma", "scalar fma", "lds reads", "packed mul", "permlane32", "readlane"};
  for (int test = 0; test < 2; ++test)
    for (int disturb = 0; disturb < 7; ++disturb) {
      (void)hipMemset(d, 0, 3 * sizeof(*d));
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, test, disturb, iters, d, d + 1, d + 2, sink);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("test form %d, second wave: %-11s  packed multiplies %.3g  wrong lo %llu  wrong hi %llu  lanes %016llx\n", test,
             dn[disturb], (double)iters * 256 * 4 * 64, h[0], h[1], h[2]);
    }
  return 0;
}
