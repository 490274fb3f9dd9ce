// Shared device helpers for the GeoSSL hot-path kernels (gfx950 / CDNA4 only).
//
// Tile vocabulary used by every dense kernel in this directory: a wave (64 lanes) owns a 32-row strip of the
// problem ("pair rows", "atom rows", "super-edge rows").  The fp32 GEMMs run on the bf16 matrix pipe through an
// exact three-way split of both operands (split.h, which also documents the MFMA operand maps); the f32 MFMA
//     A: lane l holds A[row = l&31][k = 2*kk + (l>>5)],  B: lane l holds B[k = 2*kk + (l>>5)][col = l&31]
//     C/D: acc[reg], reg in [0,16): row = (reg&3) + 8*(reg>>2) + 4*(l>>5), col = l&31      (shared by both MFMAs)
// is kept only for row-GEMM shapes off the split path (gemm.hip: k_linear).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native 16-byte vector: unlike the float4 struct it is a first-class value (arrays of it are always promoted to
// registers; an array of float4 copied to LDS as a whole struct can end up in scratch memory)
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GEOSSL_MAX_LAYERS 12
#define GEOSSL_PI_F 3.14159265358979323846f

#define GEOSSL_CHECK_LAUNCH()                  \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)

namespace geossl {

// Opt a kernel into the full 160 KB of LDS: one attribute call per (kernel, device) - the attribute belongs to the
// device's copy of the code object, so a process that drives two devices sets it on each - remembered in a small
// table keyed by the kernel's ADDRESS (instantiations of one kernel template share a pointer type, so a flag per
// template instantiation of this helper would be shared between them).  Keeps the call out of graph capture too.
inline void allow_big_lds_ptr(const void* kernel) {
  struct Seen { const void* fn; int dev; };
  static Seen seen[512];
  static int nseen = 0;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < nseen; ++i)
    if (seen[i].fn == kernel && seen[i].dev == dev) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (dev >= 0 && nseen < 512) seen[nseen++] = Seen{kernel, dev};
}
template <typename K>
inline void allow_big_lds(K kernel) { allow_big_lds_ptr(reinterpret_cast<const void*>(kernel)); }


// hipcc contracts a * b + c into an fma wherever it can (-ffp-contract=fast), and __fmul_rn / __fadd_rn are plain
// operators to it.  Where the reference's own rounding sequence matters (edge decisions of the radius graph, the
// message sum), the arithmetic goes through these helpers: the pragma clears the contract flag of their operations.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
// fl32(fl32(fl32(x*x) + fl32(y*y)) + fl32(z*z))
__device__ __forceinline__ float norm2_rn(float x, float y, float z) {
#pragma clang fp contract(off)
  const float xx = x * x, yy = y * y, zz = z * z;
  return (xx + yy) + zz;
}

// Block barrier for kernels whose waves exchange data through LDS only: this wave's LDS traffic done, then s_barrier.
// __syncthreads() also waits for every outstanding GLOBAL request of the wave (vmcnt(0), part of its fence), which
// ends the flight of requests issued one tile ahead and makes the block wait for the acknowledgement of every store.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Largest value of a wave (non-negative inputs; result uniform): four DPP steps inside the 16-lane rows (quad
// permutes, half mirror, mirror - vector-ALU speed, where a __shfl_xor ladder is six dependent LDS-crossbar round trips),
// the four rows combined through scalar registers.
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)));
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ int wave_max_i32(int v) {  // the same for signed integers
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int c_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ShiftedSoftplus (schnet.py:210-216): F.softplus(x) (beta 1, threshold 20) - fp32(log 2).
// softplus(x) = max(x,0) + log1p(exp(-|x|)); above torch's threshold 20 the log1p term (< 2.1e-9) is below the
// fp32 resolution of x, so the identity branch is reproduced without a compare.  exp/log run on the hardware
// v_exp_f32 / v_log_f32 (argument magnitude <= ~17 where it matters: absolute error < 1e-7), with a short
// series for small exp(-|x|) where 1+z would lose the low bits.
#define GEOSSL_SSP_SHIFT 0.693147182464599609375f  // float(torch.log(torch.tensor(2.0)))
// exp(x) for x <= ~0.7 on the raw v_exp_f32 (2^x): no range reduction or denormal fix-up code (results below the
// normal range flush to 0, which is what every caller wants), no branches
__device__ __forceinline__ float exp_neg(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
// Branch-free: the library logf / a ternary around it compile to an exec-masked branch per element (plus the
// s_nop padding of the exec hazards) - 64 of them per lane and row block in the filter kernels.  1 + z is in (1, 2],
// so the raw v_log_f32 (log2, ~1 ulp) needs no denormal handling, and no log1p series is needed either: the
// log term is added to max(x, 0) - log 2, so only its ABSOLUTE error (< 1e-7) matters - the same as in the
// reference, which also forms log1p(exp(x)) - log 2 in fp32.
__device__ __forceinline__ float ssp(float x) {
  const float z = exp_neg(-fabsf(x));
  const float l = __builtin_amdgcn_logf(1.0f + z);  // log2(1 + exp(-|x|))
  return fmaf(l, 0.693147180559945309417f, fmaxf(x, 0.0f)) - GEOSSL_SSP_SHIFT;
}
// d ssp / dx = sigmoid(x), recovered from the saved output t = ssp(x):
// exp(-softplus(x)) = 1 - sigmoid(x)  =>  sigmoid(x) = 1 - 0.5*exp(-t)   (0.5 = exp(-log 2))
__device__ __forceinline__ float dssp_from_out(float t) { return 1.0f - 0.5f * exp_neg(-t); }

// A wave's C-layout 32x32 block (16 accumulator registers: lane = column, register = row) transposed through a
// wave-private 16x32 LDS stage (2 KB) in two halves, handing each lane 16-byte row pieces:
// consume(row in [0,32), col4 in {0,4,..,28}, float4).  Narrow 4-byte global accesses are issue-bound on this chip,
// so epilogues work on these pieces.  Used by the f32-MFMA row GEMM (gemm.hip: k_linear, shapes off the split path).
template <class ValFn, class ConsumeFn>
__device__ __forceinline__ void transpose_c_block(float* stage, int lane, ValFn val, ConsumeFn consume) {
  const int j = lane & 31, kh = lane >> 5;
  const int rr = lane >> 3, c4 = lane & 7;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = 8 * half + q;
      const int row16 = (r & 3) + 8 * ((r >> 2) & 1) + 4 * kh;
      stage[row16 * 32 + j] = val(r);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row16 = 8 * u + rr;
      const float4 v = *reinterpret_cast<const float4*>(stage + row16 * 32 + 4 * c4);
      consume(16 * half + row16, 4 * c4, v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// Keep a loaded value opaque to the optimiser.  `ok ? p[i] : 0` (and a load followed by a select on the same
// condition) is compiled to a predicated branch around the load with a full s_waitcnt vmcnt(0) per element, which
// serialises a batch of independent requests; load from a clamped (always valid) address, pin the value with this,
// then select.
__device__ __forceinline__ float pin(float v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ float4 pin(float4 v) {
  asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
  return v;
}

// sum_b p[b * stride], b in [b0, b1), in index order with a compensation term (Kahan): the partials of a weight gradient
// can cancel to a small fraction of their magnitudes, and a plain running fp32 sum over a few hundred of them then
// loses 3-4 digits of the RESULT (seen on PaiNN's filter_net.weight with ragged molecules: 7e-4 against fp64, while a
// blocked fp32 sum is at 1e-6).  Still a fixed order: bit-reproducible.
__device__ __forceinline__ float kahan_sum_strided(const float* __restrict__ p, int b0, int b1, int stride) {
#pragma clang fp reassociate(off) contract(off)
  float s = 0.0f, c = 0.0f;
#pragma unroll 4
  for (int b = b0; b < b1; ++b) {
    const float y = p[(size_t)b * stride] - c;
    const float t = s + y;
    c = (t - s) - y;
    s = t;
  }
  return s;
}

// Row count of a launch whose grid and buffers were sized for a CAPACITY (a captured HIP graph replayed on batches of
// different sizes, pretrain_GeoSSL.StepGraphs): `cap` is the by-value count the launch was built with, `dyn` (nullable)
// points at the batch's real count in device memory.  Rows at and past the real count do not exist: never read, never
// written, their blocks leave zero partial sums.
__device__ __forceinline__ int dyn_count(int cap, const int32_t* __restrict__ dyn) {
  return dyn != nullptr ? min(cap, *dyn) : cap;  // (uniform address: a scalar load)
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace geossl
