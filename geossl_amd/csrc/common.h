// Shared device helpers for the GeoSSL hot-path kernels (gfx950 / CDNA4 only).
//
// Tile vocabulary used by every dense kernel in this directory:
//   * a wave (64 lanes) owns a 32-row strip of the problem ("pair rows", "atom rows",
//     "super-edge rows") and all NC*32 output columns of it;
//   * the contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32, one rounding per product,
//     bitwise a k-ordered fmaf chain) with the operand maps of the CDNA4 ISA:
//         A: lane l holds A[row = l&31][k = 2*kk + (l>>5)]
//         B: lane l holds B[k = 2*kk + (l>>5)][col = l&31]
//         C/D: acc[reg], reg in [0,16): row = (reg&3) + 8*(reg>>2) + 4*(l>>5), col = l&31
//   * the B operand (a weight matrix) lives in LDS as Bs[k][col] (row stride BS floats) so that the
//     32 lanes of a half-wave read 32 consecutive floats (conflict-free ds_read_b32);
//   * the A operand lives in a wave-private LDS tile, row-major with an XOR swizzle on the low five
//     column bits (a_idx), which makes both the C-layout write (lanes = columns) and the A-fragment
//     read (lanes = rows) conflict-free at a power-of-two row stride.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

__device__ __forceinline__ int c_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// swizzled index into a wave-private [32][AS] A tile (AS multiple of 32)
__device__ __forceinline__ int a_idx(int row, int k, int AS) { return row * AS + (k ^ (row & 31)); }

// ShiftedSoftplus (schnet.py:210-216): F.softplus(x) (beta 1, threshold 20) - fp32(log 2).
// softplus(x) = max(x,0) + log1p(exp(-|x|)); above torch's threshold 20 the log1p term (< 2.1e-9) is below the
// fp32 resolution of x, so the identity branch is reproduced without a compare.  exp/log run on the hardware
// v_exp_f32 / v_log_f32 (argument magnitude <= ~17 where it matters: absolute error < 1e-7), with a short
// series for small exp(-|x|) where 1+z would lose the low bits.
#define GEOSSL_SSP_SHIFT 0.693147182464599609375f  // float(torch.log(torch.tensor(2.0)))
__device__ __forceinline__ float ssp(float x) {
  const float z = __expf(-fabsf(x));
  const float l = z < 0.0078125f ? z * (1.0f - z * (0.5f - z * 0.33333334f)) : __logf(1.0f + z);
  return (fmaxf(x, 0.0f) + l) - GEOSSL_SSP_SHIFT;
}
// d ssp / dx = sigmoid(x), recovered from the saved output t = ssp(x):
// exp(-softplus(x)) = 1 - sigmoid(x)  =>  sigmoid(x) = 1 - 0.5*exp(-t)   (0.5 = exp(-log 2))
__device__ __forceinline__ float dssp_from_out(float t) { return 1.0f - 0.5f * __expf(-t); }

// acc[c] += A(32 x K) * B(K x 32*NC); A from a swizzled wave-private LDS tile, B from LDS/global Bs[k][col].
// Software pipelined by hand: the fragments of k-step kk+1 are requested before the MFMAs of k-step kk issue
// (hipcc places the ds_read right in front of its first use otherwise and the wave stalls on LDS latency).
// Plain (compiler-scheduled) variant for B operands read from global memory, where several k-steps of loads in
// flight matter more than the exact issue order.
template <int NC, typename BPtr>
__device__ __forceinline__ void mma_tile_gb(f32x16 (&acc)[NC], const float* At, int AS, BPtr Bs, int BS, int K2,
                                            int lane) {
  const int j = lane & 31, kh = lane >> 5;
  const float* arow = At + j * AS;
#pragma unroll 8
  for (int kk = 0; kk < K2; ++kk) {
    const int k = 2 * kk + kh;
    const float a = arow[k ^ j];
    const float* bp = Bs + (size_t)k * BS + j;
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[32 * c], acc[c], 0, 0, 0);
  }
}

template <int NC, typename BPtr>
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[NC], const float* At, int AS, BPtr Bs, int BS, int K2,
                                         int lane) {
  const int j = lane & 31, kh = lane >> 5;
  const float* arow = At + j * AS;
  float a0, a1, b0[NC], b1[NC];
  {
    a0 = arow[kh ^ j];
    const float* bp = Bs + (size_t)kh * BS + j;
#pragma unroll
    for (int c = 0; c < NC; ++c) b0[c] = bp[32 * c];
  }
  int kk = 0;
  for (; kk + 1 < K2; kk += 2) {
    {
      const int k = 2 * (kk + 1) + kh;
      a1 = arow[k ^ j];
      const float* bp = Bs + (size_t)k * BS + j;
#pragma unroll
      for (int c = 0; c < NC; ++c) b1[c] = bp[32 * c];
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the requests above the MFMAs that hide their latency
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[c], acc[c], 0, 0, 0);
    if (kk + 2 < K2) {
      const int k = 2 * (kk + 2) + kh;
      a0 = arow[k ^ j];
      const float* bp = Bs + (size_t)k * BS + j;
#pragma unroll
      for (int c = 0; c < NC; ++c) b0[c] = bp[32 * c];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[c], acc[c], 0, 0, 0);
  }
  if (kk < K2) {  // odd K2 tail
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[c], acc[c], 0, 0, 0);
  }
}

// write a wave's C-layout accumulators into its swizzled A tile (columns 32*c0 ...)
template <int NC>
__device__ __forceinline__ void acc_to_tile(const f32x16 (&acc)[NC], float* At, int AS, int lane) {
  const int col = lane & 31;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = c_row(r, lane);
      At[a_idx(row, 32 * c + col, AS)] = acc[c][r];
    }
}

// Load a torch-layout weight W[nout][ldw] (first K columns) transposed into Bs[k][n] (stride BS), zero padded
// to KP rows / NP columns.  All threads of the block participate.
__device__ __forceinline__ void load_weight_T(const float* __restrict__ W, int nout, int K, int ldw, float* Bs,
                                              int BS, int KP, int NP, int tid, int nthreads) {
  for (int i = tid; i < KP * NP; i += nthreads) {
    const int n = i / KP, k = i - n * KP;  // consecutive threads -> consecutive k (coalesced global read)
    Bs[k * BS + n] = (n < nout && k < K) ? W[(size_t)n * ldw + k] : 0.0f;
  }
}
// Load W[nrows][ldw] (first K columns) as is into Bs[n][k] (stride BS), zero padded to NRP rows / KP columns.
__device__ __forceinline__ void load_weight_N(const float* __restrict__ W, int nrows, int K, int ldw, float* Bs,
                                              int BS, int NRP, int KP, int tid, int nthreads) {
  for (int i = tid; i < NRP * KP; i += nthreads) {
    const int n = i / KP, k = i - n * KP;
    Bs[n * BS + k] = (n < nrows && k < K) ? W[(size_t)n * ldw + k] : 0.0f;
  }
}

// Store a wave's C-layout 32x32 block (16 accumulator registers: lane = column, register = row) to a row-major
// global tile with 16-byte stores.  Narrow stores are issue-bound on this chip (one 4-byte-per-lane store costs
// about as much issue time as a 16-byte one), so the block is transposed through a wave-private 16x32 LDS stage
// (2 KB) in two halves: 8 ds_write_b32 + 2 ds_read_b128 + 2 global_store_dwordx4 per half instead of 16 dword
// stores per block.  dst points at (row 0, column 0) of the 32x32 block; ld = row stride in floats (multiple of 4);
// rows >= nrows_valid are skipped.  val(r) returns the value of accumulator register r.
template <class ValFn>
__device__ __forceinline__ void store_c_block_x4(float* __restrict__ dst, size_t ld, int nrows_valid, float* stage,
                                                 int lane, ValFn val) {
  const int j = lane & 31, kh = lane >> 5;
  const int rr = lane >> 3, c4 = lane & 7;  // read side: row within an 8-row group, 16-byte chunk within the row
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = 8 * half + q;                        // accumulator register
      const int row16 = (r & 3) + 8 * ((r >> 2) & 1) + 4 * kh;  // row within this half's 16 rows
      stage[row16 * 32 + j] = val(r);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS ops complete in order
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row16 = 8 * u + rr;
      const float4 v = *reinterpret_cast<const float4*>(stage + row16 * 32 + 4 * c4);
      const int row = 16 * half + row16;
      if (row < nrows_valid) *reinterpret_cast<float4*>(dst + (size_t)row * ld + 4 * c4) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the stage is overwritten
  }
}

// Same transposition, handing each lane 16-byte row pieces: consume(row in [0,32), col4 in {0,4,..,28}, float4).
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace geossl
