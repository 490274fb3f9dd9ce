// Continuous-filter network forward for all interaction blocks (K3): InteractionBlock.mlp applied inside
// CFConv.forward, schnet.py:141-145,186-187 (with GaussianSmearing, :205-207, folded in):
//     Wf_l[p] = ( ssp( rbf(d_p) A1_l^T + b1_l ) A2_l^T + b2_l ) * C(d_p)          for every pair slot p, layer l.
//
// fp32 results on the bf16 matrix pipe (see split.h).  Both GEMMs are evaluated TRANSPOSED with the pair rows on
// the MFMA N axis (lanes) and the weights as the A operand:   t^T = A1 rbf^T,   Wf^T = A2 t^T.   A wave owns 32
// pair rows end to end (no block-level sync in the main loop):
//   * rbf^T B fragments: lane (row j, half kh) evaluates the 8 Gaussians of its k-step for its own row - every
//     exp is computed exactly once;
//   * the C layout of t^T (lane = pair row, registers = 4 consecutive hidden features per group) IS the B
//     fragment layout of the second GEMM under a fixed permutation of the contraction index that the weight
//     fragments share, so ssp + splitting happen in registers with no transposition through LDS;
//   * the same layout gives 16-byte row-major global stores of t (saved for the backward) and Wf;
//   * the weights live in LDS pre-split and pre-permuted as ready-made A fragments (1 KB per fragment, lane i at
//     byte 16 i: conflict-free ds_read_b128), 144 KB for F = 128, G <= 64; one block of 8 waves per CU.
#include "common.h"
#include "geossl_hip.h"
#include "split.h"

using namespace geossl;

namespace {

#ifdef FF_TIMING
__device__ long long ff_dbg[2 * 64 * 8 + 2 * 64];  // + wall clock (100 MHz) at every tile start
#define FF_MARK(slot)                                                                                     \
  do {                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    if (blockIdx.x == 3 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == 4) && ff_t < 64)        \
      ff_dbg[((wave == 0 ? 0 : 1) * 64 + ff_t) * 8 + (slot)] = clock64();                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  } while (0)
#else
#define FF_MARK(slot) do {} while (0)
#endif

// NMB = F/32 row blocks of the weight matrices; K1S = 16-wide k-steps of the first GEMM (>= ceil(G/16))
template <int NMB, int K1S>
__global__ __launch_bounds__(512) void k_filter_fwd(const float* __restrict__ pair_d,
                                                    const float* __restrict__ pair_c, int P,
                                                    GeosslFilterWeights w, int G,
                                                    const float* __restrict__ offset, float coeff,
                                                    float* __restrict__ Tout, float* __restrict__ Wf,
                                                    const int32_t* __restrict__ dyn_P) {
  constexpr int F = 32 * NMB, K2S = F / 16;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  u32x4* W2f = reinterpret_cast<u32x4*>(smem_raw);          // [NMB][K2S][3][64] A fragments of A2
  u32x4* W1f = W2f + NMB * K2S * 3 * 64;                    // [NMB][K1S][3][64] A fragments of A1
  float* b1s = reinterpret_cast<float*>(W1f + NMB * K1S * 3 * 64);  // [F]
  float* b2s = b1s + F;                                     // [F]
  float* offs = b2s + F;                                    // [16*K1S] Gaussian centres, zero padded
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int l = blockIdx.y;
  // ---- one-time weight formatting: split + permute into ready-made A fragments
  {
    const float* __restrict__ w2 = w.w2[l];
    for (int i = tid; i < NMB * K2S * 64; i += 512) {
      const int ln = i & 63, ks = (i >> 6) % K2S, mb = i / (64 * K2S);
      // contraction-index permutation kperm (split.h): elements 0..3 <- features 4kh.., 4..7 <- features 8+4kh..
      const float* row = w2 + (size_t)(32 * mb + (ln & 31)) * F + 16 * ks + 4 * (ln >> 5);
      const float4 lo = *reinterpret_cast<const float4*>(row), hi = *reinterpret_cast<const float4*>(row + 8);
      const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      const Frag3 f = split8(v);
      u32x4* dst = W2f + ((size_t)(mb * K2S + ks) * 3) * 64 + ln;
      dst[0] = f.h;
      dst[64] = f.m;
      dst[128] = f.l;
    }
    const float* __restrict__ w1 = w.w1[l];
    for (int i = tid; i < NMB * K1S * 64; i += 512) {
      const int ln = i & 63, ks = (i >> 6) % K1S, mb = i / (64 * K1S);
      const float* row = w1 + (size_t)(32 * mb + (ln & 31)) * G;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = 16 * ks + 8 * (ln >> 5) + e;
        v[e] = g < G ? row[g] : 0.0f;
      }
      const Frag3 f = split8(v);
      u32x4* dst = W1f + ((size_t)(mb * K1S + ks) * 3) * 64 + ln;
      dst[0] = f.h;
      dst[64] = f.m;
      dst[128] = f.l;
    }
    for (int i = tid; i < F; i += 512) {
      b1s[i] = w.b1[l][i];
      b2s[i] = w.b2[l][i];
    }
    for (int i = tid; i < 16 * K1S; i += 512) offs[i] = i < G ? offset[i] : 0.0f;
  }
  __syncthreads();
  const size_t lbase = (size_t)l * P;  // (the layer stride of T / Wf is the by-value - capacity - count)
  P = dyn_count(P, dyn_P);
  if (P <= 0) return;
  const int nrb = (P + 31) / 32;
  // ---- main loop: a wave takes 32 pair rows through both GEMMs; no block-level synchronisation
  // row blocks dealt wave-index-major: the waves that get one block more than the others are then spread one per SIMD
  // over all blocks instead of filling the first blocks (a SIMD's two waves share its matrix pipe)
#ifdef FF_TIMING
  int ff_t = -1;
#endif
  for (int rb = blockIdx.x + gridDim.x * wave; rb < nrb; rb += gridDim.x * 8) {
#ifdef FF_TIMING
    ++ff_t;
#endif
    FF_MARK(0);
#ifdef FF_TIMING
    if (blockIdx.x == 3 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == 4) && ff_t < 64)
      ff_dbg[2 * 64 * 8 + (wave == 0 ? 0 : 64) + ff_t] = wall_clock64();
#endif
    const int row = 32 * rb + j;
    const bool live = row < P;
    const float d = live ? pair_d[row] : 0.0f;
    const float cw = live ? pair_c[row] : 0.0f;
    // first GEMM, transposed: acc1[mb] = (A1 rbf^T)[32mb.., rows]; bias along the register (feature) axis
    f32x16 acc1[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(b1s + 32 * mb + 8 * q + 4 * kh);
        acc1[mb][4 * q] = b.x;
        acc1[mb][4 * q + 1] = b.y;
        acc1[mb][4 * q + 2] = b.z;
        acc1[mb][4 * q + 3] = b.w;
      }
    // rbf^T B fragments of all k-steps (every exp computed once)
    Frag3 bfr[K1S];
#pragma unroll
    for (int ks = 0; ks < K1S; ++ks) {
      float v[8];
      const float4 o0 = *reinterpret_cast<const float4*>(offs + 16 * ks + 8 * kh);
      const float4 o1 = *reinterpret_cast<const float4*>(offs + 16 * ks + 8 * kh + 4);
      const float o[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float diff = d - o[e];
        v[e] = exp_neg(coeff * (diff * diff));  // schnet.py:206-207 (padded centres meet zero weights)
      }
      bfr[ks] = split8(v);
    }
    // First GEMM one 32-feature block at a time, software-pipelined inside the wave: the bf16 MFMAs of block mb
    // are interleaved with the ssp / split / store (vector work) of block mb-1 - the matrix pipe overlaps with vector
    // instructions of the SAME wave only (split.h).  ssp output: t to HBM (row-major, 16 bytes per store) when
    // training and, split, the B fragments of the second GEMM: registers 0..7 of block mb are k-step 2mb, registers
    // 8..15 k-step 2mb+1, element e = register & 7.
    FF_MARK(1);
    Frag3 tb[K2S];
    float* trow = Tout != nullptr ? Tout + (lbase + row) * F + 4 * kh : nullptr;
    auto finish_block = [&](int mb) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ssp(acc1[mb][8 * half + e]);
        if (trow != nullptr && live) {
          *reinterpret_cast<float4*>(trow + 32 * mb + 16 * half) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(trow + 32 * mb + 16 * half + 8) = make_float4(v[4], v[5], v[6], v[7]);
        }
        tb[2 * mb + half] = split8(v);
      }
    };
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
#pragma unroll
      for (int ks = 0; ks < K1S; ++ks) {
        Frag3 af;
        const u32x4* src = W1f + ((size_t)(mb * K1S + ks) * 3) * 64 + lane;
        af.h = src[0];
        af.m = src[64];
        af.l = src[128];
        mma6(acc1[mb], af, bfr[ks]);
      }
      if (mb > 0) {
        finish_block(mb - 1);
        // interleave: one MFMA, then a slice of the vector work of the previous block
#pragma unroll
        for (int i = 0; i < 6 * K1S; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    FF_MARK(2);
    finish_block(NMB - 1);
    __builtin_amdgcn_sched_barrier(0);
    FF_MARK(3);
    // second GEMM, transposed, two 32-feature output blocks at a time
    constexpr int MP = NMB >= 2 ? 2 : 1;
    float* orow = Wf + (lbase + row) * F + 4 * kh;
#pragma unroll
    for (int mb0 = 0; mb0 < NMB; mb0 += MP) {
      f32x16 acc2[MP];
#pragma unroll
      for (int u = 0; u < MP; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = *reinterpret_cast<const float4*>(b2s + 32 * (mb0 + u) + 8 * q + 4 * kh);
          acc2[u][4 * q] = b.x;
          acc2[u][4 * q + 1] = b.y;
          acc2[u][4 * q + 2] = b.z;
          acc2[u][4 * q + 3] = b.w;
        }
      Frag3 af[MP], an[MP];
#pragma unroll
      for (int u = 0; u < MP; ++u) {
        const u32x4* src = W2f + ((size_t)((mb0 + u) * K2S) * 3) * 64 + lane;
        af[u].h = src[0];
        af[u].m = src[64];
        af[u].l = src[128];
      }
#pragma unroll
      for (int ks = 0; ks < K2S; ++ks) {
        if (ks + 1 < K2S) {  // fragments of the next k-step are requested before this step's MFMAs issue
#pragma unroll
          for (int u = 0; u < MP; ++u) {
            const u32x4* src = W2f + ((size_t)((mb0 + u) * K2S + ks + 1) * 3) * 64 + lane;
            an[u].h = src[0];
            an[u].m = src[64];
            an[u].l = src[128];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < MP; ++u) acc2[u] = mfma_bf16(af[u].l, tb[ks].h, acc2[u]);
#pragma unroll
        for (int u = 0; u < MP; ++u) acc2[u] = mfma_bf16(af[u].h, tb[ks].l, acc2[u]);
#pragma unroll
        for (int u = 0; u < MP; ++u) acc2[u] = mfma_bf16(af[u].m, tb[ks].m, acc2[u]);
#pragma unroll
        for (int u = 0; u < MP; ++u) acc2[u] = mfma_bf16(af[u].m, tb[ks].h, acc2[u]);
#pragma unroll
        for (int u = 0; u < MP; ++u) acc2[u] = mfma_bf16(af[u].h, tb[ks].m, acc2[u]);
#pragma unroll
        for (int u = 0; u < MP; ++u) acc2[u] = mfma_bf16(af[u].h, tb[ks].h, acc2[u]);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < K2S) {
#pragma unroll
          for (int u = 0; u < MP; ++u) af[u] = an[u];
        }
      }
      if (mb0 == 0) FF_MARK(4);
      else FF_MARK(6);
      if (live) {
#pragma unroll
        for (int u = 0; u < MP; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q)  // schnet.py:187: filter times the cosine envelope
            *reinterpret_cast<float4*>(orow + 32 * (mb0 + u) + 8 * q) =
                make_float4(acc2[u][4 * q] * cw, acc2[u][4 * q + 1] * cw, acc2[u][4 * q + 2] * cw,
                            acc2[u][4 * q + 3] * cw);
      }
      if (mb0 == 0) FF_MARK(5);
      else FF_MARK(7);
    }
  }
}



// ---- The same network on TWO fp16 pieces per operand (split.h): 3 MFMAs per product instead of 6.  Scales (powers of
// two, exact): A1 and A2 by their largest magnitude (once per block), the Gaussians by 2^14 (they are <= 1), the hidden
// row of a pair by ITS largest magnitude (per lane pair: the row of pair j lives in lanes j and j + 32).  The biases
// are added to the unscaled fp32 results (one FMA each), so the accumulators start from 0.
#ifndef FFH_THREADS
#define FFH_THREADS 512  // 8 waves per CU; 768 / 1024 were slower inside the step
#endif
template <int NMB, int K1S>
__global__ __launch_bounds__(FFH_THREADS) void k_filter_fwd_h(const float* __restrict__ pair_d,
                                                      const float* __restrict__ pair_c, int P,
                                                      GeosslFilterWeights w, int G,
                                                      const float* __restrict__ offset, float coeff,
                                                      float* __restrict__ Tout, float* __restrict__ Wf,
                                                    const int32_t* __restrict__ dyn_P) {
  constexpr int F = 32 * NMB, K2S = F / 16;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  u32x4* W2f = reinterpret_cast<u32x4*>(smem_raw);          // [NMB][K2S][2][64] A fragments of A2 * s2
  u32x4* W1f = W2f + NMB * K2S * 2 * 64;                    // [NMB][K1S][2][64] A fragments of A1 * s1
  float* b1s = reinterpret_cast<float*>(W1f + NMB * K1S * 2 * 64);  // [F]
  float* b2s = b1s + F;                                     // [F]
  float* offs = b2s + F;                                    // [16*K1S] Gaussian centres, zero padded
  float* red = offs + 16 * K1S;                             // [32] block reduction of the two weight maxima, [32]: row-block counter
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int l = blockIdx.y;
  float s1, s2;
  {
    const float* __restrict__ w2 = w.w2[l];
    const float* __restrict__ w1 = w.w1[l];
    float m1 = 0.0f, m2 = 0.0f;
    for (int i = tid; i < F * G; i += FFH_THREADS) m1 = fmaxf(m1, fabsf(w1[i]));
    for (int i = tid; i < F * F; i += FFH_THREADS) m2 = fmaxf(m2, fabsf(w2[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      m1 = fmaxf(m1, __shfl_xor(m1, o));
      m2 = fmaxf(m2, __shfl_xor(m2, o));
    }
    if (lane == 0) {
      red[wave] = m1;
      red[16 + wave] = m2;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FFH_THREADS / 64; ++i) {
      m1 = fmaxf(m1, red[i]);
      m2 = fmaxf(m2, red[16 + i]);
    }
    int e1, e2;
    s1 = pow2_scale_to_2p14(m1, e1);
    s2 = pow2_scale_to_2p14(m2, e2);
    for (int i = tid; i < NMB * K2S * 64; i += FFH_THREADS) {
      const int ln = i & 63, ks = (i >> 6) % K2S, mb = i / (64 * K2S);
      // contraction-index permutation kperm (split.h): elements 0..3 <- features 4kh.., 4..7 <- features 8+4kh..
      const float* row = w2 + (size_t)(32 * mb + (ln & 31)) * F + 16 * ks + 4 * (ln >> 5);
      const float4 lo = *reinterpret_cast<const float4*>(row), hi = *reinterpret_cast<const float4*>(row + 8);
      const float v[8] = {lo.x * s2, lo.y * s2, lo.z * s2, lo.w * s2, hi.x * s2, hi.y * s2, hi.z * s2, hi.w * s2};
      const Frag2 f = split8h(v);
      u32x4* dst = W2f + ((size_t)(mb * K2S + ks) * 2) * 64 + ln;
      dst[0] = f.h;
      dst[64] = f.l;
    }
    for (int i = tid; i < NMB * K1S * 64; i += FFH_THREADS) {
      const int ln = i & 63, ks = (i >> 6) % K1S, mb = i / (64 * K1S);
      const float* row = w1 + (size_t)(32 * mb + (ln & 31)) * G;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = 16 * ks + 8 * (ln >> 5) + e;
        v[e] = g < G ? row[g] * s1 : 0.0f;
      }
      const Frag2 f = split8h(v);
      u32x4* dst = W1f + ((size_t)(mb * K1S + ks) * 2) * 64 + ln;
      dst[0] = f.h;
      dst[64] = f.l;
    }
    for (int i = tid; i < F; i += FFH_THREADS) {
      b1s[i] = w.b1[l][i];
      b2s[i] = w.b2[l][i];
    }
    for (int i = tid; i < 16 * K1S; i += FFH_THREADS) offs[i] = i < G ? offset[i] : 0.0f;
  }
  int* next_rb = reinterpret_cast<int*>(red + 32);  // the block's row-block counter
  if (tid == 0) *next_rb = 0;
  __syncthreads();
  const float inv1 = 1.0f / (s1 * 16384.0f), inv2 = 1.0f / s2;  // powers of two: exact
  const size_t lbase = (size_t)l * P;  // (the layer stride of T / Wf is the by-value - capacity - count)
  P = dyn_count(P, dyn_P);
  if (P <= 0) return;
  const int nrb = (P + 31) / 32;
  // Row blocks blockIdx.x + gridDim.x t of the layer are handed out on demand: of the two waves of a SIMD the older one
  // wins the issue arbitration (in-kernel marks: 23k against 32k cycles per row block while both run), so equal shares
  // would leave the younger wave to finish alone.  The next block's index and its two scalars per row are fetched one
  // row block ahead.
  auto take = [&]() {
    int t = 0;
    if (lane == 0) t = atomicAdd(next_rb, 1);
    return (int)blockIdx.x + (int)gridDim.x * __builtin_amdgcn_readfirstlane(t);
  };
  int rb = take();
  float d_n = pair_d[min(32 * rb + j, P - 1)], cw_n = pair_c[min(32 * rb + j, P - 1)];
  while (rb < nrb) {
    const int row = 32 * rb + j;
    const bool live = row < P;
    const float d = d_n;
    const float cw = live ? cw_n : 0.0f;
    const int rb_next = take();
    d_n = pair_d[min(32 * rb_next + j, P - 1)];
    cw_n = pair_c[min(32 * rb_next + j, P - 1)];
    // every LDS read of the loop is tile-invariant: an opaque zero in the addresses keeps the compiler from hoisting
    // (and then spilling) fragments, biases and centres across the tile loop
    int z = 0;
    asm volatile("" : "+v"(z));
    const int lz = lane + z, khz = 4 * kh + z;
    // rbf^T B fragments of all k-steps (every exp computed once), scaled by 2^14
    Frag2 bfr[K1S];
#pragma unroll
    for (int ks = 0; ks < K1S; ++ks) {
      float v[8];
      const float4 o0 = *reinterpret_cast<const float4*>(offs + 16 * ks + 2 * khz);
      const float4 o1 = *reinterpret_cast<const float4*>(offs + 16 * ks + 2 * khz + 4);
      const float o[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float diff = d - o[e];
        v[e] = exp_neg(coeff * (diff * diff));  // schnet.py:206-207 (padded centres meet zero weights)
      }
      bfr[ks] = split8h_scaled(v, 16384.0f);
    }
    // first GEMM, transposed: acc1[mb] = (s1 A1)(2^14 rbf^T)[32mb.., rows]
    f32x16 acc1[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[mb][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < K1S; ++ks) {
        const u32x4* src = W1f + ((size_t)(mb * K1S + ks) * 2) * 64 + lz;
        const u32x4 ah = src[0], al = src[64];
        acc1[mb] = mfma_f16(al, bfr[ks].h, acc1[mb]);
        acc1[mb] = mfma_f16(ah, bfr[ks].l, acc1[mb]);
        acc1[mb] = mfma_f16(ah, bfr[ks].h, acc1[mb]);
      }
    }
    // t = ssp(u), u = acc1 * inv1 + b1: to HBM (training), and its largest magnitude in the row
    float* trow = Tout != nullptr ? Tout + (lbase + row) * F + 4 * kh : nullptr;
    float tmax = 0.0f;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(b1s + 32 * mb + 8 * q + khz);
        const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = ssp(fmaf(acc1[mb][4 * q + e], inv1, bb[e]));
          acc1[mb][4 * q + e] = t;
          tmax = fmaxf(tmax, fabsf(t));
        }
        if (trow != nullptr && live)
          *reinterpret_cast<float4*>(trow + 32 * mb + 8 * q) =
              make_float4(acc1[mb][4 * q], acc1[mb][4 * q + 1], acc1[mb][4 * q + 2], acc1[mb][4 * q + 3]);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    int et;
    const float st = pow2_scale_to_2p14(tmax, et);
    const float k2 = inv2 * __builtin_amdgcn_ldexpf(1.0f, et - 14);  // undoes s2 and st
    // B fragments of the second GEMM: registers 0..7 of block mb are k-step 2mb, registers 8..15 k-step 2mb+1
    Frag2 tb[K2S];
#pragma unroll
    for (int ks = 0; ks < K2S; ++ks) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc1[ks >> 1][8 * (ks & 1) + e];
      tb[ks] = split8h_scaled(v, st);
    }
    // second GEMM, transposed, two 32-feature output blocks at a time
    constexpr int MP = NMB >= 2 ? 2 : 1;
    float* orow = Wf + (lbase + row) * F + 4 * kh;
#pragma unroll
    for (int mb0 = 0; mb0 < NMB; mb0 += MP) {
      f32x16 acc2[MP];
#pragma unroll
      for (int u = 0; u < MP; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[u][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < K2S; ++ks) {
        u32x4 ah[MP], al[MP];
#pragma unroll
        for (int u = 0; u < MP; ++u) {
          const u32x4* src = W2f + ((size_t)((mb0 + u) * K2S + ks) * 2) * 64 + lz;
          ah[u] = src[0];
          al[u] = src[64];
        }
#pragma unroll
        for (int u = 0; u < MP; ++u) acc2[u] = mfma_f16(al[u], tb[ks].h, acc2[u]);
#pragma unroll
        for (int u = 0; u < MP; ++u) acc2[u] = mfma_f16(ah[u], tb[ks].l, acc2[u]);
#pragma unroll
        for (int u = 0; u < MP; ++u) acc2[u] = mfma_f16(ah[u], tb[ks].h, acc2[u]);
      }
#pragma unroll
      for (int u = 0; u < MP; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // schnet.py:187: (A2 t + b2) times the cosine envelope
          const float4 b = *reinterpret_cast<const float4*>(b2s + 32 * (mb0 + u) + 8 * q + khz);
          const float4 o = make_float4(fmaf(acc2[u][4 * q], k2, b.x) * cw, fmaf(acc2[u][4 * q + 1], k2, b.y) * cw,
                                       fmaf(acc2[u][4 * q + 2], k2, b.z) * cw, fmaf(acc2[u][4 * q + 3], k2, b.w) * cw);
          if (live) *reinterpret_cast<float4*>(orow + 32 * (mb0 + u) + 8 * q) = o;
        }
    }
    rb = rb_next;
  }
}

#ifdef FF_TIMING
}  // namespace
extern "C" int geossl_filter_fwd_debug_read(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ff_dbg), sizeof(long long) * (2 * 64 * 8 + 2 * 64));
}
namespace {
#endif

}  // namespace

extern "C" int geossl_cfconv_filter_fwd(const float* pair_d, const float* pair_c, int64_t P,
                                        const GeosslFilterWeights* w, int L, int F, int G, const float* offset,
                                        float coeff, float* T, float* Wf, hipStream_t stream) {
  return geossl_cfconv_filter_fwd_dyn(pair_d, pair_c, P, w, L, F, G, offset, coeff, T, Wf, nullptr, stream);
}

extern "C" int geossl_cfconv_filter_fwd_dyn(const float* pair_d, const float* pair_c, int64_t P,
                                            const GeosslFilterWeights* w, int L, int F, int G, const float* offset,
                                            float coeff, float* T, float* Wf, const int32_t* dyn_P,
                                            hipStream_t stream) {
  if (P <= 0 || L <= 0) return 0;
  if (L > GEOSSL_MAX_L || (F != 32 && F != 64 && F != 128) || G > 64 || G < 1) return (int)hipErrorInvalidValue;
  const int nrb = (int)((P + 31) / 32);
  int per_layer = 256 / L;  // one 8-wave block per CU, a layer per block (its weights stay in LDS)
  if (per_layer < 1) per_layer = 1;
  if (per_layer > (nrb + 7) / 8) per_layer = (nrb + 7) / 8;
  dim3 grid(per_layer, L);
#define LAUNCH(NMB, K1S)                                                                                        \
  do {                                                                                                          \
    const size_t lds = (size_t)(NMB * (2 * NMB) + NMB * K1S) * 3 * 1024 + (2 * 32 * NMB + 16 * K1S) * 4;        \
    allow_big_lds(&k_filter_fwd<NMB, K1S>);                                                                     \
    hipLaunchKernelGGL((k_filter_fwd<NMB, K1S>), grid, dim3(512), lds, stream, pair_d, pair_c, (int)P, *w, G,   \
                       offset, coeff, T, Wf, dyn_P);                                                            \
  } while (0)
#define LAUNCH_F(NMB)                   \
  do {                                  \
    if (G <= 16) LAUNCH(NMB, 1);        \
    else if (G <= 32) LAUNCH(NMB, 2);   \
    else if (G <= 48) LAUNCH(NMB, 3);   \
    else LAUNCH(NMB, 4);                \
  } while (0)
  // default: two fp16 pieces per operand (3 MFMAs per product); GEOSSL_FILTER_FWD_BF16X3 selects the three-bf16-piece form
  const bool bf16x3 = getenv("GEOSSL_FILTER_FWD_BF16X3") != nullptr || getenv("GEOSSL_ARITH_24BIT") != nullptr;  // (read per call: bench.py times both forms in one process)
#define LAUNCH_H(NMB, K1S)                                                                                      \
  do {                                                                                                          \
    const size_t lds = (size_t)(NMB * (2 * NMB) + NMB * K1S) * 2 * 1024 + (2 * 32 * NMB + 16 * K1S + 36) * 4;   \
    allow_big_lds(&k_filter_fwd_h<NMB, K1S>);                                                                   \
    hipLaunchKernelGGL((k_filter_fwd_h<NMB, K1S>), grid, dim3(FFH_THREADS), lds, stream, pair_d, pair_c, (int)P, *w, G, \
                       offset, coeff, T, Wf, dyn_P);                                                            \
  } while (0)
#define LAUNCH_HF(NMB)                    \
  do {                                    \
    if (G <= 16) LAUNCH_H(NMB, 1);        \
    else if (G <= 32) LAUNCH_H(NMB, 2);   \
    else if (G <= 48) LAUNCH_H(NMB, 3);   \
    else LAUNCH_H(NMB, 4);                \
  } while (0)
  if (!bf16x3) {
    if (F == 128) LAUNCH_HF(4); else if (F == 64) LAUNCH_HF(2); else LAUNCH_HF(1);
  } else {
    if (F == 128) LAUNCH_F(4); else if (F == 64) LAUNCH_F(2); else LAUNCH_F(1);
  }
#undef LAUNCH_HF
#undef LAUNCH_H
#undef LAUNCH_F
#undef LAUNCH
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
