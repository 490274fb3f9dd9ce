// Chains of atom-row Linear layers in ONE launch.
//
// Between two neighbour aggregations SchNet applies three row-local Linear layers back to back (schnet.py:191 conv.lin2,
// :165-166 act + lin with the residual of :97, then the next block's conv.lin1 at :189; after the last block lin2, lin
// and the head :99-101), and so does the backward pass (dX through conv.lin1 + residual, through lin and act, through
// conv.lin2).  As separate launches each of them is one non-overlapped read / multiply / drain pass over 36 864 rows
// (19-27 us per dispatch, ~13 us of it fixed); here a wave keeps its 32 rows in registers across the whole chain:
//
//   * every product is evaluated transposed on the bf16 matrix pipe (split.h): Y^T = W X^T, weight fragments as the A
//     operand, the wave's rows on the lanes; with the contraction index in `kperm` order the C layout of one product
//     (lane = row, registers = 4 consecutive columns per group) IS the B-operand layout of the next, so a stage's result
//     is activated, stored (16-byte row pieces), split and consumed without leaving the registers;
//   * a stage needs its 96 KB operand image (F = 128) only one 32-column block at a time: the blocks ("chunks", 24 KB)
//     of all stages stream through an LDS ring by LDS-DMA (global_load_lds, no staging registers), two chunks ahead of
//     the one being multiplied; one wait + barrier per chunk, placed after the chunk's MFMAs, where everything
//     outstanding (the DMA of later chunks, the stores of the previous epilogue) is at least one MFMA loop old;
//   * four waves per block, 256 registers per wave (two sets of split fragments are live at a time: chains are cut at
//     three stages, a fourth would spill) and a three-slot ring (72 KB), so two blocks share a CU: one wave's stores,
//     DMA issue and barrier waits are covered by its SIMD neighbour.
//
// Stage s:  Y_s = epi_s( X_s W_s^T + b_s ),  X_{s+1} = Y_s;  epi = [ssp] [* ssp'(tprev)] [+ res], like geossl_linear.
#include "common.h"
#include "geossl_hip.h"
#include "split.h"
#ifdef LOOP_TIMING
// per-phase wall-clock marks (s_memrealtime, 100 MHz) of wave 1 of block 5 of k_layer_loop: (tag, time) pairs, read with
// geossl_loop_debug_read (tools/loop_timing.py; a debug build of this file with -DLOOP_TIMING)
__device__ long long loop_dbg[2 * 1024];
__device__ int loop_dbg_n;
#define LOOP_MARK(tag)                                                                                     \
  do {                                                                                                     \
    if (blockIdx.x == 5 && threadIdx.x == 64) {                                                            \
      const int k_ = loop_dbg_n;                                                                           \
      if (k_ < 1024) {                                                                                     \
        loop_dbg[2 * k_] = (tag);                                                                          \
        loop_dbg[2 * k_ + 1] = (long long)__builtin_amdgcn_s_memrealtime();                                \
        loop_dbg_n = k_ + 1;                                                                               \
      }                                                                                                    \
    }                                                                                                      \
  } while (0)
#else
#define LOOP_MARK(tag) do {} while (0)
#endif
#include "aggregate_reg.h"

#include <cstdlib>

using namespace geossl;

namespace {

// image_z[mb][ks][piece][lane] (u32x4) with the contraction index of every k-step in kperm order:
//   lane (n = lane & 31, kh = lane >> 5), element e  <->  B[k = 16 ks + kperm(e, kh)][n]
template <int KS>
__global__ __launch_bounds__(256) void k_chain_prepare(GeosslPrepareBatch batch, int NO, int transB_call) {
  constexpr int K = 16 * KS;
  const int z = blockIdx.y;
  const int transB = batch.tb[z] == 0 ? transB_call : (batch.tb[z] == 1 ? 1 : 0);
  const float* __restrict__ W = batch.W[z];
  u32x4* __restrict__ image = reinterpret_cast<u32x4*>(batch.image[z]);
  const int ldw = batch.ldw[z] != 0 ? batch.ldw[z] : (transB ? K : NO);  // row stride of W
  const int nitems = (NO / 32) * KS * 64;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float sw = 1.0f;
  if constexpr (KS == 8) {
    // F = 128 (weight-stationary kernel): two fp16 pieces of W * 2^(14 - eW) per 32-column output block (one wave of the
    // chain kernel owns one block), eW = exponent of the block's largest |W| (split.h).  A block of this kernel formats
    // half of one output block (512 items each) and finds that block's largest magnitude itself (4096 values).
    __shared__ float red[4];
    float mw = 0.0f;
    const int mb_blk = (blockIdx.x * 256) / (64 * KS);
    if (transB) {  // rows 32 mb .. 32 mb + 31 of W [NO][K]
      for (int q = threadIdx.x; q < 32 * K / 4; q += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(W + (size_t)(32 * mb_blk + q / (K / 4)) * ldw + 4 * (q % (K / 4)));
        mw = fmaxf(fmaxf(mw, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
      }
    } else {       // columns 32 mb .. 32 mb + 31 of W [K][NO]
      for (int q = threadIdx.x; q < 32 * K; q += 256) mw = fmaxf(mw, fabsf(W[(size_t)(q >> 5) * ldw + 32 * mb_blk + (q & 31)]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mw = fmaxf(mw, __shfl_xor(mw, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mw;
    __syncthreads();
    mw = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int eW;
    sw = pow2_scale_to_2p14(mw, eW);
    if ((i & (64 * KS - 1)) == 0) reinterpret_cast<int*>(image)[(NO / 32) * KS * 2 * 64 * 4 + mb_blk] = eW;
  }
  if (i >= nitems) return;
  const int ln = i & 63, ks = (i >> 6) % KS, mb = i / (64 * KS);
  const int n = 32 * mb + (ln & 31), kh = ln >> 5;
  float v[8];
  if (transB) {  // W [NO][K]: B[k][n] = W[n][k]
    const f32x4 lo = *reinterpret_cast<const f32x4*>(W + (size_t)n * ldw + 16 * ks + 4 * kh);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(W + (size_t)n * ldw + 16 * ks + 8 + 4 * kh);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
    v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
  } else {       // W [K][NO]: B[k][n] = W[k][n]
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = W[(size_t)(16 * ks + kperm(e, kh)) * ldw + n];
  }
  if constexpr (KS == 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= sw;
    const Frag2 f = split8h(v);
    u32x4* dst = image + ((size_t)(mb * KS + ks) * 2) * 64 + ln;
    dst[0] = f.h;
    dst[64] = f.l;
  } else {
    const Frag3 f = split8(v);
    u32x4* dst = image + ((size_t)(mb * KS + ks) * 3) * 64 + ln;
    dst[0] = f.h;
    dst[64] = f.m;
    dst[128] = f.l;
  }
}

#ifdef CHAIN_TIMING
__device__ long long chain_dbg[8 * 64];
#define CHAIN_MARK(slot)                                                                     \
  do {                                                                                       \
    if (blockIdx.x == 7 && threadIdx.x == 64 && g == blockIdx.x) chain_dbg[dbg_n++ * 8 + (slot)] = clock64(); \
  } while (0)
#else
#define CHAIN_MARK(slot) do {} while (0)
#endif
#ifndef CHAIN_SLOTS_N
#define CHAIN_SLOTS_N 3
#endif
#ifndef CHAIN_WPS
#define CHAIN_WPS 2
#endif
constexpr int CHAIN_SLOTS = CHAIN_SLOTS_N;
// ring depth: the chunk after the one in work must have been requested at least one barrier earlier (>= 2 slots)
constexpr int chain_slots(int nch) { return nch < 2 ? 2 : (nch < CHAIN_SLOTS ? nch : CHAIN_SLOTS); }

template <int KS, int NS>
__global__ __launch_bounds__(256, CHAIN_WPS) void k_row_chain(GeosslChain ch, const float* __restrict__ X, int ldx, int R) {
  constexpr int F = 16 * KS, NMB = KS / 2, NCH = NS * NMB;
  constexpr int CHUNK = KS * 3 * 64;                          // u32x4 per chunk (one 32-column block of one stage)
  constexpr int NSLOT = chain_slots(NCH);
  constexpr int PPW = (KS * 3) / 4;                            // 1 KB pieces of a chunk per wave (KS * 3 pieces, 4 waves)
  static_assert((KS * 3) % 4 == 0 || KS == 2, "pieces split evenly over the four waves");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  u32x4* ring = reinterpret_cast<u32x4*>(smem_raw);                          // [NSLOT][CHUNK]
  float* bias_s = reinterpret_cast<float*>(ring + (size_t)NSLOT * CHUNK);    // [NS][F]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nrb = (R + 31) / 32, ngroups = (nrb + 3) / 4;
  int my_groups = 0;
  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) ++my_groups;
  const int total = my_groups * NCH;                          // chunks this block streams (the chain once per group)
  // LDS-DMA of stream position cc (chunk cc % NCH of the chain) into slot cc % NSLOT: this wave's share of the pieces
  auto issue = [&](int cc) {
    const int c = cc % NCH, s = c / NMB, mb = c - s * NMB;
    const u32x4* src = reinterpret_cast<const u32x4*>(ch.st[s].image) + (size_t)mb * CHUNK;
    u32x4* dst = ring + (size_t)(cc % NSLOT) * CHUNK;
    if constexpr (KS == 2) {  // 6 pieces: waves 0, 1 take two, waves 2, 3 one
      for (int p = wave; p < KS * 3; p += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(dst + p * 64), 16, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < PPW; ++u) {
        const int p = wave * PPW + u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(dst + p * 64), 16, 0, 0);
      }
    }
  };
  for (int cc = 0; cc < NSLOT && cc < total; ++cc) issue(cc);
  for (int i = tid; i < NS * F; i += 256) {
    const int s = i / F;
    bias_s[i] = ch.st[s].bias != nullptr ? ch.st[s].bias[i - s * F] : 0.0f;
  }
  int cc = 0;  // stream position of the chunk in work
#ifdef CHAIN_TIMING
  int dbg_n = 0;
#endif
  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int rb = 4 * g + wave;
    const int row = 32 * rb + j;
    const bool live = row < R;
    const size_t rowc = (size_t)min(row, R - 1);
    CHAIN_MARK(0);
    // this wave's 32 rows of X as B fragments (kperm order: columns 16ks + 4kh + {0..3} and 16ks + 8 + 4kh + {0..3})
    Frag3 xf[KS], yf[KS];
    {
      const float* xr = X + rowc * ldx + 4 * kh;
      f32x4 raw[KS][2];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        raw[ks][0] = *reinterpret_cast<const f32x4*>(xr + 16 * ks);
        raw[ks][1] = *reinterpret_cast<const f32x4*>(xr + 16 * ks + 8);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float v[8] = {raw[ks][0].x, raw[ks][0].y, raw[ks][0].z, raw[ks][0].w,
                            raw[ks][1].x, raw[ks][1].y, raw[ks][1].z, raw[ks][1].w};
        xf[ks] = split8(v);
      }
    }
    if (cc == 0) __syncthreads();  // first group: chunks 0 .. NSLOT-1 landed (the barrier waits for the DMA), biases staged
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const GeosslChainStage st = ch.st[s];
      const int flags = st.flags;
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb, ++cc) {
        CHAIN_MARK(1);
        // epilogue operands of this chunk (row pieces in C layout), requested ahead of the MFMAs
        f32x4 tp[4], rs[4];
        const size_t eo = rowc * st.ld + 32 * mb + 4 * kh;
        if (st.tprev != nullptr) {
#pragma unroll
          for (int q = 0; q < 4; ++q) tp[q] = *reinterpret_cast<const f32x4*>(st.tprev + eo + 8 * q);
        }
        if (st.res != nullptr) {
#pragma unroll
          for (int q = 0; q < 4; ++q) rs[q] = *reinterpret_cast<const f32x4*>(st.res + eo + 8 * q);
        }
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + s * F + 32 * mb + 8 * q + 4 * kh);
          acc[4 * q] = b.x;
          acc[4 * q + 1] = b.y;
          acc[4 * q + 2] = b.z;
          acc[4 * q + 3] = b.w;
        }
        {
          const u32x4* Ws = ring + (size_t)(cc % NSLOT) * CHUNK + lane;
          Frag3 af, an;
          af.h = Ws[0]; af.m = Ws[64]; af.l = Ws[128];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
              const u32x4* src = Ws + (size_t)((ks + 1) * 3) * 64;
              an.h = src[0]; an.m = src[64]; an.l = src[128];
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef CHAIN_ABLATE_MFMA
            acc[0] += __uint_as_float(af.h[0] ^ xf[ks].h[0]) + __uint_as_float(af.l[3] ^ xf[ks].m[1]);
#else
            mma6(acc, af, xf[ks]);
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KS) af = an;
          }
        }
        CHAIN_MARK(2);
        // every wave is done with this slot; the DMA issued so far has landed (the barrier drains it): refill the slot
        __syncthreads();
        CHAIN_MARK(3);
#ifndef CHAIN_ABLATE_DMA
        if (cc + NSLOT < total) issue(cc + NSLOT);
#endif
        CHAIN_MARK(5);
        // epilogue in registers: lane = row, register 4q + e = column 32mb + 8q + 4kh + e
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r];
        if (flags & GEOSSL_EPI_SSP) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = ssp(v[r]);
        }
        if (st.tprev != nullptr) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[4 * q] *= dssp_from_out(tp[q].x);
            v[4 * q + 1] *= dssp_from_out(tp[q].y);
            v[4 * q + 2] *= dssp_from_out(tp[q].z);
            v[4 * q + 3] *= dssp_from_out(tp[q].w);
          }
        }
        if (st.res != nullptr) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[4 * q] += rs[q].x;
            v[4 * q + 1] += rs[q].y;
            v[4 * q + 2] += rs[q].z;
            v[4 * q + 3] += rs[q].w;
          }
        }
        CHAIN_MARK(6);
#ifdef CHAIN_ABLATE_STORE
        if (st.out != nullptr && live && flags == 12345) {
#else
        if (st.out != nullptr && live) {
#endif
          float* o = st.out + (size_t)row * st.ld + 32 * mb + 4 * kh;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(o + 8 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        }
        CHAIN_MARK(4);
        if (s + 1 < NS) {  // registers 0..7 / 8..15 are k-steps 2mb / 2mb+1 of the next stage (kperm)
          const float lo[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
          const float hi[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
          yf[2 * mb] = split8(lo);
          yf[2 * mb + 1] = split8(hi);
        }
      }
      if (s + 1 < NS) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = yf[ks];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Eight-wave form with a balanced split.  At the bench size the chain has 1152 row blocks for 1024 SIMDs: with one wave
// per row block every ninth SIMD carries two (the launch lasts two row-block times while 7/8 of the chip waits).  Here a
// block has four REGULAR waves (one row block each, as above) and four TEAM waves that share a fifth row block: team
// wave m computes output column block m of every stage - a quarter of the row block's work per SIMD - and the four
// exchange their results between stages through LDS (split fragments, double buffered).  Per round a block takes 4 + 1
// row blocks: 256 blocks x 5 cover 1280, so the 128 row blocks beyond 1024 cost a quarter row-block time on 128 CUs.
// The team waves read their weight fragments straight from the operand image in L2 (all k-steps of their chunk
// requested at once), they take part in every chunk barrier and in the LDS-DMA of the ring.
template <int KS, int NS>
__global__ __launch_bounds__(512, 2) void k_row_chain8(GeosslChain ch, const float* __restrict__ X, int ldx, int R) {
  constexpr int F = 16 * KS, NMB = KS / 2, NCH = NS * NMB;
  constexpr int CHUNK = KS * 3 * 64;
  constexpr int NSLOT = chain_slots(NCH);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  u32x4* ring = reinterpret_cast<u32x4*>(smem_raw);                          // [NSLOT][CHUNK]
  u32x4* xchg = ring + (size_t)NSLOT * CHUNK;                                // [2][KS][3][64] team exchange
  float* bias_s = reinterpret_cast<float*>(xchg + (size_t)2 * CHUNK);        // [NS][F]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nrb = (R + 31) / 32, B = gridDim.x;
  const int rounds = (nrb + 5 * B - 1) / (5 * B);
  const int total = rounds * NCH;
  auto issue = [&](int cc) {
    const int c = cc % NCH, s = c / NMB, mb = c - s * NMB;
    const u32x4* src = reinterpret_cast<const u32x4*>(ch.st[s].image) + (size_t)mb * CHUNK;
    u32x4* dst = ring + (size_t)(cc % NSLOT) * CHUNK;
    for (int p = wave; p < KS * 3; p += 8)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * 64 + lane),
                                       (__attribute__((address_space(3))) void*)(dst + p * 64), 16, 0, 0);
  };
  for (int cc = 0; cc < NSLOT && cc < total; ++cc) issue(cc);
  for (int i = tid; i < NS * F; i += 512) {
    const int s = i / F;
    bias_s[i] = ch.st[s].bias != nullptr ? ch.st[s].bias[i - s * F] : 0.0f;
  }
  // Order of one chunk iteration: MFMAs, epilogue in registers (this is where the chunk's tprev / res loads are
  // waited for: the vector-memory counter is in order, so everything issued before them - the previous chunk's stores,
  // the last LDS-DMA - is one MFMA loop old by then), barrier, then this chunk's stores and the next LDS-DMA.  Nothing
  // young is outstanding at the barrier's vmcnt(0).
  // the epilogue of one chunk: activation and epilogue operands; leaves the values in v
  auto epilogue = [&](const GeosslChainStage& st, int mb, const f32x16& acc, const f32x4 (&tp)[4], const f32x4 (&rs)[4],
                      float (&v)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    if (st.flags & GEOSSL_EPI_SSP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = ssp(v[r]);
    }
    if (st.tprev != nullptr) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[4 * q] *= dssp_from_out(tp[q].x);
        v[4 * q + 1] *= dssp_from_out(tp[q].y);
        v[4 * q + 2] *= dssp_from_out(tp[q].z);
        v[4 * q + 3] *= dssp_from_out(tp[q].w);
      }
    }
    if (st.res != nullptr) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[4 * q] += rs[q].x;
        v[4 * q + 1] += rs[q].y;
        v[4 * q + 2] += rs[q].z;
        v[4 * q + 3] += rs[q].w;
      }
    }
  };
  auto store_v = [&](const GeosslChainStage& st, int mb, uint32_t ro, bool live, const float (&v)[16])
                     __attribute__((always_inline)) {
    if (st.out != nullptr && live) {
      char* o = reinterpret_cast<char*>(st.out) + 128 * mb;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(o + 32 * q + ro) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    }
  };
  auto load_x = [&](uint32_t rowc, Frag3 (&xf)[KS]) __attribute__((always_inline)) {
    const char* xb = reinterpret_cast<const char*>(X);
    const uint32_t xo = rowc * (uint32_t)ldx * 4u + 16u * kh;
    f32x4 raw[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      raw[ks][0] = *reinterpret_cast<const f32x4*>(xb + 64 * ks + xo);
      raw[ks][1] = *reinterpret_cast<const f32x4*>(xb + 64 * ks + 32 + xo);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float v[8] = {raw[ks][0].x, raw[ks][0].y, raw[ks][0].z, raw[ks][0].w,
                          raw[ks][1].x, raw[ks][1].y, raw[ks][1].z, raw[ks][1].w};
      xf[ks] = split8(v);
    }
  };
  int cc = 0;
  for (int rd = 0; rd < rounds; ++rd) {
    const int base = rd * 5 * B;
    if (wave < 4) {
      // ---- regular wave: one row block through the whole chain, weights from the ring
      const int rb = base + 4 * (int)blockIdx.x + wave;
      const int row = 32 * rb + j;
      const bool live = row < R;
      const uint32_t rowc = (uint32_t)min(row, R - 1);
      Frag3 xf[KS], yf[KS];
      load_x(rowc, xf);
      __syncthreads();  // round start: (first round) chunks 0 .. NSLOT-1 landed, biases staged; the team's X is published
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const GeosslChainStage st = ch.st[s];
        const uint32_t ro = rowc * (uint32_t)st.ld * 4u + 16u * kh;  // byte offset of this lane's row pieces (< 4 GB)
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb, ++cc) {
          f32x4 tp[4], rs[4];
          if (st.tprev != nullptr) {
            const char* tb = reinterpret_cast<const char*>(st.tprev) + 128 * mb;
#pragma unroll
            for (int q = 0; q < 4; ++q) tp[q] = *reinterpret_cast<const f32x4*>(tb + 32 * q + ro);
          }
          if (st.res != nullptr) {
            const char* rbp = reinterpret_cast<const char*>(st.res) + 128 * mb;
#pragma unroll
            for (int q = 0; q < 4; ++q) rs[q] = *reinterpret_cast<const f32x4*>(rbp + 32 * q + ro);
          }
          f32x16 acc;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + s * F + 32 * mb + 8 * q + 4 * kh);
            acc[4 * q] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
          }
          {
            const u32x4* Ws = ring + (size_t)(cc % NSLOT) * CHUNK + lane;
            Frag3 af, an;
            af.h = Ws[0]; af.m = Ws[64]; af.l = Ws[128];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              if (ks + 1 < KS) {
                const u32x4* src = Ws + (size_t)((ks + 1) * 3) * 64;
                an.h = src[0]; an.m = src[64]; an.l = src[128];
              }
              __builtin_amdgcn_sched_barrier(0);
              mma6(acc, af, xf[ks]);
              __builtin_amdgcn_sched_barrier(0);
              if (ks + 1 < KS) af = an;
            }
          }
          float v[16];
          epilogue(st, mb, acc, tp, rs, v);
          if (s + 1 < NS) {  // registers 0..7 / 8..15 are k-steps 2mb / 2mb+1 of the next stage (kperm)
            const float lo[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
            const float hi[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
            yf[2 * mb] = split8(lo);
            yf[2 * mb + 1] = split8(hi);
          }
          __syncthreads();  // every wave is done with this slot; the DMA issued so far has landed: refill the slot
          store_v(st, mb, ro, live, v);
          if (cc + NSLOT < total) issue(cc + NSLOT);
        }
        if (s + 1 < NS) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) xf[ks] = yf[ks];
        }
      }
    } else {
      // ---- team wave m: column block m of every stage of the block's shared row block
      const int m = wave - 4;
      const int rb = base + 4 * B + (int)blockIdx.x;
      const bool active = m < NMB && rb < nrb;   // wave-uniform
      const int row = 32 * rb + j;
      const bool live = active && row < R;
      const uint32_t rowc = (uint32_t)min(row, R - 1);
      if (active) {  // this wave's quarter of the row block's X (k-steps 2m, 2m+1), split, into exchange buffer 1
        const char* xb = reinterpret_cast<const char*>(X) + 128 * m;
        const uint32_t xo = rowc * (uint32_t)ldx * 4u + 16u * kh;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(xb + xo), r1 = *reinterpret_cast<const f32x4*>(xb + 32 + xo);
        const f32x4 r2 = *reinterpret_cast<const f32x4*>(xb + 64 + xo), r3 = *reinterpret_cast<const f32x4*>(xb + 96 + xo);
        const float lo[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        const float hi[8] = {r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
        const Frag3 f0 = split8(lo), f1 = split8(hi);
        u32x4* xd = xchg + (size_t)CHUNK + (size_t)(2 * m * 3) * 64 + lane;
        xd[0] = f0.h; xd[64] = f0.m; xd[128] = f0.l;
        xd[192] = f1.h; xd[256] = f1.m; xd[320] = f1.l;
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const GeosslChainStage st = ch.st[s];
#pragma unroll
        for (int mbc = 0; mbc < NMB; ++mbc, ++cc) {
          float v[16];
          uint32_t ro = 0;
          if (active && mbc == 0) {
            // all weight fragments of chunk (s, m) from the image (L2), requested before anything else
            // (uniform base + 32-bit lane offset: scalar-base addressing, no per-load 64-bit address registers)
            const char* img = reinterpret_cast<const char*>(st.image) + (size_t)(m * KS) * 3 * 1024;
            const uint32_t voff = (uint32_t)lane * 16u;
            Frag3 af[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              const char* bk = img + ks * 3 * 1024;
              asm volatile("" : "+s"(bk));  // keep the base scalar: loads become saddr + lane offset + immediate
              af[ks].h = *reinterpret_cast<const u32x4*>(bk + voff);
              af[ks].m = *reinterpret_cast<const u32x4*>(bk + 1024 + voff);
              af[ks].l = *reinterpret_cast<const u32x4*>(bk + 2048 + voff);
            }
            f32x4 tp[4], rs[4];
            ro = rowc * (uint32_t)st.ld * 4u + 16u * kh;
            if (st.tprev != nullptr) {
              const char* tb = reinterpret_cast<const char*>(st.tprev) + 128 * m;
#pragma unroll
              for (int q = 0; q < 4; ++q) tp[q] = *reinterpret_cast<const f32x4*>(tb + 32 * q + ro);
            }
            if (st.res != nullptr) {
              const char* rbp = reinterpret_cast<const char*>(st.res) + 128 * m;
#pragma unroll
              for (int q = 0; q < 4; ++q) rs[q] = *reinterpret_cast<const f32x4*>(rbp + 32 * q + ro);
            }
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + s * F + 32 * m + 8 * q + 4 * kh);
              acc[4 * q] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
            }
            {  // the row block's input of this stage: what the four team waves published one stage ago (or X)
              const u32x4* xs = xchg + (size_t)((s + 1) & 1) * CHUNK + lane;
              Frag3 xa, xn;
              xa.h = xs[0]; xa.m = xs[64]; xa.l = xs[128];
#pragma unroll
              for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
                  const u32x4* src = xs + (size_t)((ks + 1) * 3) * 64;
                  xn.h = src[0]; xn.m = src[64]; xn.l = src[128];
                }
                __builtin_amdgcn_sched_barrier(0);
                mma6(acc, af[ks], xa);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < KS) xa = xn;
              }
            }
            epilogue(st, m, acc, tp, rs, v);
            if (s + 1 < NS) {  // publish this wave's two k-steps of the next stage's input
              const float lo[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
              const float hi[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
              const Frag3 f0 = split8(lo), f1 = split8(hi);
              u32x4* xd = xchg + (size_t)(s & 1) * CHUNK + (size_t)(2 * m * 3) * 64 + lane;
              xd[0] = f0.h; xd[64] = f0.m; xd[128] = f0.l;
              xd[192] = f1.h; xd[256] = f1.m; xd[320] = f1.l;
            }
          }
          __syncthreads();
          if (active && mbc == 0) store_v(st, m, ro, live, v);
          if (cc + NSLOT < total) issue(cc + NSLOT);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-stationary form (F = 128): a block of four waves, one per SIMD, works on up to RB row blocks.  Wave m owns
// output column block m of every stage: its 24 weight fragments of a stage sit in registers for all the block's row
// blocks (no LDS ring, no LDS-DMA, no per-chunk barrier), the row blocks' inputs sit in LDS as split fragments (24 KB
// per row block) and every wave walks them; a stage's results stay in registers (16 per row block) until all waves
// are done reading the stage's input, then every wave writes its two k-steps of the next stage's input in place:
// two LDS-only barriers per stage.  The stores of a row block are issued one row block late: every vector-memory
// request of a wave completes in order, so the wait for the next epilogue operands also covers whatever was issued
// before them.  WPS = 2: two such blocks per CU (RB = 3: 72 KB of LDS, 256 registers) that run out of step - the
// memory phases of one fall into the arithmetic of the other; WPS = 1: one block per CU with RB = 5 and 512 registers
// (epilogue operands requested one row block ahead).
// SILU: the instantiation that knows GEOSSL_EPI_SILU / GEOSSL_EPI_MUL_DSILU and the second output (PaiNN's Dense layers);
// the other one carries none of it.
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }                    // k_silu_fwd
__device__ __forceinline__ float dsilu_f(float x) {                                                    // k_silu_bwd
  const float sg = 1.0f / (1.0f + expf(-x));
  return sg * (1.0f + x * (1.0f - sg));
}
// The body works on rows row0 + 32 i + j, i < nloc <= RB, that lie below row_end (k_row_chain_cu: a run of whole row
// blocks of the launch; k_layer_loop: the atoms of a block's molecules); R = rows of the tensors (buffer ranges).
template <int NS, int RB, int WPS, bool SILU, typename ChainT>
__device__ __forceinline__ void chain_cu_body(const ChainT& ch, const float* __restrict__ X, int ldx, int R,
                                              int row0, int row_end, int nloc) {
  constexpr int KS = 8, F = 128;
  constexpr int RBF = KS * 2 * 64;      // u32x4 per row block of fragments (two fp16 pieces, split.h)
#ifndef CHAIN_CU_NEB2
#define CHAIN_CU_NEB2 1
#endif
  // buffers of epilogue operands (2: requested one row block ahead; measured equal with two blocks per CU, where the
  // 32 registers are better left to the allocator)
  constexpr int NEB = WPS == 1 ? 2 : CHAIN_CU_NEB2;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  u32x4* xbuf = reinterpret_cast<u32x4*>(smem_raw);                       // [RB][KS][2][64]
  float* bias_s = reinterpret_cast<float*>(xbuf + (size_t)RB * RBF);      // [NS][F]
  float* rmax = bias_s + NS * F;                                          // [RB][4][32] largest |value| of a row in a wave's columns
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int m = __builtin_amdgcn_readfirstlane(tid >> 6);                  // this wave's column block
  const uint32_t voff = (uint32_t)lane * 16u;
  Frag2 af[KS];
  int eW = 0;            // exponent of the stage's weight scale (stored behind the image by k_chain_prepare)
  float krow[RB];        // 2^(e_row - 14) of the rows whose fragments sit in xbuf: undoes their scale
  auto request_weights = [&](const auto& st) __attribute__((always_inline)) {
    const char* img = reinterpret_cast<const char*>(st.image) + (size_t)(m * KS) * 2 * 1024;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const char* bk = img + ks * 2 * 1024;
      asm volatile("" : "+s"(bk));  // keep the base scalar: loads become saddr + lane offset + immediate
      af[ks].h = *reinterpret_cast<const u32x4*>(bk + voff);
      af[ks].l = *reinterpret_cast<const u32x4*>(bk + 1024 + voff);
    }
    eW = reinterpret_cast<const int*>(st.image)[4 * KS * 2 * 64 * 4 + m];  // this wave's column block
  };
  // row maxima of this wave's 16 values per lane -> LDS (lane pair j / j + 32 holds the wave's 32 columns of row j)
  auto publish_max = [&](int i, float mx) __attribute__((always_inline)) {
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (kh == 0) rmax[(i * 4 + m) * 32 + j] = mx;
  };
  // scale of row block i's rows from the four waves' maxima: returns 2^(14 - e), keeps 2^(e - 14)
  auto row_scale = [&](int i) __attribute__((always_inline)) {
    const float* r = rmax + (i * 4) * 32 + j;
    const float mx = fmaxf(fmaxf(r[0], r[32]), fmaxf(r[64], r[96]));
    const int e = mag_exponent(mx);
    krow[i] = __builtin_amdgcn_ldexpf(1.0f, e - 14);
    return __builtin_amdgcn_ldexpf(1.0f, 14 - e);
  };
  request_weights(ch.st[0]);
  for (int i = tid; i < NS * F; i += 256) {
    const int s = i / F;
    bias_s[i] = ch.st[s].bias != nullptr ? ch.st[s].bias[i - s * F] : 0.0f;
  }
  // ---- input rows: this wave's 32 columns (k-steps 2m, 2m+1) of every row block, split, into the fragment buffer
  auto load_input = [&](const float* __restrict__ Xp, int ldxp) __attribute__((always_inline)) {
    f32x4 raw[RB][4];
    const char* xb = reinterpret_cast<const char*>(Xp) + 128 * m;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const uint32_t rowc = (uint32_t)min(row0 + 32 * min(i, nloc - 1) + j, row_end - 1);
      const uint32_t xo = rowc * (uint32_t)ldxp * 4u + 16u * kh;
#pragma unroll
      for (int q = 0; q < 4; ++q) raw[i][q] = *reinterpret_cast<const f32x4*>(xb + 32 * q + xo);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      if (i < nloc) {
        float mx = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          mx = fmaxf(fmaxf(mx, fmaxf(fabsf(raw[i][q].x), fabsf(raw[i][q].y))), fmaxf(fabsf(raw[i][q].z), fabsf(raw[i][q].w)));
        publish_max(i, mx);
      }
    }
    lds_barrier();  // the row maxima of all four waves
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      if (i < nloc) {
        const float sr = row_scale(i);
        const float lo[8] = {raw[i][0].x, raw[i][0].y, raw[i][0].z, raw[i][0].w,
                             raw[i][1].x, raw[i][1].y, raw[i][1].z, raw[i][1].w};
        const float hi[8] = {raw[i][2].x, raw[i][2].y, raw[i][2].z, raw[i][2].w,
                             raw[i][3].x, raw[i][3].y, raw[i][3].z, raw[i][3].w};
        const Frag2 f0 = split8h_scaled(lo, sr), f1 = split8h_scaled(hi, sr);
        u32x4* xd = xbuf + (size_t)i * RBF + (size_t)(2 * m * 2) * 64 + lane;
        xd[0] = f0.h; xd[64] = f0.l;
        xd[128] = f1.h; xd[192] = f1.l;
      }
    }
  };
  load_input(X, ldx);
  lds_barrier();  // biases staged, fragments published
  LOOP_MARK(100);
  float vout[RB][16];  // a stage's results (this wave's column block of every row block)
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const auto st = ch.st[s];
    const int eWs = eW;
    f32x4 tp[NEB][4], rs[NEB][4];
    // Every vector-memory instruction of the stage loop is issued unconditionally (buffer addressing: a null operand
    // is a zero-sized buffer whose loads return 0 and whose stores are dropped, rows past R get an out-of-range
    // offset), so the number of requests between a load and its use is a constant and the wait for row block i's
    // operands leaves the younger requests - the next row block's operands, the last stores - in flight.
    const uint32_t nbytes = (uint32_t)R * (uint32_t)st.ld * 4u;
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st.tprev), 0, st.tprev != nullptr ? nbytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st.res), 0, st.res != nullptr ? nbytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(st.out, 0, st.out != nullptr ? nbytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(st.out_act, 0, (SILU && st.out_act != nullptr) ? nbytes : 0u, 0x00020000);
    const bool act_silu = SILU && (st.flags & GEOSSL_EPI_SILU) != 0;
    auto row_off = [&](int i) __attribute__((always_inline)) {
      const int row = row0 + 32 * i + j;
      return row < row_end ? (uint32_t)row * (uint32_t)st.ld * 4u + 128u * m + 16u * kh : 0xFFFFFF00u;  // out of range (also + 96): dropped
    };
    auto request_epi = [&](int i, f32x4 (&t)[4], f32x4 (&r)[4]) __attribute__((always_inline)) {
      const uint32_t ro = row_off(i);
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_t, ro + 32 * q, 0, 0));
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, ro + 32 * q, 0, 0));
    };
    auto store_rb = [&](int i) __attribute__((always_inline)) {
      const uint32_t ro = row_off(i);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_raw_buffer_store_b128(
            __builtin_bit_cast(u32x4, f32x4{vout[i][4 * q], vout[i][4 * q + 1], vout[i][4 * q + 2], vout[i][4 * q + 3]}), rs_o,
            ro + 32 * q, 0, 0);
      if constexpr (SILU) {  // the activated copy (issued unconditionally like every request of the loop: a null
                             // destination is a zero-sized buffer)
        float av[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) av[e] = act_silu ? silu_f(vout[i][e]) : vout[i][e];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          __builtin_amdgcn_raw_buffer_store_b128(
              __builtin_bit_cast(u32x4, f32x4{av[4 * q], av[4 * q + 1], av[4 * q + 2], av[4 * q + 3]}), rs_a, ro + 32 * q, 0, 0);
      }
    };
    if (NEB == 2) request_epi(0, tp[0], rs[0]);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      if (i < nloc) {
        if (NEB == 2) {
          if (i + 1 < nloc) request_epi(i + 1, tp[(i + 1) & 1], rs[(i + 1) & 1]);
        } else {
          request_epi(i, tp[0], rs[0]);  // ahead of this row block's MFMAs; the other block of the CU covers the rest
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        {
          const u32x4* xs = xbuf + (size_t)i * RBF + lane;
          Frag2 xa, xn;
          xa.h = xs[0]; xa.l = xs[64];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
              const u32x4* src = xs + (size_t)((ks + 1) * 2) * 64;
              xn.h = src[0]; xn.l = src[64];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma_f16(af[ks].l, xa.h, acc);
            acc = mfma_f16(af[ks].h, xa.l, acc);
            acc = mfma_f16(af[ks].h, xa.h, acc);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KS) xa = xn;
          }
        }
        {  // undo the two operand scales (powers of two) and add the bias
          const float kk = krow[i] * __builtin_amdgcn_ldexpf(1.0f, eWs - 14);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_s + s * F + 32 * m + 8 * q + 4 * kh);
            acc[4 * q] = fmaf(acc[4 * q], kk, bq.x);
            acc[4 * q + 1] = fmaf(acc[4 * q + 1], kk, bq.y);
            acc[4 * q + 2] = fmaf(acc[4 * q + 2], kk, bq.z);
            acc[4 * q + 3] = fmaf(acc[4 * q + 3], kk, bq.w);
          }
        }
        // epilogue in registers: lane = row, register 4q + e = column 32m + 8q + 4kh + e
        const f32x4(&t)[4] = tp[NEB == 2 ? (i & 1) : 0];
        const f32x4(&r)[4] = rs[NEB == 2 ? (i & 1) : 0];
        if (s > 0 && (st.flags & GEOSSL_CHAIN_ADD_PREV)) {  // + the result of the stage before (another pass over a wide input)
#pragma unroll
          for (int e = 0; e < 16; ++e) vout[i][e] += acc[e];
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) vout[i][e] = acc[e];
        }
        if (st.flags & GEOSSL_EPI_SSP) {
#pragma unroll
          for (int e = 0; e < 16; ++e) vout[i][e] = ssp(vout[i][e]);
        }
        if (SILU && st.tprev != nullptr && (st.flags & GEOSSL_EPI_MUL_DSILU)) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            vout[i][4 * q] *= dsilu_f(t[q].x);
            vout[i][4 * q + 1] *= dsilu_f(t[q].y);
            vout[i][4 * q + 2] *= dsilu_f(t[q].z);
            vout[i][4 * q + 3] *= dsilu_f(t[q].w);
          }
        } else if (st.tprev != nullptr) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            vout[i][4 * q] *= dssp_from_out(t[q].x);
            vout[i][4 * q + 1] *= dssp_from_out(t[q].y);
            vout[i][4 * q + 2] *= dssp_from_out(t[q].z);
            vout[i][4 * q + 3] *= dssp_from_out(t[q].w);
          }
        }
        if (st.res != nullptr) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            vout[i][4 * q] += r[q].x;
            vout[i][4 * q + 1] += r[q].y;
            vout[i][4 * q + 2] += r[q].z;
            vout[i][4 * q + 3] += r[q].w;
          }
        }
        if (s + 1 < NS) {  // largest magnitudes of the results: scale of the next stage's input rows
          float mx = 0.0f;
#pragma unroll
          for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fabsf(vout[i][e]));
          publish_max(i, mx);
        }
        if (i > 0) store_rb(i - 1);      // one row block late (see the header)
        if (i + 1 == nloc) store_rb(i);  // (register arrays are only ever indexed by unrolled constants)
      }
    }
    LOOP_MARK(110 + s);
    if (s + 1 < NS) {
      request_weights(ch.st[s + 1]);  // the fragments of this stage are dead; in flight across the exchange
      if (ch.st[s + 1].flags & GEOSSL_CHAIN_SAME_INPUT) continue;  // the next stage reads the same input fragments
      lds_barrier();                  // every wave is done reading this stage's input
      if (ch.st[s + 1].flags & GEOSSL_CHAIN_NEW_INPUT) {  // the next stage brings its own input rows
        load_input(ch.st[s + 1].xin, ch.st[s + 1].ldxin);
        lds_barrier();
        continue;
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        if (i < nloc) {  // registers 0..7 / 8..15 are k-steps 2m / 2m+1 of the next stage (kperm)
          const float sr = row_scale(i);
          float lo[8], hi[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            lo[e] = vout[i][e];
            hi[e] = vout[i][8 + e];
          }
          if constexpr (SILU) {
            if (st.flags & GEOSSL_EPI_SILU) {  // the next stage reads silu(Y); |silu(y)| <= |y|: the row scale holds
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                lo[e] = silu_f(lo[e]);
                hi[e] = silu_f(hi[e]);
              }
            }
          }
          const Frag2 f0 = split8h_scaled(lo, sr), f1 = split8h_scaled(hi, sr);
          u32x4* xd = xbuf + (size_t)i * RBF + (size_t)(2 * m * 2) * 64 + lane;
          xd[0] = f0.h; xd[64] = f0.l;
          xd[128] = f1.h; xd[192] = f1.l;
        }
      }
      lds_barrier();
    }
  }
}

template <int NS, int RB, int WPS, bool SILU>
__global__ __launch_bounds__(256, WPS) void k_row_chain_cu(GeosslChain ch, const float* __restrict__ X, int ldx, int R,
                                                           const int32_t* __restrict__ dyn_R) {
  R = dyn_count(R, dyn_R);  // (a grid sized for more rows only makes the runs of row blocks shorter: nloc <= RB holds)
  const int nrb = (R + 31) / 32;
  // contiguous runs of row blocks: the first `extra` blocks take one more (they are the first to be placed on a CU, so
  // a CU that holds two blocks holds at most one long one)
  const int nblk = (int)gridDim.x, base = nrb / nblk, extra = nrb - base * nblk, b = (int)blockIdx.x;
  const int rb0 = b * base + min(b, extra);
  const int nloc = base + (b < extra ? 1 : 0);                             // <= RB, uniform
  if (nloc == 0) return;  // (only with a device-side row count below the grid's: block-uniform, before any barrier)
  chain_cu_body<NS, RB, WPS, SILU>(ch, X, ldx, R, 32 * rb0, R, nloc);
}

// ---------------------------------------------------------------------------------------------------------------
// The layer loop of the SchNet backbone in ONE launch (geossl_schnet_layer_loop).  Everything between the filter
// network and the heads is local to a molecule: the row-local Linear chains (this file) and the neighbour aggregation
// (aggregate_reg.h) alternate, and neither ever reads a row of another molecule.  A block of four waves therefore owns a
// few molecules (<= 96 atom rows, <= RB row blocks) through ALL operations: chain of stages over its rows (wave m =
// column block m, as in k_row_chain_cu), block barrier, aggregation of its molecules (wave w = every fourth molecule,
// as in k_aggregate_reg), barrier, next chain ...  No grid-wide dependency exists, so 12 + 14 launches of a step become
// two - and the blocks need not be in the same phase: the second half of the grid starts `stagger` sleeps late, so that
// one half is in its (latency-bound, HBM idle) chain while the other half streams filter rows at full HBM speed.
struct LoopStage {  // a stage of GeosslChain with what the loop does not use fixed at compile time
  const uint32_t* image;
  const float* bias;
  const float* res;
  const float* tprev;
  float* out;
  int flags, pad;
  static constexpr int ld = 128, ldxin = 0;
  static constexpr const float* xin = nullptr;
  static constexpr float* out_act = nullptr;
};
struct LoopOp {
  const float* X;   // chain: input rows; aggregation: x
  const float* Wf;  // aggregation: filter rows of the block
  float* out;       // aggregation: destination rows
  int kind, nstage, swap, pad;
  LoopStage st[3];  // (chain_cu_body reads ch.st[s]: the operation is its own chain)
};
constexpr int LOOP_MAX_OPS = 14;
struct LoopArgs {
  LoopOp op[LOOP_MAX_OPS];
  const int4* plan;  // per block: first row, end row, first molecule, end molecule
  const int32_t* mol_ptr;
  const int32_t* pair_ptr;
  const uint8_t* pair_flag;
  int nops, R, stagger, pad;
  int nmol, mols_per_block;  // ragged form (NMAX = 0): block b owns molecules [b * mols_per_block, ...) of nmol
};

// NMAX > 0: uniform batch, blocks from the host's plan, the walk of one size class (a wave per molecule).  NMAX = 0:
// RAGGED molecules (1 .. 33 atoms) of a small batch - block b owns `mols_per_block` (1 or 2: at most 66 atom rows =
// three row blocks) consecutive molecules, found from mol_ptr on the device (no host plan: the loop then also serves a
// capacity bucket, whose index structures are device data), and aggregates each with all four waves
// (aggregate_block_body).  At 1024 molecules per view a block-owns-its-molecules loop loses to the separate launches
// (14 launches rebalance 14 times: DESIGN.md section 7); at the reference's batch size every launch of the pass is a
// 5 .. 9 us latency and the loop is what removes them.
// NW = 8 (the WIDE form, round 6): a block of eight waves for a RAGGED launch with at most one block per CU (the
// reference's batch size: 256 molecule-views on 256 CUs).  The pass lasts as long as the block with the largest molecule,
// whose aggregation is a serial sequence of round trips per wave (two targets in flight, nine targets per wave at 33
// atoms).  Waves 4 .. 7 take every other target of the aggregations - half the round trips per wave - and shadow the
// chains' barriers (a chain of NS plain stages crosses 2 NS LDS barriers: chain_cu_body); the chain itself stays the
// four-wave body, one column block per wave.  Set B at 128 molecules per view: 0.778 -> 0.749 ms per step.
template <int NMAX, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void k_layer_loop(LoopArgs a) {
  constexpr int F = 128;
  int4 pl;
  if constexpr (NMAX > 0) {
    pl = a.plan[blockIdx.x];
  } else {
    pl.z = (int)blockIdx.x * a.mols_per_block;
    pl.w = min(pl.z + a.mols_per_block, a.nmol);
    if (pl.z >= pl.w) return;  // (block-uniform, before any barrier)
    pl.x = a.mol_ptr[pl.z];
    pl.y = a.mol_ptr[pl.w];
    if (pl.x >= pl.y) return;
  }
  const int nloc = (pl.y - pl.x + 31) / 32;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  if ((int)blockIdx.x >= ((int)gridDim.x + 1) / 2)
    for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  LOOP_MARK(1);
  for (int o = 0; o < a.nops; ++o) {
    const LoopOp& op = a.op[o];
    LOOP_MARK(10 + op.kind);
    if (op.kind == 0) {
      if (NW > 4 && wave >= 4) {
        for (int i = 0; i < 2 * op.nstage; ++i) __builtin_amdgcn_s_barrier();  // (the barriers of chain_cu_body, nothing else)
      } else if (op.nstage == 1) chain_cu_body<1, 3, 2, false>(op, op.X, F, a.R, pl.x, pl.y, nloc);
      else if (op.nstage == 2) chain_cu_body<2, 3, 2, false>(op, op.X, F, a.R, pl.x, pl.y, nloc);
      else chain_cu_body<3, 3, 2, false>(op, op.X, F, a.R, pl.x, pl.y, nloc);
    } else {
      if constexpr (NMAX > 0 && NW == 4) {
        for (int mm = pl.z + wave; mm < pl.w; mm += 4) {
          const int a0 = a.mol_ptr[mm], n = a.mol_ptr[mm + 1] - a0, base = a.pair_ptr[mm];
          aggregate_reg_body<NMAX>(op.X, op.Wf, a.pair_flag, a0, n, base, lane, 2 * lane, F, op.swap, op.out);
        }
      } else {
        // ragged form (and every wide launch): the block's molecules one after the other, each by all waves of the block
        // (aggregate_block_body: a wave sums the target atoms w, w + NW, ... - no wave waits for another one's serial walk)
        extern __shared__ __attribute__((aligned(16))) uint8_t loop_smem[];
        for (int mm = pl.z; mm < pl.w; ++mm) {
          const int a0 = a.mol_ptr[mm], n = a.mol_ptr[mm + 1] - a0, base = a.pair_ptr[mm];
          if (mm > pl.z) __syncthreads();  // (the staging area is reused)
          aggregate_block_body<NW>(op.X, op.Wf, a.pair_flag, a0, n, base, op.swap, op.out, loop_smem);
        }
      }
    }
    // the next operation reads what this one wrote (rows of the block's own molecules, through L2), and reuses the LDS
    LOOP_MARK(20);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LOOP_MARK(21);
    __syncthreads();
  }
  LOOP_MARK(2);
}

template <int KS>
int launch_chain(const GeosslChain& ch, const float* X, int ldx, int64_t R, hipStream_t stream,
                 const int32_t* dyn_R = nullptr) {
  constexpr int NMB = KS / 2, CHUNK_BYTES = KS * 3 * 1024;
  if (dyn_R != nullptr && KS != 8) return (int)hipErrorInvalidValue;  // device-side row counts: weight-stationary form only
  const int nrb = (int)((R + 31) / 32), ngroups = (nrb + 3) / 4;
  const int nslot = chain_slots(ch.nstage * NMB);
  static const bool four_waves = getenv("GEOSSL_CHAIN4") != nullptr;  // the four-wave form, kept for A/B runs
  static const bool eight_waves = getenv("GEOSSL_CHAIN8") != nullptr;  // the streaming eight-wave form, for A/B runs
  // weight-stationary form: F = 128 (its images are in the two-fp16-piece format: no other kernel reads them); every
  // row-piece offset must fit the 32-bit range of a buffer descriptor: longer inputs run as two launches of half the rows
  const bool cu_form = KS == 8;
  bool same_input = false, silu = false;
  for (int s2 = 0; s2 < ch.nstage; ++s2) silu |= (ch.st[s2].flags & (GEOSSL_EPI_SILU | GEOSSL_EPI_MUL_DSILU)) != 0;
  if (ch.nstage > 3) silu = true;  // chains of four and five stages are instantiated in that form only
  if (silu && !cu_form) return (int)hipErrorInvalidValue;  // silu epilogues, long chains: weight-stationary form only
  for (int s2 = 0; s2 < ch.nstage; ++s2) {
    if (cu_form && ((int64_t)R * ch.st[s2].ld * 4 >= (int64_t)0xFFFFFF00u || (int64_t)R * ldx * 4 >= (int64_t)0xFFFFFF00u)) {
      if (dyn_R != nullptr) return (int)hipErrorInvalidValue;  // (two launches of half the rows: by-value counts only)
      const int64_t r0 = ((R / 2 + 31) / 32) * 32;
      GeosslChain hi = ch;
      for (int s3 = 0; s3 < ch.nstage; ++s3) {
        GeosslChainStage& st = hi.st[s3];
        if (st.out != nullptr) st.out += r0 * st.ld;
        if (st.out_act != nullptr) st.out_act += r0 * st.ld;
        if (st.res != nullptr) st.res += r0 * st.ld;
        if (st.tprev != nullptr) st.tprev += r0 * st.ld;
        if (st.xin != nullptr) st.xin += r0 * st.ldxin;
      }
      const int rc = launch_chain<KS>(ch, X, ldx, r0, stream);
      return rc != 0 ? rc : launch_chain<KS>(hi, X + r0 * ldx, ldx, R - r0, stream);
    }
    same_input |= (ch.st[s2].flags & (GEOSSL_CHAIN_SAME_INPUT | GEOSSL_CHAIN_NEW_INPUT | GEOSSL_CHAIN_ADD_PREV)) != 0;
    if ((ch.st[s2].flags & GEOSSL_CHAIN_NEW_INPUT) && (ch.st[s2].xin == nullptr || ch.st[s2].ldxin < 16 * KS ||
                                                        (ch.st[s2].ldxin & 3)))
      return (int)hipErrorInvalidValue;
  }
  // the stage-input flags exist in the weight-stationary form only, and not on the first stage
  if (same_input && (!cu_form || (ch.st[0].flags & (GEOSSL_CHAIN_SAME_INPUT | GEOSSL_CHAIN_NEW_INPUT | GEOSSL_CHAIN_ADD_PREV))))
    return (int)hipErrorInvalidValue;
  if constexpr (KS == 8) {
    static const bool one_per_cu_env = getenv("GEOSSL_CHAIN_CU1") != nullptr;  // 512-register form, one block per CU
    const bool one_per_cu = one_per_cu_env && !silu;
    static const int slots_env = getenv("GEOSSL_CHAIN_SLOTS") != nullptr ? atoi(getenv("GEOSSL_CHAIN_SLOTS")) : 0;  // experiments
    const int RBV = one_per_cu ? 5 : 3, slots = slots_env > 0 ? slots_env : (one_per_cu ? 256 : 512);
    const int need = (nrb + RBV - 1) / RBV;                       // blocks so that none takes more than RB row blocks
    const int fill = nrb < slots ? nrb : slots;                   // blocks so that every slot of the chip has work
    const int grid = need > fill ? need : fill;
    const size_t lds = (size_t)RBV * KS * 2 * 1024 + (size_t)ch.nstage * 16 * KS * sizeof(float) + (size_t)RBV * 128 * sizeof(float);
#define LAUNCH_CU(NSV)                                                                                                 \
  do {                                                                                                                 \
    if (silu) {                                                                                                        \
      allow_big_lds(&k_row_chain_cu<NSV, 3, 2, true>);                                                                 \
      hipLaunchKernelGGL((k_row_chain_cu<NSV, 3, 2, true>), dim3(grid), dim3(256), lds, stream, ch, X, ldx, (int)R, dyn_R);   \
    } else if (one_per_cu) {                                                                                           \
      allow_big_lds(&k_row_chain_cu<NSV, 5, 1, false>);                                                                \
      hipLaunchKernelGGL((k_row_chain_cu<NSV, 5, 1, false>), dim3(grid), dim3(256), lds, stream, ch, X, ldx, (int)R, dyn_R);  \
    } else {                                                                                                           \
      allow_big_lds(&k_row_chain_cu<NSV, 3, 2, false>);                                                                \
      hipLaunchKernelGGL((k_row_chain_cu<NSV, 3, 2, false>), dim3(grid), dim3(256), lds, stream, ch, X, ldx, (int)R, dyn_R);  \
    }                                                                                                                  \
  } while (0)
#define LAUNCH_CU_LONG(NSV)                                                                                            \
  do {                                                                                                                 \
    allow_big_lds(&k_row_chain_cu<NSV, 3, 2, true>);                                                                   \
    hipLaunchKernelGGL((k_row_chain_cu<NSV, 3, 2, true>), dim3(grid), dim3(256), lds, stream, ch, X, ldx, (int)R, dyn_R);     \
  } while (0)
    switch (ch.nstage) {
      case 1: LAUNCH_CU(1); break;
      case 2: LAUNCH_CU(2); break;
      case 3: LAUNCH_CU(3); break;
      case 4: LAUNCH_CU_LONG(4); break;
      case 5: LAUNCH_CU_LONG(5); break;
      default: return (int)hipErrorInvalidValue;
    }
#undef LAUNCH_CU_LONG
#undef LAUNCH_CU
  } else if (four_waves) {  // (F = 64 / 32 only: the streaming forms are not instantiated for F = 128, whose images they cannot read)
    const int grid = ngroups < 2048 ? ngroups : 2048;  // two blocks per CU are resident (72 KB of LDS, <= 256 registers)
    const size_t lds = (size_t)nslot * CHUNK_BYTES + (size_t)ch.nstage * 16 * KS * sizeof(float);
#define LAUNCH_NS(NSV)                                                                                           \
  do {                                                                                                           \
    allow_big_lds(&k_row_chain<KS, NSV>);                                                                        \
    hipLaunchKernelGGL((k_row_chain<KS, NSV>), dim3(grid), dim3(256), lds, stream, ch, X, ldx, (int)R);          \
  } while (0)
    switch (ch.nstage) {
      case 1: LAUNCH_NS(1); break;
      case 2: LAUNCH_NS(2); break;
      case 3: LAUNCH_NS(3); break;
      default: return (int)hipErrorInvalidValue;
    }
#undef LAUNCH_NS
  } else {
    // one block per CU; per round a block takes 4 row blocks (regular waves) + 1 shared by its team waves
    const int want = (nrb + 4) / 5;
    const int grid = want < 256 ? want : 256;
    const size_t lds = (size_t)(nslot + 2) * CHUNK_BYTES + (size_t)ch.nstage * 16 * KS * sizeof(float);
#define LAUNCH_NS8(NSV)                                                                                          \
  do {                                                                                                           \
    allow_big_lds(&k_row_chain8<KS, NSV>);                                                                       \
    hipLaunchKernelGGL((k_row_chain8<KS, NSV>), dim3(grid), dim3(512), lds, stream, ch, X, ldx, (int)R);         \
  } while (0)
    switch (ch.nstage) {
      case 1: LAUNCH_NS8(1); break;
      case 2: LAUNCH_NS8(2); break;
      case 3: LAUNCH_NS8(3); break;
      default: return (int)hipErrorInvalidValue;
    }
#undef LAUNCH_NS8
  }
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

}  // namespace

#ifdef LOOP_TIMING
extern "C" int geossl_loop_debug_read(long long* host, int* n, int reset) {
  int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(loop_dbg), sizeof(long long) * 2 * 1024);
  if (rc == 0) rc = (int)hipMemcpyFromSymbol(n, HIP_SYMBOL(loop_dbg_n), sizeof(int));
  if (rc == 0 && reset) {
    const int zero = 0;
    rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(loop_dbg_n), &zero, sizeof(int));
  }
  return rc;
}
#endif
#ifdef CHAIN_TIMING
extern "C" int geossl_chain_debug_read(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(chain_dbg), sizeof(long long) * 8 * 64);
}
#endif

extern "C" int64_t geossl_chain_image_words(int F) {
  if (F != 32 && F != 64 && F != 128) return 0;
  return (int64_t)(F / 32) * (F / 16) * 3 * 64 * 4;
}

extern "C" int geossl_chain_prepare(const GeosslPrepareBatch* batch, int nprob, int F, int transB, hipStream_t stream) {
  if (nprob <= 0) return 0;
  if (nprob > GEOSSL_PREPARE_MAX || geossl_chain_image_words(F) == 0) return (int)hipErrorInvalidValue;
  for (int z = 0; z < nprob; ++z)
    if (batch->ldw[z] != 0 && (batch->ldw[z] < F || (batch->ldw[z] & 3) || ((uintptr_t)batch->W[z] & 15)))
      return (int)hipErrorInvalidValue;
  const int KS = F / 16, nitems = (F / 32) * KS * 64;
  dim3 grid((nitems + 255) / 256, nprob);
  if (KS == 8) hipLaunchKernelGGL((k_chain_prepare<8>), grid, dim3(256), 0, stream, *batch, F, transB);
  else if (KS == 4) hipLaunchKernelGGL((k_chain_prepare<4>), grid, dim3(256), 0, stream, *batch, F, transB);
  else hipLaunchKernelGGL((k_chain_prepare<2>), grid, dim3(256), 0, stream, *batch, F, transB);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_linear_chain(const float* X, int ldx, const GeosslChain* chain, int64_t R, int F,
                                   hipStream_t stream) {
  return geossl_linear_chain_dyn(X, ldx, chain, R, F, nullptr, stream);
}

extern "C" int geossl_linear_chain_dyn(const float* X, int ldx, const GeosslChain* chain, int64_t R, int F,
                                       const int32_t* dyn_R, hipStream_t stream) {
  if (R <= 0) return 0;
  if (chain == nullptr || chain->nstage < 1 || chain->nstage > GEOSSL_CHAIN_MAX || geossl_chain_image_words(F) == 0)
    return (int)hipErrorInvalidValue;
  if (ldx < F || (ldx & 3)) return (int)hipErrorInvalidValue;
  for (int s = 0; s < chain->nstage; ++s) {
    const GeosslChainStage& st = chain->st[s];
    if (st.image == nullptr) return (int)hipErrorInvalidValue;
    if ((st.out != nullptr || st.res != nullptr || st.tprev != nullptr || st.out_act != nullptr) && (st.ld < F || (st.ld & 3)))
      return (int)hipErrorInvalidValue;
  }
  if (F == 128) return launch_chain<8>(*chain, X, ldx, R, stream, dyn_R);
  if (F == 64) return launch_chain<4>(*chain, X, ldx, R, stream, dyn_R);
  return launch_chain<2>(*chain, X, ldx, R, stream, dyn_R);
}

static int fill_loop_ops(const GeosslLoopOp* ops, int nops, int F, LoopArgs& a) {
  for (int o = 0; o < nops; ++o) {
    const GeosslLoopOp& src = ops[o];
    LoopOp& d = a.op[o];
    d.kind = src.kind; d.swap = src.swap; d.X = src.X; d.Wf = src.Wf; d.out = src.out; d.pad = 0;
    d.nstage = src.kind == 0 ? src.chain.nstage : 0;
    if (src.X == nullptr) return (int)hipErrorInvalidValue;
    if (src.kind == 0) {
      if (src.chain.nstage < 1 || src.chain.nstage > 3) return (int)hipErrorInvalidValue;
      for (int s2 = 0; s2 < 3; ++s2) {
        LoopStage& ls = d.st[s2];
        if (s2 < src.chain.nstage) {
          const GeosslChainStage& st = src.chain.st[s2];
          // plain stages only: dense [N][F] operands, no stage-input flags, no silu forms
          if (st.image == nullptr || (st.flags & ~(GEOSSL_EPI_BIAS | GEOSSL_EPI_SSP | GEOSSL_EPI_RESIDUAL | GEOSSL_EPI_MUL_DSSP)) ||
              ((st.out != nullptr || st.res != nullptr || st.tprev != nullptr) && st.ld != F) || st.xin != nullptr ||
              st.out_act != nullptr)
            return (int)hipErrorInvalidValue;
          ls.image = st.image; ls.bias = st.bias; ls.res = st.res; ls.tprev = st.tprev; ls.out = st.out; ls.flags = st.flags;
        } else {
          ls.image = nullptr; ls.bias = nullptr; ls.res = nullptr; ls.tprev = nullptr; ls.out = nullptr; ls.flags = 0;
        }
        ls.pad = 0;
      }
    } else if (src.kind == 1) {
      if (src.Wf == nullptr || src.out == nullptr) return (int)hipErrorInvalidValue;
      for (int s2 = 0; s2 < 3; ++s2) d.st[s2] = LoopStage{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    } else {
      return (int)hipErrorInvalidValue;
    }
  }
  return 0;
}

// The wide form (eight waves per block) pays when a block has its CU to itself: at most one block per CU, one molecule per
// block.  GEOSSL_LOOP_NO_WIDE: the four-wave forms of round 5 (A/B timing).
static bool loop_wide(int64_t nblocks, bool one_molecule_per_block) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    cus = (hipGetDevice(&dev) == hipSuccess &&
           hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : -1;
  }
  return one_molecule_per_block && cus > 0 && nblocks <= cus && getenv("GEOSSL_LOOP_NO_WIDE") == nullptr;
}

extern "C" int geossl_schnet_layer_loop(const GeosslLoopOp* ops, int nops, const int32_t* plan, int nblocks,
                                        const int32_t* mol_ptr, const int32_t* pair_ptr, const uint8_t* pair_flag,
                                        int max_n, int uniform, int64_t N, int F, int stagger, hipStream_t stream) {
  if (nops <= 0 || nblocks <= 0) return 0;
  if (ops == nullptr || plan == nullptr || nops > LOOP_MAX_OPS || F != 128 || !uniform || max_n > 20 || N <= 0)
    return (int)hipErrorInvalidValue;
  if (N * (int64_t)F * 4 >= (int64_t)0xFFFFFF00u) return (int)hipErrorInvalidValue;  // 32-bit buffer offsets
  LoopArgs a;
  const int rc = fill_loop_ops(ops, nops, F, a);
  if (rc != 0) return rc;
  a.plan = reinterpret_cast<const int4*>(plan);
  a.mol_ptr = mol_ptr; a.pair_ptr = pair_ptr; a.pair_flag = pair_flag;
  a.nops = nops; a.R = (int)N; a.stagger = stagger; a.pad = 0; a.nmol = 0; a.mols_per_block = 0;
  const size_t lds = (size_t)3 * 8 * 2 * 1024 + (size_t)3 * 128 * sizeof(float) + (size_t)3 * 128 * sizeof(float);
  // uniform batches only (every molecule has max_n atoms, says the caller): the walk of one size class.  A form that
  // holds every class (waves of a block on different walks, molecules of 27 atoms and more shared by two or four waves)
  // was built and measured on set B: 650-670 us per pass against 470 us as separate launches - removed.
  // (the wide form - eight waves per block, loop_wide - is for the ragged launch only: on equal-sized molecules the
  // register walk of ONE wave reads every filter row once, 8.7 us per aggregation of 256 x 18 atoms, where eight waves on
  // the block form read it twice and take 10.7: at that size the aggregation is bound by fetching the 20 MB of filter rows)
  if (max_n <= 18) {
    allow_big_lds(&k_layer_loop<18>);
    hipLaunchKernelGGL((k_layer_loop<18>), dim3(nblocks), dim3(256), lds, stream, a);
  } else {
    allow_big_lds(&k_layer_loop<20>);
    hipLaunchKernelGGL((k_layer_loop<20>), dim3(nblocks), dim3(256), lds, stream, a);
  }
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

// The layer loop over RAGGED molecules of a small batch (k_layer_loop<0>): B molecules of 1 .. 33 atoms described by
// mol_ptr / pair_ptr on the device, `mols_per_block` (1 or 2) consecutive molecules per block, one round of blocks
// (ceil(B / mols_per_block) <= 512 * 2: two blocks of four waves per CU).  N = rows of the atom tensors (a capacity is
// fine: no row past mol_ptr[B] is touched).
extern "C" int geossl_schnet_layer_loop_ragged(const GeosslLoopOp* ops, int nops, const int32_t* mol_ptr,
                                               const int32_t* pair_ptr, const uint8_t* pair_flag, int64_t B,
                                               int mols_per_block, int64_t N, int F, hipStream_t stream) {
  if (nops <= 0 || B <= 0) return 0;
  if (ops == nullptr || mol_ptr == nullptr || pair_ptr == nullptr || nops > LOOP_MAX_OPS || F != 128 || N <= 0 ||
      mols_per_block < 1 || mols_per_block > 2)
    return (int)hipErrorInvalidValue;
  if (N * (int64_t)F * 4 >= (int64_t)0xFFFFFF00u) return (int)hipErrorInvalidValue;  // 32-bit buffer offsets
  const int64_t nblocks = (B + mols_per_block - 1) / mols_per_block;
  if (nblocks > 1024) return (int)hipErrorInvalidValue;
  LoopArgs a;
  const int rc = fill_loop_ops(ops, nops, F, a);
  if (rc != 0) return rc;
  a.plan = nullptr;
  a.mol_ptr = mol_ptr; a.pair_ptr = pair_ptr; a.pair_flag = pair_flag;
  a.nops = nops; a.R = (int)N; a.stagger = 0; a.pad = 0; a.nmol = (int)B; a.mols_per_block = mols_per_block;
  const size_t lds = (size_t)3 * 8 * 2 * 1024 + (size_t)3 * 128 * sizeof(float) + (size_t)3 * 128 * sizeof(float);
  if (loop_wide(nblocks, mols_per_block == 1)) {
    allow_big_lds(&k_layer_loop<0, 8>);
    hipLaunchKernelGGL((k_layer_loop<0, 8>), dim3((unsigned)nblocks), dim3(512), lds, stream, a);
  } else {
    allow_big_lds(&k_layer_loop<0>);
    hipLaunchKernelGGL((k_layer_loop<0>), dim3((unsigned)nblocks), dim3(256), lds, stream, a);
  }
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
