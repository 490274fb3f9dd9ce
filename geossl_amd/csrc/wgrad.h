// Column GEMM (weight gradient) on the matrix pipe with two fp16 pieces per operand (split.h), shared by the atom-row
// Linear layers and the NCSN head:       dW_z[m][n] = sum_r A_z[r][m] B_z[r][n]      (r < R, reduced over rows)
//                  db_z[m]    = sum_r A_z[r][m]                 (optional)
//                  dd_z[m]    = sum_r A_z[r][m] e_z[r]          (optional, e = a per-row scalar)
// The contraction runs over rows, so both operands are wanted feature-major (lane = column, 8 rows per lane): each
// lane obtains its column of 8 rows with eight 4-byte loads (a half-wave reads 128 contiguous bytes of a row),
// splits them in registers and publishes the fragments in LDS; every wave then multiplies its tiles.  An "Ops"
// policy says how an operand element is obtained, so operands that are cheap functions of saved state (a relu mask
// times a per-row scalar, a sum of two gathered atom rows) are rebuilt on the fly instead of being written to HBM
// first.  32 rows per iteration, the requests of the next iteration fly during the MFMAs; one partial per row chunk,
// summed in chunk order by k_reduce_partials (no atomics, bit-reproducible).
// Operand scales: every 32-column operand block carries a running power-of-two exponent (the largest magnitude its
// converting wave has met so far), published next to its fragments; an accumulator tile follows the sum of its two
// blocks' exponents and is scaled down by the difference when one of them grows.
//
// Ops interface (row = row0 + 16 ks + 8 kh + e for element (ks, e) of a lane in half kh; `col` = operand column):
//   struct RawA / RawB                     per-lane request state of one 32-row x 32-column operand block
//   prime_a / prime_b (z, col, row0, row_end, kh, raw)       once, before the first request (index prefetch)
//   request_a / request_b (z, col, row0, row_end, kh, raw)   issue the loads (clamped addresses, no predication)
//   finish_a (z, col, row0, row_end, kh, raw, out[2][8], e[2][8])   values after arrival; e only if kDot
//   finish_b (z, col, row0, row_end, kh, raw, out[2][8])
// finish_* must return 0 for rows >= row_end; columns past M / N are zeroed by the kernel.
#pragma once
#include "common.h"
#include "geossl_hip.h"
#include "split.h"
#include "tn.h"

namespace geossl {

struct WgradOut {
  float* dW[GEOSSL_TN_MAX];
  float* db[GEOSSL_TN_MAX];
  float* dd[GEOSSL_TN_MAX];
};

// PIECES = 2: two fp16 pieces per operand under running power-of-two block scales (3 MFMAs per product, 22-bit products:
// the default).  PIECES = 3: three bf16 pieces, no scales (bf16 has fp32's exponent range), 6 MFMAs per product, 24-bit
// products - the form of GEOSSL_ARITH_24BIT (every dense product of the step at fp32's own product width; A/B and
// accuracy runs).
template <int NCM, int NCN, class Ops, int PIECES = 2>
__global__ __launch_bounds__(256, 2) void k_wgrad_split(Ops ops, int R, int chunk, int M, int N,
                                                        float* __restrict__ partial, float* __restrict__ partial_bias,
                                                        float* __restrict__ partial_dot,
                                                        const int32_t* __restrict__ dyn_R, int nblk, int nprob) {
  R = dyn_count(R, dyn_R);  // (row chunks past the real rows write zero partials)
  // Which (problem z, row chunk c) this block is.  Consecutive block ids go round the eight XCDs, each with an L2 of its
  // own; problems of one launch often share an operand (the three column blocks of a Dense(F, 3F) gradient read the same
  // input rows, the two halves of mu_channel_mix's the same mu rows).  So a chunk belongs to ONE XCD (c mod 8) and all
  // problems' blocks of that chunk follow each other there: they run at the same time and the shared rows come from
  // that L2 once.  nblk < 0: the plain order (z = blockIdx.y, c = blockIdx.x; GEOSSL_WGRAD_PLAIN_ORDER, A/B timing).
  int z, cidx;
  if (nblk < 0) {
    nblk = (int)gridDim.x;
    z = blockIdx.y;
    cidx = blockIdx.x;
  } else {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    z = slot % nprob;
    cidx = 8 * (slot / nprob) + xcd;
    if (cidx >= nblk) return;  // (the chunk count rounded up to a multiple of eight: block-uniform, before any barrier)
  }
  constexpr int SA = (NCM + 3) / 4, SB = (NCN + 3) / 4;  // operand blocks converted per wave: A block wave + 4u
  constexpr int T = NCM * NCN, TPW = (T + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  u32x4* Fr = reinterpret_cast<u32x4*>(smem_raw);  // [NCM + NCN][2 k-steps][PIECES][64]: A blocks first, then B blocks
  int* eblk = reinterpret_cast<int*>(Fr + (size_t)(NCM + NCN) * 2 * PIECES * 64);  // [NCM + NCN] running exponents
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int row_begin = cidx * chunk, row_end = min(R, row_begin + chunk);
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  float bsum[SA], dsum[SA];
#pragma unroll
  for (int u = 0; u < SA; ++u) bsum[u] = dsum[u] = 0.0f;
  int ea[SA], eb[SB], eacc[TPW];  // running exponents: of the blocks this wave converts, of its accumulator tiles
#pragma unroll
  for (int u = 0; u < SA; ++u) ea[u] = -100;
#pragma unroll
  for (int u = 0; u < SB; ++u) eb[u] = -100;
#pragma unroll
  for (int i = 0; i < TPW; ++i) eacc[i] = -200;
  // scale of an operand block for this tile: wave-wide largest magnitude -> running exponent -> 2^(14 - e)
  auto block_scale = [&](const float (&v)[2][8], int& e_run) __attribute__((always_inline)) {
    float mx = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int q = 0; q < 8; ++q) mx = fmaxf(mx, fabsf(v[ks][q]));
    mx = wave_max(mx);
    e_run = max(e_run, __builtin_amdgcn_readfirstlane(mag_exponent(mx)));
    return __builtin_amdgcn_ldexpf(1.0f, 14 - e_run);
  };
  // operand requests run PF row tiles ahead of their use (register sets in rotation); the block barriers are LDS-only,
  // so the requests stay in flight across them
#ifndef WGRAD_PF
#define WGRAD_PF 2
#endif
  constexpr int PF = Ops::kDot ? 1 : WGRAD_PF;  // (the gathering policy keeps index state in its request set: one ahead)
  typename Ops::RawA ra_[PF][SA];
  typename Ops::RawB rb_[PF][SB];
  auto request_into = [&](int row0, typename Ops::RawA (&ra)[SA], typename Ops::RawB (&rb)[SB]) {
#pragma unroll
    for (int u = 0; u < SA; ++u)
      if (wave + 4 * u < NCM) ops.request_a(z, min(32 * (wave + 4 * u) + j, M - 1), row0, row_end, kh, ra[u]);
#pragma unroll
    for (int u = 0; u < SB; ++u)
      if (wave + 4 * u < NCN) ops.request_b(z, min(32 * (wave + 4 * u) + j, N - 1), row0, row_end, kh, rb[u]);
  };
  if (row_begin < row_end) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {  // every request set carries its own per-column state
#pragma unroll
      for (int u = 0; u < SA; ++u)
        if (wave + 4 * u < NCM) ops.prime_a(z, min(32 * (wave + 4 * u) + j, M - 1), row_begin, row_end, kh, ra_[p][u]);
#pragma unroll
      for (int u = 0; u < SB; ++u)
        if (wave + 4 * u < NCN) ops.prime_b(z, min(32 * (wave + 4 * u) + j, N - 1), row_begin, row_end, kh, rb_[p][u]);
    }
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (row_begin + 32 * p < row_end) request_into(row_begin + 32 * p, ra_[p], rb_[p]);
  }
  // one 32-row tile: operands of set (ra, rb) -> fragments -> MFMAs; the set is refilled with tile row0 + 32 PF
  auto tile = [&](int row0, typename Ops::RawA (&ra)[SA], typename Ops::RawB (&rb)[SB]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < SA; ++u) {
      const int blk = wave + 4 * u;
      if (blk >= NCM) continue;
      const int col = 32 * blk + j;
      float v[2][8], e[2][8];
      ops.finish_a(z, min(col, M - 1), row0, row_end, kh, ra[u], v, e);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (col >= M) v[ks][q] = 0.0f;
          bsum[u] += v[ks][q];
          if (Ops::kDot) dsum[u] = fmaf(v[ks][q], e[ks][q], dsum[u]);
        }
      if constexpr (PIECES == 3) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const Frag3 f = split8(v[ks]);
          u32x4* dst = Fr + (size_t)((blk * 2 + ks) * 3) * 64 + lane;
          dst[0] = f.h;
          dst[64] = f.m;
          dst[128] = f.l;
        }
      } else {
        const float sc = block_scale(v, ea[u]);
        if (lane == 0) eblk[blk] = ea[u];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const Frag2 f = split8h_scaled(v[ks], sc);
          u32x4* dst = Fr + (size_t)((blk * 2 + ks) * 2) * 64 + lane;
          dst[0] = f.h;
          dst[64] = f.l;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int blk = wave + 4 * u;
      if (blk >= NCN) continue;
      const int col = 32 * blk + j;
      float v[2][8];
      ops.finish_b(z, min(col, N - 1), row0, row_end, kh, rb[u], v);
      if (col >= N) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int q = 0; q < 8; ++q) v[ks][q] = 0.0f;
      }
      if constexpr (PIECES == 3) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const Frag3 f = split8(v[ks]);
          u32x4* dst = Fr + (size_t)(((NCM + blk) * 2 + ks) * 3) * 64 + lane;
          dst[0] = f.h;
          dst[64] = f.m;
          dst[128] = f.l;
        }
      } else {
        const float sc = block_scale(v, eb[u]);
        if (lane == 0) eblk[NCM + blk] = eb[u];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const Frag2 f = split8h_scaled(v[ks], sc);
          u32x4* dst = Fr + (size_t)(((NCM + blk) * 2 + ks) * 2) * 64 + lane;
          dst[0] = f.h;
          dst[64] = f.l;
        }
      }
    }
    lds_barrier();
    if (row0 + 32 * PF < row_end) request_into(row0 + 32 * PF, ra, rb);  // in flight for PF tiles
    // the accumulator tiles follow their operand blocks' exponents
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int t = wave + 4 * i;
      if (t >= T || PIECES == 3) continue;
      const int en = __builtin_amdgcn_readfirstlane(eblk[t / NCN] + eblk[NCM + t % NCN]);
      if (en != eacc[i]) {
        const float f = __builtin_amdgcn_ldexpf(1.0f, eacc[i] - en);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] *= f;
        eacc[i] = en;
      }
    }
    if constexpr (PIECES == 3) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int t = wave + 4 * i;
          if (t >= T) continue;
          const u32x4* sa = Fr + (size_t)(((t / NCN) * 2 + ks) * 3) * 64 + lane;
          const u32x4* sb = Fr + (size_t)(((NCM + t % NCN) * 2 + ks) * 3) * 64 + lane;
          Frag3 a3, b3;
          a3.h = sa[0]; a3.m = sa[64]; a3.l = sa[128];
          b3.h = sb[0]; b3.m = sb[64]; b3.l = sb[128];
          mma6(acc[i], a3, b3);
        }
    } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // two tiles at a time: their twelve MFMAs alternate between the two accumulators
#pragma unroll
      for (int i0 = 0; i0 < TPW; i0 += 2) {
        Frag2 af[2], bf[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int t = wave + 4 * (i0 + d);
          if (i0 + d >= TPW || t >= T) continue;
          const u32x4* sa = Fr + (size_t)(((t / NCN) * 2 + ks) * 2) * 64 + lane;
          const u32x4* sb = Fr + (size_t)(((NCM + t % NCN) * 2 + ks) * 2) * 64 + lane;
          af[d].h = sa[0]; af[d].l = sa[64];
          bf[d].h = sb[0]; bf[d].l = sb[64];
        }
#define GEOSSL_WG_STEP(pa, pb)                                                              \
  _Pragma("unroll") for (int d = 0; d < 2; ++d) {                                           \
    if (i0 + d < TPW && wave + 4 * (i0 + d) < T)                                            \
      acc[i0 + d] = mfma_f16(af[d].pa, bf[d].pb, acc[i0 + d]);                              \
  }
        GEOSSL_WG_STEP(l, h)
        GEOSSL_WG_STEP(h, l)
        GEOSSL_WG_STEP(h, h)
#undef GEOSSL_WG_STEP
      }
    }
    }
    lds_barrier();
  };
  for (int row0 = row_begin; row0 < row_end; row0 += 32 * PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (row0 + 32 * p < row_end) tile(row0 + 32 * p, ra_[p], rb_[p]);  // (block-uniform condition)
  }
  const size_t pb = (size_t)z * nblk + cidx;
  float* Pp = partial + pb * M * N;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = wave + 4 * i;
    if (t >= T) continue;
    const int mb = t / NCN, nb = t % NCN, n = 32 * nb + j;
    if (n >= N) continue;
    const float kk = PIECES == 3 ? 1.0f : __builtin_amdgcn_ldexpf(1.0f, eacc[i] - 28);  // undo the two operand scales
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * mb + c_row(r, lane);
      if (m < M) Pp[(size_t)m * N + n] = acc[i][r] * kk;
    }
  }
#pragma unroll
  for (int u = 0; u < SA; ++u) {
    const int blk = wave + 4 * u;
    if (blk >= NCM) continue;
    const float sb = bsum[u] + __shfl_xor(bsum[u], 32, 64);
    const float sd = dsum[u] + __shfl_xor(dsum[u], 32, 64);
    if (kh == 0 && 32 * blk + j < M) {
      if (partial_bias != nullptr) partial_bias[pb * M + 32 * blk + j] = sb;
      if (partial_dot != nullptr) partial_dot[pb * M + 32 * blk + j] = sd;
    }
  }
}

// GEOSSL_ARITH_24BIT (read per call: bench.py times both forms in one process): the three-piece form
inline bool arith_24bit() { return getenv("GEOSSL_ARITH_24BIT") != nullptr; }

// workspace: nprob * nblk * (M*N + 2*M) floats (geossl_tn_workspace_floats).  dW rows are written with leading
// dimension dW_ld (>= N); dd is written with stride dd_stride (e.g. the last column of a [M][N+1] weight).
template <int NCM, int NCN, class Ops>
int launch_wgrad_split(const Ops& ops, int nprob, int64_t R, int M, int N, const WgradOut& out, int dW_ld, int dd_stride,
                       float* workspace, int accumulate, hipStream_t stream, const int32_t* dyn_R = nullptr) {
  if (nprob <= 0 || R <= 0) return 0;
  if (nprob > GEOSSL_TN_MAX) return (int)hipErrorInvalidValue;
  int chunk, nblk;
  geossl_tn_plan(R, nprob, &chunk, &nblk);
  float* partial = workspace;
  float* pbias = partial + (size_t)nprob * nblk * M * N;
  float* pdot = pbias + (size_t)nprob * nblk * M;
  bool any_b = false, any_d = false;
  for (int z = 0; z < nprob; ++z) {
    any_b |= out.db[z] != nullptr;
    any_d |= out.dd[z] != nullptr;
  }
  static const bool plain_order = getenv("GEOSSL_WGRAD_PLAIN_ORDER") != nullptr;
  const dim3 grid = plain_order ? dim3(nblk, nprob) : dim3((unsigned)(8 * ((nblk + 7) / 8) * nprob));
  const int nblk_arg = plain_order ? -1 : nblk;
  if (arith_24bit()) {  // GEOSSL_ARITH_24BIT: three bf16 pieces, six MFMAs per product
    const size_t lds = (size_t)(NCM + NCN) * 2 * 3 * 1024 + (size_t)(NCM + NCN) * sizeof(int);
    allow_big_lds(&k_wgrad_split<NCM, NCN, Ops, 3>);
    hipLaunchKernelGGL((k_wgrad_split<NCM, NCN, Ops, 3>), grid, dim3(256), lds, stream, ops, (int)R, chunk, M,
                       N, partial, any_b ? pbias : nullptr, any_d ? pdot : nullptr, dyn_R, nblk_arg, nprob);
  } else {
    const size_t lds = (size_t)(NCM + NCN) * 2 * 2 * 1024 + (size_t)(NCM + NCN) * sizeof(int);
    allow_big_lds(&k_wgrad_split<NCM, NCN, Ops>);
    hipLaunchKernelGGL((k_wgrad_split<NCM, NCN, Ops>), grid, dim3(256), lds, stream, ops, (int)R, chunk, M,
                       N, partial, any_b ? pbias : nullptr, any_d ? pdot : nullptr, dyn_R, nblk_arg, nprob);
  }
  GEOSSL_CHECK_LAUNCH();
  ReduceMulti rm;  // dW, db and dd partial sums in one launch
  rm.add(partial, M * N, N, dW_ld, 1, out.dW, nprob);
  if (any_b) rm.add(pbias, M, M, M, 1, out.db, nprob);
  if (any_d) rm.add(pdot, M, M, M, dd_stride, out.dd, nprob);
  hipLaunchKernelGGL(k_reduce_multi, dim3(rm.blocks(), nprob), dim3(256), 0, stream, rm, nblk, accumulate);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

// helper: eight rows of one column of a row-major matrix, clamped addresses
__device__ __forceinline__ void request_col8(const float* __restrict__ src, int ld, int col, int row0, int row_end,
                                             int kh, float (&v)[2][8]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e)  // every operand element is read exactly once: non-temporal
      v[ks][e] = __builtin_nontemporal_load(src + (size_t)min(row0 + 16 * ks + 8 * kh + e, row_end - 1) * ld + col);
}
// helper: pin the arrived values (common.h) and zero the rows past row_end
__device__ __forceinline__ void finish_col8(int row0, int row_end, int kh, const float (&raw)[2][8],
                                            float (&out)[2][8]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = pin(raw[ks][e]);
      out[ks][e] = row0 + 16 * ks + 8 * kh + e < row_end ? t : 0.0f;
    }
}

// A_z [R][M], B_z [R][N] row-major in global memory
struct PlainOps {
  GeosslTnBatch batch;
  int lda, ldb;
  static constexpr bool kDot = false;
  struct RawA { float v[2][8]; };
  struct RawB { float v[2][8]; };
  __device__ __forceinline__ void prime_a(int, int, int, int, int, RawA&) const {}
  __device__ __forceinline__ void prime_b(int, int, int, int, int, RawB&) const {}
  __device__ __forceinline__ void request_a(int z, int col, int row0, int row_end, int kh, RawA& r) const {
    request_col8(batch.A[z], lda, col, row0, row_end, kh, r.v);
  }
  __device__ __forceinline__ void request_b(int z, int col, int row0, int row_end, int kh, RawB& r) const {
    request_col8(batch.B[z], ldb, col, row0, row_end, kh, r.v);
  }
  __device__ __forceinline__ void finish_a(int, int, int row0, int row_end, int kh, RawA& r, float (&out)[2][8],
                                           float (&)[2][8]) const {
    finish_col8(row0, row_end, kh, r.v, out);
  }
  __device__ __forceinline__ void finish_b(int, int, int row0, int row_end, int kh, RawB& r, float (&out)[2][8]) const {
    finish_col8(row0, row_end, kh, r.v, out);
  }
};

}  // namespace geossl
