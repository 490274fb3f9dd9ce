// Column GEMM (weight-gradient) template shared by the atom-row, pair-row and super-edge-row kernels.
//
// For problem z:  dW_z[m][n] = sum_r A_z[r][m] * B_z[r][n]   (r < R, reduced over rows)
//                 db_z[m]    = sum_r A_z[r][m]                (optional)
//                 dd_z[m]    = sum_r A_z[r][m] * e_z[r]       (optional, e = a per-row scalar)
// A "Loader" materialises 64-row slices of A and B (and e) in LDS, so operands that are cheap functions of
// saved state (ssp outputs, Gaussian smearing of a distance, gathered atom rows) are rebuilt on the fly instead
// of being written to HBM first.  Stage 1 writes one partial per row chunk, stage 2 sums the partials in chunk
// order — no atomics, bit-reproducible.
#pragma once
#include "common.h"
#include "geossl_hip.h"

namespace geossl {

struct TnOut {
  float* dW[GEOSSL_TN_MAX];
  float* db[GEOSSL_TN_MAX];
  float* dd[GEOSSL_TN_MAX];
};

template <int NCM, int NCN, class Loader>
__global__ __launch_bounds__(256) void k_tn(Loader ld, int R, int chunk, int M, int N, float* __restrict__ partial,
                                            float* __restrict__ partial_bias, float* __restrict__ partial_dot) {
  constexpr int MP = 32 * NCM, NP = 32 * NCN, T = NCM * NCN, TPW = (T + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [64][MP]
  float* Bs = smem + 64 * MP;        // [64][NP]
  float* es = smem + 64 * (MP + NP); // [64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int z = blockIdx.y;
  const int row_begin = blockIdx.x * chunk, row_end = min(R, row_begin + chunk);
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  float bsum = 0.0f, dsum = 0.0f;
  for (int row0 = row_begin; row0 < row_end; row0 += 64) {
    ld.template load<MP, NP>(z, row0, row_end, M, N, As, Bs, es, tid);
    __syncthreads();
    {
      // operands of the wave's tiles for one k-step: NA distinct A fragments (M blocks), NB distinct B fragments
      constexpr int NB_ = (T >= 4 ? (NCN >= 4 ? 1 : (4 / NCN > TPW ? TPW : 4 / NCN)) : 1);
      (void)NB_;
      float af[2][TPW], bf[2][TPW];
      auto fetch = [&](int kk, float (&a)[TPW], float (&b)[TPW]) {
        const float* ap = As + (2 * kk + kh) * MP + j;
        const float* bp = Bs + (2 * kk + kh) * NP + j;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int t = wave + 4 * i;
          a[i] = t < T ? ap[32 * (t / NCN)] : 0.0f;
          b[i] = t < T ? bp[32 * (t % NCN)] : 0.0f;
        }
      };
      fetch(0, af[0], bf[0]);
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) {
        if (kk + 1 < 32) fetch(kk + 1, af[(kk + 1) & 1], bf[(kk + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int t = wave + 4 * i;
          if (t < T) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][i], bf[kk & 1][i], acc[i], 0, 0, 0);
        }
      }
    }
    if (partial_bias != nullptr && tid < MP) {
#pragma unroll 8
      for (int r = 0; r < 64; ++r) bsum += As[r * MP + tid];
    }
    if (partial_dot != nullptr && tid < MP) {
#pragma unroll 8
      for (int r = 0; r < 64; ++r) dsum += As[r * MP + tid] * es[r];
    }
    __syncthreads();
  }
  const size_t pb = (size_t)z * gridDim.x + blockIdx.x;
  float* Pp = partial + pb * M * N;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = wave + 4 * i;
    if (t >= T) continue;
    const int mb = t / NCN, nb = t % NCN, n = 32 * nb + j;
    if (n >= N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * mb + c_row(r, lane);
      if (m < M) Pp[(size_t)m * N + n] = acc[i][r];
    }
  }
  if (partial_bias != nullptr && tid < M) partial_bias[pb * M + tid] = bsum;
  if (partial_dot != nullptr && tid < M) partial_dot[pb * M + tid] = dsum;
}

// out_z[(i/ncols)*ld + (i%ncols)*cstride] = (accumulate ? out : 0) + sum_b partial[z][b][i]
__global__ void k_reduce_partials(GeosslReduceBatch batch, const float* __restrict__ partial, int nblk, int len,
                                  int ncols, int ld, int cstride, int accumulate);

inline int64_t tn_workspace_floats(int64_t R, int M, int N, int nprob) {
  int chunk, nblk;
  geossl_tn_plan(R, nprob, &chunk, &nblk);
  return (int64_t)nprob * nblk * ((int64_t)M * N + 2 * M);
}

// dW rows are written with leading dimension dW_ld (>= N); dd is written with stride dd_stride (e.g. the last
// column of a [M][N+1] weight: dW_ld = dd_stride = N+1).
template <class Loader>
int launch_tn(const Loader& ld, int nprob, int64_t R, int M, int N, const TnOut& out, int dW_ld, int dd_stride,
              float* workspace, int accumulate, hipStream_t stream) {
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
  const int NCM = (M + 31) / 32, NCN = (N + 31) / 32;
  const size_t lds = ((size_t)64 * 32 * (NCM + NCN) + 64) * sizeof(float);
  dim3 grid(nblk, nprob);
#define GEOSSL_TN_LAUNCH(a, b)                                                                                   \
  do {                                                                                                           \
    static bool attr_set = false;                                                                                \
    if (!attr_set) {                                                                                             \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tn<a, b, Loader>),                                    \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                               \
      attr_set = true;                                                                                           \
    }                                                                                                            \
    hipLaunchKernelGGL((k_tn<a, b, Loader>), grid, dim3(256), lds, stream, ld, (int)R, chunk, M, N, partial,     \
                       any_b ? pbias : nullptr, any_d ? pdot : nullptr);                                         \
  } while (0)
  if (NCM == 4 && NCN == 4) GEOSSL_TN_LAUNCH(4, 4);
  else if (NCM == 4 && NCN == 2) GEOSSL_TN_LAUNCH(4, 2);
  else if (NCM == 2 && NCN == 4) GEOSSL_TN_LAUNCH(2, 4);
  else if (NCM == 2 && NCN == 2) GEOSSL_TN_LAUNCH(2, 2);
  else if (NCM == 1 && NCN == 1) GEOSSL_TN_LAUNCH(1, 1);
  else if (NCM == 1 && NCN == 2) GEOSSL_TN_LAUNCH(1, 2);
  else if (NCM == 2 && NCN == 1) GEOSSL_TN_LAUNCH(2, 1);
  else if (NCM == 4 && NCN == 1) GEOSSL_TN_LAUNCH(4, 1);
  else if (NCM == 1 && NCN == 4) GEOSSL_TN_LAUNCH(1, 4);
  else return (int)hipErrorInvalidValue;
#undef GEOSSL_TN_LAUNCH
  GEOSSL_CHECK_LAUNCH();
  GeosslReduceBatch rb;
  for (int z = 0; z < GEOSSL_TN_MAX; ++z) rb.out[z] = z < nprob ? out.dW[z] : nullptr;
  const int len = M * N;
  hipLaunchKernelGGL(k_reduce_partials, dim3((len + 63) / 64, nprob), dim3(256), 0, stream, rb, partial, nblk, len, N,
                     dW_ld, 1, accumulate);
  GEOSSL_CHECK_LAUNCH();
  if (any_b) {
    for (int z = 0; z < GEOSSL_TN_MAX; ++z) rb.out[z] = z < nprob ? out.db[z] : nullptr;
    hipLaunchKernelGGL(k_reduce_partials, dim3((M + 63) / 64, nprob), dim3(256), 0, stream, rb, pbias, nblk, M, M, M,
                       1, accumulate);
    GEOSSL_CHECK_LAUNCH();
  }
  if (any_d) {
    for (int z = 0; z < GEOSSL_TN_MAX; ++z) rb.out[z] = z < nprob ? out.dd[z] : nullptr;
    hipLaunchKernelGGL(k_reduce_partials, dim3((M + 63) / 64, nprob), dim3(256), 0, stream, rb, pdot, nblk, M, M,
                       M, dd_stride, accumulate);
    GEOSSL_CHECK_LAUNCH();
  }
  return 0;
}

// 64-row slice of a row-major [R][ld] matrix (first ncols columns, ncols % 4 == 0, ld % 4 == 0) -> dst[64][NPAD],
// 16-byte loads and stores, zero fill past row_end / ncols.
template <int NPAD>
__device__ __forceinline__ void load_rows_f4(const float* __restrict__ src, int ld, int ncols, int row0, int row_end,
                                             float* dst, int tid) {
  constexpr int Q = NPAD / 4;
#pragma unroll 4
  for (int i = tid; i < 64 * Q; i += 256) {
    const int r = i / Q, c4 = i - r * Q, row = row0 + r;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (row < row_end && 4 * c4 < ncols) v = *reinterpret_cast<const float4*>(src + (size_t)row * ld + 4 * c4);
    *reinterpret_cast<float4*>(dst + r * NPAD + 4 * c4) = v;
  }
}

// Plain loader: A_z [R][M], B_z [R][N] row-major in global memory (M, N multiples of 4).
struct PlainLoader {
  GeosslTnBatch batch;
  int lda, ldb;  // row strides of A_z / B_z (>= M / N)
  template <int MP, int NP>
  __device__ __forceinline__ void load(int z, int row0, int row_end, int M, int N, float* As, float* Bs, float* es,
                                       int tid) const {
    load_rows_f4<MP>(batch.A[z], lda, M, row0, row_end, As, tid);
    load_rows_f4<NP>(batch.B[z], ldb, N, row0, row_end, Bs, tid);
  }
};

}  // namespace geossl
