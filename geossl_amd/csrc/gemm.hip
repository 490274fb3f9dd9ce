// Dense atom-row kernels: Linear forward / backward-input (row GEMM against a weight tile held in LDS)
// and Linear backward-weight (column GEMM: dW = dY^T X, reduced over rows, two-stage deterministic).
//
// Replaces the ATen Linear call sites of the hot path: schnet.py:99,101,166,189,191 and their autograd.
#include "common.h"
#include "geossl_hip.h"
#include "tn.h"

using namespace geossl;

// ------------------------------------------------------------------------------------------------
// Row GEMM.  Y[r][n] = epi( sum_k X[r][k] * B[k][n] ), r < R, n < 32*NC*gridDim.y.
//   transB = 1: W is torch-layout [NO][K]  -> B[k][n] = W[n][k]   (Linear forward)
//   transB = 0: W is [K][NO] row-major      -> B[k][n] = W[k][n]   (Linear backward-input: dX = dY W)
// A fragments come straight from global memory: lane (row j, half kh) loads 16 B = X[r0+j][8q+4kh .. +3]
// and feeds them to four consecutive MFMA k-steps, B rows taken in the same (permuted) k order, so the
// sum runs over the same products in a fixed order and no LDS staging of X is needed.  Needs K % 8 == 0.
template <int NC>
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ X, const float* __restrict__ W,
                                                const float* __restrict__ bias, const float* __restrict__ res,
                                                const float* __restrict__ tprev, float* __restrict__ Y, int R, int K,
                                                int NO, int ldx, int ldy, int transB, int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NB = 32 * NC;  // columns handled by this block
  constexpr int BS = NB + 1;   // odd row stride: the transposing store (lanes = consecutive k) is conflict-free
  float* Bs = smem;            // [K][BS]
  float* stage = smem + ((K * BS + 3) & ~3) + (threadIdx.x >> 6) * 512;  // wave-private 16x32 transposition stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.y * NB;
  // stage the weight slice: 16-byte global loads, several in flight per thread
  if (transB) {
    const int KQ = K / 4;
#pragma unroll 4
    for (int i = tid; i < NB * KQ; i += 256) {
      const int n = i / KQ, k4 = i - n * KQ;
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (n0 + n < NO) v = *reinterpret_cast<const float4*>(W + (size_t)(n0 + n) * K + 4 * k4);
      float* d = Bs + (4 * k4) * BS + n;
      d[0] = v.x; d[BS] = v.y; d[2 * BS] = v.z; d[3 * BS] = v.w;
    }
  } else {
    constexpr int NQ = NB / 4;
#pragma unroll 4
    for (int i = tid; i < K * NQ; i += 256) {
      const int k = i / NQ, n4 = i - k * NQ;
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (n0 + 4 * n4 < NO) v = *reinterpret_cast<const float4*>(W + (size_t)k * NO + n0 + 4 * n4);
      float* d = Bs + k * BS + 4 * n4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
  __syncthreads();
  const int j = lane & 31, kh = lane >> 5;
  const int ntiles = (R + 127) / 128;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int r0 = t * 128 + wave * 32;
    if (r0 >= R) continue;
    f32x16 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float b = (flags & GEOSSL_EPI_BIAS) && (n0 + 32 * c + j < NO) ? bias[n0 + 32 * c + j] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = b;
    }
    const int arow = min(r0 + j, R - 1);  // clamp: rows past R are computed and discarded
    const float4* xp = reinterpret_cast<const float4*>(X + (size_t)arow * ldx + 4 * kh);
#pragma unroll 2
    for (int q = 0; q < K / 8; ++q) {
      const float4 a4 = xp[2 * q];
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* bp = Bs + (8 * q + 4 * kh + s) * BS + j;
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bp[32 * c], acc[c], 0, 0, 0);
      }
    }
    // epilogue on 16-byte row pieces (accumulator blocks transposed through the wave's LDS stage): bias, ssp,
    // ssp' and residual are applied with wide loads, the result leaves with wide stores
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int cb = n0 + 32 * c;
      transpose_c_block(stage, lane, [&](int r) { return acc[c][r]; }, [&](int rl, int c4, float4 v) {
        const int row = r0 + rl, col = cb + c4;
        if (row >= R || col >= NO) return;
        const size_t o = (size_t)row * ldy + col;
        if (flags & GEOSSL_EPI_SSP) { v.x = ssp(v.x); v.y = ssp(v.y); v.z = ssp(v.z); v.w = ssp(v.w); }
        if (flags & GEOSSL_EPI_MUL_DSSP) {
          const float4 tp = *reinterpret_cast<const float4*>(tprev + o);
          v.x *= dssp_from_out(tp.x); v.y *= dssp_from_out(tp.y); v.z *= dssp_from_out(tp.z); v.w *= dssp_from_out(tp.w);
        }
        if (flags & GEOSSL_EPI_RESIDUAL) {
          const float4 rs = *reinterpret_cast<const float4*>(res + o);
          v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
        }
        *reinterpret_cast<float4*>(Y + o) = v;
      });
    }
  }
}

extern "C" int geossl_linear(const float* X, int ldx, const float* W, const float* bias, const float* res,
                             const float* tprev, float* Y, int ldy, int64_t R, int K, int NO, int transB, int flags,
                             hipStream_t stream) {
  if (R <= 0) return 0;
  if (K % 8 != 0 || K > 256 || NO > 256) return (int)hipErrorInvalidValue;
  const int ntiles = (int)((R + 127) / 128);
  const int NOp = (NO + 31) / 32 * 32;
  // split the output columns over gridDim.y when there are too few row tiles to fill 256 CUs
  int NC = (NOp % 128 == 0) ? 4 : ((NOp % 64 == 0) ? 2 : 1);
  if (NC == 4 && ntiles < 512) NC = 2;
  const int ny = NOp / (32 * NC);
  dim3 grid(ntiles < 1024 ? ntiles : 1024, ny);
  const size_t lds = ((size_t)((K * (32 * NC + 1) + 3) & ~3) + 4 * 512) * sizeof(float);
  if (ldx < K || ldy < NO || (ldx & 3) || (ldy & 3) || (NO & 3)) return (int)hipErrorInvalidValue;
#define LAUNCH(NCV)                                                                                               \
  do {                                                                                                            \
    if (lds > 64 * 1024)                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear<NCV>),                                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                          \
    hipLaunchKernelGGL((k_linear<NCV>), grid, dim3(256), lds, stream, X, W, bias, res, tprev, Y, (int)R, K, NO,   \
                       ldx, ldy, transB, flags);                                                                  \
  } while (0)
  switch (NC) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 4: LAUNCH(4); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef LAUNCH
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Column GEMM (weight gradient), plain operands — template in tn.h.
namespace geossl {
// block = 64 outputs x 4 slices of the partial list; the four slice sums are combined in slice order.
__global__ __launch_bounds__(256) void k_reduce_partials(GeosslReduceBatch batch, const float* __restrict__ partial,
                                                         int nblk, int len, int ncols, int ld, int cstride,
                                                         int accumulate) {
  __shared__ float red[4][64];
  const int z = blockIdx.y;
  float* out = batch.out[z];
  if (out == nullptr) return;
  const float* p = partial + (size_t)z * nblk * len;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int per = (nblk + 3) / 4, b0 = slice * per, b1 = min(nblk, b0 + per);
  float s = 0.0f;
  if (i < len) {
#pragma unroll 8
    for (int b = b0; b < b1; ++b) s += p[(size_t)b * len + i];
  }
  red[slice][lane] = s;
  __syncthreads();
  if (slice == 0 && i < len) {
    const size_t o = (size_t)(i / ncols) * ld + (size_t)(i % ncols) * cstride;
    float v = accumulate ? out[o] : 0.0f;
    v += red[0][lane];
    v += red[1][lane];
    v += red[2][lane];
    v += red[3][lane];
    out[o] = v;
  }
}
}  // namespace geossl

extern "C" void geossl_tn_plan(int64_t R, int nprob, int* chunk, int* nblk) {
  // about 1024 row chunks over all problems of the launch (4 per CU), each a multiple of 64 rows
  int target = 1024 / (nprob > 0 ? nprob : 1);
  if (target < 32) target = 32;
  if (target > 512) target = 512;
  int64_t c = (R + target - 1) / target;
  c = (c + 63) / 64 * 64;
  if (c < 64) c = 64;
  *chunk = (int)c;
  *nblk = (int)((R + c - 1) / c);
  if (*nblk < 1) *nblk = 1;
}

extern "C" int64_t geossl_tn_workspace_floats(int64_t R, int M, int N, int nprob) {
  return tn_workspace_floats(R, M, N, nprob);
}

extern "C" int geossl_linear_wgrad(const GeosslTnBatch* batch, int nprob, int64_t R, int M, int N, int lda, int ldb,
                                   int ldw, float* workspace, int accumulate, hipStream_t stream) {
  if (lda < M || ldb < N || ldw < N || (lda & 3) || (ldb & 3) || (M & 3) || (N & 3)) return (int)hipErrorInvalidValue;
  PlainLoader ld;
  ld.batch = *batch;
  ld.lda = lda;
  ld.ldb = ldb;
  TnOut out;
  for (int z = 0; z < GEOSSL_TN_MAX; ++z) {
    out.dW[z] = batch->dW[z];
    out.db[z] = batch->db[z];
    out.dd[z] = nullptr;
  }
  return launch_tn(ld, nprob, R, M, N, out, ldw, 1, workspace, accumulate, stream);
}

extern "C" int geossl_abi_version(void) { return GEOSSL_ABI_VERSION; }
