// Continuous-filter network forward for all interaction blocks (K3): InteractionBlock.mlp applied inside
// CFConv.forward, schnet.py:141-145,186-187 (with GaussianSmearing, :205-207, folded in):
//     Wf_l[p] = ( ssp( rbf(d_p) A1_l^T + b1_l ) A2_l^T + b2_l ) * C(d_p)          for every pair slot p, layer l.
//
// Column-split waves with register-resident weights: wave w of a block owns output columns [32w, 32w+32) of BOTH
// GEMMs and keeps its slices of A1_l (G x 32) and A2_l (F x 32) as MFMA B fragments in registers for the whole
// launch (a block serves one layer).  The only LDS tile is the hidden activation t of the block's 128 pair rows
// (64 KB at F = 128), so two blocks fit per CU and one block's ssp / store phase overlaps the other's MFMA phase:
//   GEMM1  every wave evaluates the Gaussian smearing of all 128 rows directly in A-fragment layout (one exp per
//          row block and k-step) and multiplies by its own 32 hidden columns;
//   ssp    in C layout; t goes to LDS (swizzled, conflict-free) and, when training, to HBM;
//   GEMM2  one register B fragment feeds four independent MFMAs (the four 32-row A fragments come from LDS);
//   out    times the envelope C(d), 128-byte row segments to HBM.
#include "common.h"
#include "geossl_hip.h"

using namespace geossl;

namespace {

template <typename K>
inline void allow_big_lds(K kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
}

// NW = F/32 waves per block; K1 = k-steps of GEMM1 (>= ceil(G/2))
template <int NW, int K1>
__global__ __launch_bounds__(64 * NW, 2) void k_filter_fwd(const float* __restrict__ pair_d,
                                                           const float* __restrict__ pair_c, int P,
                                                           GeosslFilterWeights w, int G,
                                                           const float* __restrict__ offset, float coeff,
                                                           float* __restrict__ Tout, float* __restrict__ Wf) {
  constexpr int F = 32 * NW, K2 = F / 2, NT = 64 * NW;
  constexpr int TS = 129;  // row stride of the k-major hidden-activation tile: C-layout writes (lanes = k) and
                           // A-fragment reads (lanes = rows) are both conflict-free, and every A read of GEMM2 is
                           // base + compile-time offset (no per-step address arithmetic)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* At = smem;            // [F (k)][TS] hidden activations of the tile's 128 rows, k-major
  float* cw = At + F * TS;     // [128] envelope of the tile's rows
  float* stage = cw + 128 + (threadIdx.x >> 6) * 512;  // wave-private 16x32 transposition stage for wide stores
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int l = blockIdx.y;
  const int col = 32 * wave + j;
  // ---- second-layer weight slice of this wave as B fragments: bw2[kk] = A2[col][2kk+kh]
  float bw2[K2];
  {
    const float4* w2 = reinterpret_cast<const float4*>(w.w2[l] + (size_t)col * F);
#pragma unroll
    for (int q = 0; q < K2 / 2; ++q) {
      const float4 v = w2[q];
      bw2[2 * q] = kh ? v.y : v.x;
      bw2[2 * q + 1] = kh ? v.w : v.z;
    }
  }
  const float* __restrict__ w1p = w.w1[l] + (size_t)col * G;  // first-layer row of this lane's column (L1 resident)
  const float b1c = w.b1[l][col], b2c = w.b2[l][col];
  const size_t lbase = (size_t)l * P;
  const int ntiles = (P + 127) / 128;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int r0 = t * 128;
    float d[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int row = r0 + 32 * rb + j;
      d[rb] = row < P ? pair_d[row] : 0.0f;
    }
    for (int i = tid; i < 128; i += NT) cw[i] = r0 + i < P ? pair_c[r0 + i] : 0.0f;
    f32x16 acc[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = b1c;
    // GEMM1: Gaussian smearing (schnet.py:206-207) in A-fragment layout x this wave's columns of A1^T
#pragma unroll 2
    for (int kk = 0; kk < K1; ++kk) {
      const int k = 2 * kk + kh;
      const bool ok = k < G;
      const float off = ok ? offset[k] : 0.0f, b = ok ? w1p[k] : 0.0f;
      float a[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const float diff = d[rb] - off;
        a[rb] = ok ? __expf(coeff * (diff * diff)) : 0.0f;
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb], b, acc[rb], 0, 0, 0);
    }
    // ssp; hidden activation to LDS (k-major) and, when training, to HBM (16-byte stores via the LDS stage)
    {
      float* tcol = At + col * TS + 4 * kh;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float tv = ssp(acc[rb][r]);
          acc[rb][r] = tv;
          tcol[32 * rb + (r & 3) + 8 * (r >> 2)] = tv;
        }
        if (Tout != nullptr)
          store_c_block_x4(Tout + (lbase + r0 + 32 * rb) * F + 32 * wave, F, P - (r0 + 32 * rb), stage, lane,
                           [&](int r) { return acc[rb][r]; });
        __builtin_amdgcn_sched_barrier(0);  // one row block at a time keeps the live set small
      }
    }
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = b2c;
    // GEMM2: four 32-row A fragments from LDS per k-step against one register B fragment
    {
      const float* abase = At + kh * TS + j;  // A[row = 32rb+j][k = 2kk+kh] = abase[kk*2*TS + 32*rb]
      float a_cur[4], a_nxt[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a_cur[rb] = abase[32 * rb];
#pragma unroll
      for (int kk = 0; kk < K2; ++kk) {
        constexpr int last = K2 - 1;
        const int kn = kk < last ? kk + 1 : kk;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a_nxt[rb] = abase[kn * 2 * TS + 32 * rb];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[rb], bw2[kk], acc[rb], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a_cur[rb] = a_nxt[rb];
      }
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      store_c_block_x4(Wf + (lbase + r0 + 32 * rb) * F + 32 * wave, F, P - (r0 + 32 * rb), stage, lane, [&](int r) {
        return acc[rb][r] * cw[32 * rb + (r & 3) + 8 * (r >> 2) + 4 * kh];  // schnet.py:187
      });
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int geossl_cfconv_filter_fwd(const float* pair_d, const float* pair_c, int64_t P,
                                        const GeosslFilterWeights* w, int L, int F, int G, const float* offset,
                                        float coeff, float* T, float* Wf, hipStream_t stream) {
  if (P <= 0 || L <= 0) return 0;
  if (L > GEOSSL_MAX_L || (F != 32 && F != 64 && F != 128) || G > 64) return (int)hipErrorInvalidValue;
  const int ntiles = (int)((P + 127) / 128);
  int per_layer = 512 / L;  // two blocks per CU
  if (per_layer < 1) per_layer = 1;
  if (per_layer > ntiles) per_layer = ntiles;
  dim3 grid(per_layer, L);
  const size_t lds = ((size_t)F * 129 + 128 + 4 * 512) * sizeof(float);
#define LAUNCH(NW, K1)                                                                                          \
  do {                                                                                                          \
    allow_big_lds(&k_filter_fwd<NW, K1>);                                                                       \
    hipLaunchKernelGGL((k_filter_fwd<NW, K1>), grid, dim3(64 * NW), lds, stream, pair_d, pair_c, (int)P, *w, G, \
                       offset, coeff, T, Wf);                                                                   \
  } while (0)
#define LAUNCH_F(NW)                    \
  do {                                  \
    if (G <= 8) LAUNCH(NW, 4);          \
    else if (G <= 52) LAUNCH(NW, 26);   \
    else LAUNCH(NW, 32);                \
  } while (0)
  if (F == 128) LAUNCH_F(4); else if (F == 64) LAUNCH_F(2); else LAUNCH_F(1);
#undef LAUNCH_F
#undef LAUNCH
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
