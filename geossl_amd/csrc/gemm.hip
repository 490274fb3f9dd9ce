// Dense atom-row kernels: Linear forward / backward-input (row GEMM against a weight tile held in LDS)
// and Linear backward-weight (column GEMM: dW = dY^T X, reduced over rows, two-stage deterministic).
//
// Replaces the ATen Linear call sites of the hot path: schnet.py:99,101,166,189,191 and their autograd.
#include "common.h"
#include "geossl_hip.h"
#include "split.h"
#include "tn.h"
#include "wgrad.h"

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

// ------------------------------------------------------------------------------------------------
// Row GEMM on the bf16 matrix pipe (split.h), evaluated transposed: Y^T = B^T X^T with the weight as the A
// operand (pre-split fragments in LDS, formatted once per block) and the rows of X on the lanes.  A wave owns 32
// rows.  The vector-memory pipe pays per 128-byte line an instruction touches, so nothing is loaded or stored
// "one row per lane" (32 lines per instruction): X, the epilogue operands and Y all move as fully coalesced
// 16-byte-per-lane accesses (8 lanes per 128-byte row segment, 8 lines per instruction) and are re-shaped between
// that layout and the MFMA layouts (B fragment: lane = row, 8 consecutive k; C: lane = row, 4 consecutive columns
// per register group) through a wave-private 32x32 LDS stage (row stride 36 floats: conflict-free both ways).
// Output columns are processed 32 at a time so the epilogue operands of a column block are requested before its
// MFMAs.  K % 32 == 0, K <= 128.
// Weight formatting shared by k_linear_split and k_linear_prepare.  A weight is cut into "items" of 8 contraction
// indices of one output column (= one lane of one A fragment).  transB (torch layout [NO][K]): item <-> (n, 8
// consecutive k), consecutive items are consecutive 32 bytes of W.  !transB ([K][NO]): item <-> fragment lane, the
// eight k are strided rows of W and a half-wave reads 128 contiguous bytes of each.  Addresses are clamped (no
// predicated loads); the caller zeroes columns past NO.
template <int KS>
__device__ __forceinline__ void load_weight_item(const float* __restrict__ W, int i, int n0, int NO, int transB,
                                                 float (&v)[8]) {
  constexpr int K = 16 * KS;
  if (transB) {
    const int n = min(n0 + i / (2 * KS), NO - 1), k8 = i % (2 * KS);
    const float4 lo = *reinterpret_cast<const float4*>(W + (size_t)n * K + 8 * k8);
    const float4 hi = *reinterpret_cast<const float4*>(W + (size_t)n * K + 8 * k8 + 4);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
    v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
  } else {
    const int ln = i & 63, ks = (i >> 6) % KS, mb = i / (64 * KS);
    const int n = min(n0 + 32 * mb + (ln & 31), NO - 1), k0 = 16 * ks + 8 * (ln >> 5);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = W[(size_t)(k0 + e) * NO + n];
  }
}
// column (relative to the first column of the item range), k-step and k half of item i
template <int KS>
__device__ __forceinline__ void weight_item_slot(int i, int transB, int& nl, int& ks, int& khh) {
  if (transB) {
    nl = i / (2 * KS);
    ks = (i % (2 * KS)) >> 1;
    khh = i & 1;
  } else {
    nl = 32 * (i / (64 * KS)) + (i & 31);
    ks = (i >> 6) % KS;
    khh = (i >> 5) & 1;
  }
}

// image_z[mb][ks][piece][lane] (u32x4) for every 32-column block mb of weight z
template <int KS>
__global__ __launch_bounds__(256) void k_linear_prepare(GeosslPrepareBatch batch, int NO, int transB) {
  const int z = blockIdx.y;
  const float* __restrict__ W = batch.W[z];
  u32x4* __restrict__ image = reinterpret_cast<u32x4*>(batch.image[z]);
  const int nitems = ((NO + 31) / 32) * KS * 64;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nitems) return;
  float v[8];
  load_weight_item<KS>(W, i, 0, NO, transB, v);
  int nl, ks, khh;
  weight_item_slot<KS>(i, transB, nl, ks, khh);
  if (nl >= NO) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.0f;
  }
  const Frag3 f = split8(v);
  u32x4* dst = image + ((size_t)((nl >> 5) * KS + ks) * 3) * 64 + (nl & 31) + 32 * khh;
  dst[0] = f.h;
  dst[64] = f.m;
  dst[128] = f.l;
}

constexpr int LSS = 36;  // row stride of the wave-private stage, floats
// E0 / E1: the launch uses tprev (ssp' epilogue) / res (residual) - compile-time, so unused operands cost nothing
template <int KS, bool E0, bool E1>
__global__ __launch_bounds__(512) void k_linear_split(const float* __restrict__ X, const float* __restrict__ W,
                                                      const float* __restrict__ bias, const float* __restrict__ res,
                                                      const float* __restrict__ tprev, float* __restrict__ Y, int R,
                                                      int NO, int nmb, int ldx, int ldy, int transB, int flags,
                                                      const u32x4* __restrict__ image) {
  constexpr int K = 16 * KS, NCH = KS / 2;  // NCH 32-column chunks of X
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  u32x4* Wf = reinterpret_cast<u32x4*>(smem_raw);  // [nmb][KS][3][64] A fragments of the block's weight columns
  float* bias_s = reinterpret_cast<float*>(Wf + (size_t)nmb * KS * 3 * 64);  // [32*nmb]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  float* stage = bias_s + 32 * nmb + wave * (32 * LSS);
  const int n0 = blockIdx.y * 32 * nmb;
  const int nrb = (R + 31) / 32;
  const int cr = lane >> 3, c4 = 4 * (lane & 7);  // coalesced access role: row 8u + cr, columns c4 .. c4+3
  // Work split.  The per-wave cost of a row block is MFMA + vector work on the same SIMD and two waves of a SIMD
  // do not overlap them, so the launch is balanced over SIMDs, not over waves: waves 0-3 of a block (one per
  // SIMD) take whole row blocks, round-robin over all blocks; the row blocks left over after the last full round
  // are cut into (row block, column block) tasks and given to waves 4-7.  All 8 waves format the weights.
  const int nprim = 4 * gridDim.x;
  const int n_whole = nrb >= nprim ? (nrb / nprim) * nprim : nrb;  // row blocks processed whole
  const int n_tasks = (nrb - n_whole) * nmb;                      // (row block, column block) tasks
  const bool primary = wave < 4;
  const int slot = blockIdx.x + gridDim.x * (wave & 3);
  int item = slot;  // primary: row block; secondary: task index
  auto item_rb = [&](int it) { return primary ? it : n_whole + it / nmb; };
  const int item_end = primary ? n_whole : n_tasks;
  // the first row block's X is requested before the weights are formatted (independent latencies overlap)
  f32x4 xr[NCH][4];
#define REQUEST_X(rbi)                                                                                          \
  do {                                                                                                          \
    _Pragma("unroll") for (int c_ = 0; c_ < NCH; ++c_) _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_)         \
        xr[c_][u_] = *reinterpret_cast<const f32x4*>(X + (size_t)min(32 * (rbi) + 8 * u_ + cr, R - 1) * ldx +  \
                                                      32 * c_ + c4);                                            \
  } while (0)
  if (item < item_end) REQUEST_X(item_rb(item));
  for (int i = tid; i < 32 * nmb; i += 512)
    bias_s[i] = (flags & GEOSSL_EPI_BIAS) && n0 + i < NO ? bias[n0 + i] : 0.0f;
  const int nitems = nmb * KS * 64;
  if (image != nullptr) {
    // prepared weights: the block's slice of the fragment image is copied as is (coalesced 16-byte loads, twelve in
    // flight per thread = the whole 96 KB image of a 128 x 128 weight in one round trip)
    const u32x4* src = image + (size_t)blockIdx.y * nitems * 3;
    for (int i0 = tid; i0 < 3 * nitems; i0 += 12 * 512) {
      u32x4 t[12];
#pragma unroll
      for (int u = 0; u < 12; ++u) t[u] = src[min(i0 + 512 * u, 3 * nitems - 1)];
#pragma unroll
      for (int u = 0; u < 12; ++u)
        if (i0 + 512 * u < 3 * nitems) Wf[i0 + 512 * u] = t[u];
    }
  } else {
    for (int i0 = tid; i0 < nitems; i0 += 4 * 512) {
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) load_weight_item<KS>(W, min(i0 + 512 * u, nitems - 1), n0, NO, transB, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 512 * u;
        int nl, ks, khh;
        weight_item_slot<KS>(i, transB, nl, ks, khh);
        const bool ok = n0 + nl < NO;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = pin(v[u][e]);
          v[u][e] = ok ? t : 0.0f;
        }
        if (i < nitems) {
          const Frag3 f = split8(v[u]);
          u32x4* dst = Wf + ((size_t)((nl >> 5) * KS + ks) * 3) * 64 + (nl & 31) + 32 * khh;
          dst[0] = f.h;
          dst[64] = f.m;
          dst[128] = f.l;
        }
      }
    }
  }
  __syncthreads();
  for (; item < item_end; item += nprim) {
    const int rb = item_rb(item);
    const int mb_begin = primary ? 0 : item % nmb, mb_end = primary ? nmb : mb_begin + 1;
    // X: coalesced registers -> stage -> B fragments (lane = row j, k = 16ks + 8kh + e)
    Frag3 xf[KS];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(stage + (8 * u + cr) * LSS + c4) = xr[c][u];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + j * LSS + 16 * s2 + 8 * kh);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + j * LSS + 16 * s2 + 8 * kh + 4);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        xf[2 * c + s2] = split8(v);
      }
    }
    // epilogue operands in the coalesced layout, requested one column block ahead of their use (before the MFMAs of
    // the previous block), clamped addresses
    f32x4 e0[4], e1[4], e0n[4], e1n[4];
    // LATE_RES (K = 128 with both operands): the residual is requested behind the block's MFMAs, not ahead of them
    constexpr bool LATE_RES = KS == 8 && E0 && E1;
    auto request_e = [&](int mb, f32x4 (&a)[4], f32x4 (&b)[4], bool want_a, bool want_b) {
      const int cc = min(n0 + 32 * mb + c4, NO - 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t oc = (size_t)min(32 * rb + 8 * u + cr, R - 1) * ldy + cc;
        if (E0 && want_a) a[u] = *reinterpret_cast<const f32x4*>(tprev + oc);
        if (E1 && want_b) b[u] = *reinterpret_cast<const f32x4*>(res + oc);
      }
    };
    // (K = 128: the eight X fragments alone are 96 registers - the operands of a column block are requested at its own
    // start, ahead of its MFMAs, not one block earlier: no second set of buffers, no spilled registers)
    constexpr bool AHEAD = KS < 8;
    if ((E0 || E1) && AHEAD) request_e(mb_begin, e0, e1, true, true);
    for (int mb = mb_begin; mb < mb_end; ++mb) {
      const int cb = n0 + 32 * mb;
      if constexpr (AHEAD) {
        if ((E0 || E1) && mb + 1 < mb_end) request_e(mb + 1, e0n, e1n, true, true);
      } else {
        if (E0 || E1) request_e(mb, e0, e1, true, !LATE_RES);
      }
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + 32 * mb + 4 * kh + 8 * q);
        acc[4 * q] = b.x;
        acc[4 * q + 1] = b.y;
        acc[4 * q + 2] = b.z;
        acc[4 * q + 3] = b.w;
      }
      Frag3 af, an;
      {
        const u32x4* src = Wf + ((size_t)(mb * KS) * 3) * 64 + lane;
        af.h = src[0]; af.m = src[64]; af.l = src[128];
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (AHEAD && ks + 1 < KS) {
          const u32x4* src = Wf + ((size_t)(mb * KS + ks + 1) * 3) * 64 + lane;
          an.h = src[0]; an.m = src[64]; an.l = src[128];
        }
        __builtin_amdgcn_sched_barrier(0);
        mma6(acc, af, xf[ks]);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < KS) {
          if constexpr (AHEAD) {
            af = an;
          } else {  // (K = 128: one set of weight fragments in flight)
            const u32x4* src = Wf + ((size_t)(mb * KS + ks + 1) * 3) * 64 + lane;
            af.h = src[0]; af.m = src[64]; af.l = src[128];
          }
        }
      }
      if constexpr (LATE_RES) request_e(mb, e0, e1, false, true);
      // C layout (lane = row j, columns 8q + 4kh + {0..3}) -> stage -> coalesced layout
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(stage + j * LSS + 8 * q + 4 * kh) =
            f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // arithmetic unconditional (keeps the operand loads up front), store predicated
        f32x4 v = *reinterpret_cast<const f32x4*>(stage + (8 * u + cr) * LSS + c4);
        if (flags & GEOSSL_EPI_SSP) { v.x = ssp(v.x); v.y = ssp(v.y); v.z = ssp(v.z); v.w = ssp(v.w); }
        if (E0) {
          v.x *= dssp_from_out(e0[u].x); v.y *= dssp_from_out(e0[u].y);
          v.z *= dssp_from_out(e0[u].z); v.w *= dssp_from_out(e0[u].w);
        }
        if (E1) { v.x += e1[u].x; v.y += e1[u].y; v.z += e1[u].z; v.w += e1[u].w; }
        const int row = 32 * rb + 8 * u + cr;
        if (row < R && cb + c4 < NO) *reinterpret_cast<f32x4*>(Y + (size_t)row * ldy + cb + c4) = v;
      }
      if constexpr (AHEAD) {
        if (E0 || E1) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            e0[u] = e0n[u];
            e1[u] = e1n[u];
          }
        }
      }
    }
    // a wave rarely has a second item (R > 32 rows x 4 waves x 256 blocks); its X is requested here rather than
    // before the MFMAs so that the 64 staging registers are not live across them
    if (item + nprim < item_end) REQUEST_X(item_rb(item + nprim));
  }
#undef REQUEST_X
}

// launch of the split kernel; W or image (exactly one non-null)
static int launch_linear_split(const float* X, int ldx, const float* W, const u32x4* image, const float* bias,
                               const float* res, const float* tprev, float* Y, int ldy, int64_t R, int K, int NO,
                               int transB, int flags, hipStream_t stream) {
  const int KS = K / 16;
  int nmb = (NO + 31) / 32;                 // 32-column blocks of the output
  const int cap = 108 / (3 * KS);            // weight fragments of one block <= 108 KB of LDS (stages: 36 KB)
  int ny = 1;
  while ((nmb + ny - 1) / ny > cap) ++ny;
  if (image != nullptr && nmb % ny != 0) return (int)hipErrorInvalidValue;  // image slices must tile evenly
  nmb = (nmb + ny - 1) / ny;
  const int nrb = (int)((R + 31) / 32);
  int nx = (nrb + 3) / 4;  // whole row blocks go to four waves per block (one per SIMD)
  if (nx > 256) nx = 256;
  const size_t lds = (size_t)nmb * KS * 3 * 1024 + (size_t)(nmb * 32 + 8 * 32 * LSS) * sizeof(float);
  const bool e0 = flags & GEOSSL_EPI_MUL_DSSP, e1 = flags & GEOSSL_EPI_RESIDUAL;
#define LAUNCH_SE(KSV, A, B)                                                                                      \
  do {                                                                                                            \
    allow_big_lds(&k_linear_split<KSV, A, B>);                                                                    \
    hipLaunchKernelGGL((k_linear_split<KSV, A, B>), dim3(nx, ny), dim3(512), lds, stream, X, W, bias, res, tprev, \
                       Y, (int)R, NO, nmb, ldx, ldy, transB, flags, image);                                       \
  } while (0)
#define LAUNCH_S(KSV)                                                                                             \
  do {                                                                                                            \
    if (e0 && e1) LAUNCH_SE(KSV, true, true);                                                                     \
    else if (e0) LAUNCH_SE(KSV, true, false);                                                                     \
    else if (e1) LAUNCH_SE(KSV, false, true);                                                                     \
    else LAUNCH_SE(KSV, false, false);                                                                            \
  } while (0)
  if (KS == 8) LAUNCH_S(8); else if (KS == 4) LAUNCH_S(4); else LAUNCH_S(2);
#undef LAUNCH_S
#undef LAUNCH_SE
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int64_t geossl_linear_image_words(int K, int NO) {
  if (!(K == 32 || K == 64 || K == 128) || NO <= 0 || NO > 256 || (NO & 3)) return 0;
  return (int64_t)((NO + 31) / 32) * (K / 16) * 3 * 64 * 4;
}

extern "C" int geossl_linear_prepare(const GeosslPrepareBatch* batch, int nprob, int K, int NO, int transB,
                                     hipStream_t stream) {
  if (nprob <= 0) return 0;
  if (nprob > GEOSSL_PREPARE_MAX || geossl_linear_image_words(K, NO) == 0) return (int)hipErrorInvalidValue;
  for (int z = 0; z < nprob; ++z)
    if (batch->ldw[z] != 0 && batch->ldw[z] != (transB ? K : NO)) return (int)hipErrorInvalidValue;  // dense weights only
  const int KS = K / 16, nitems = ((NO + 31) / 32) * KS * 64;
  dim3 grid((nitems + 255) / 256, nprob);
  if (KS == 8) hipLaunchKernelGGL((k_linear_prepare<8>), grid, dim3(256), 0, stream, *batch, NO, transB);
  else if (KS == 4) hipLaunchKernelGGL((k_linear_prepare<4>), grid, dim3(256), 0, stream, *batch, NO, transB);
  else hipLaunchKernelGGL((k_linear_prepare<2>), grid, dim3(256), 0, stream, *batch, NO, transB);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_linear_prepared(const float* X, int ldx, const uint32_t* image, const float* bias,
                                      const float* res, const float* tprev, float* Y, int ldy, int64_t R, int K,
                                      int NO, int flags, hipStream_t stream) {
  if (R <= 0) return 0;
  if (image == nullptr || geossl_linear_image_words(K, NO) == 0) return (int)hipErrorInvalidValue;
  if (ldx < K || ldy < NO || (ldx & 3) || (ldy & 3)) return (int)hipErrorInvalidValue;
  return launch_linear_split(X, ldx, reinterpret_cast<const float*>(image), reinterpret_cast<const u32x4*>(image), bias,
                             res, tprev, Y, ldy, R, K, NO, 1, flags, stream);
}

extern "C" int geossl_linear(const float* X, int ldx, const float* W, const float* bias, const float* res,
                             const float* tprev, float* Y, int ldy, int64_t R, int K, int NO, int transB, int flags,
                             hipStream_t stream) {
  if (R <= 0) return 0;
  if (K % 8 != 0 || K > 256 || NO > 256) return (int)hipErrorInvalidValue;
  if (ldx < K || ldy < NO || (ldx & 3) || (ldy & 3) || (NO & 3)) return (int)hipErrorInvalidValue;
  if (K == 32 || K == 64 || K == 128)
    return launch_linear_split(X, ldx, W, nullptr, bias, res, tprev, Y, ldy, R, K, NO, transB, flags, stream);
  const int ntiles = (int)((R + 127) / 128);
  const int NOp = (NO + 31) / 32 * 32;
  // split the output columns over gridDim.y when there are too few row tiles to fill 256 CUs
  int NC = (NOp % 128 == 0) ? 4 : ((NOp % 64 == 0) ? 2 : 1);
  if (NC == 4 && ntiles < 512) NC = 2;
  const int ny = NOp / (32 * NC);
  dim3 grid(ntiles < 1024 ? ntiles : 1024, ny);
  const size_t lds = ((size_t)((K * (32 * NC + 1) + 3) & ~3) + 4 * 512) * sizeof(float);
#define LAUNCH(NCV)                                                                                               \
  do {                                                                                                            \
    if (lds > 64 * 1024) allow_big_lds(&k_linear<NCV>);                                                           \
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
// several reductions in one launch (tn.h: ReduceMulti); same arithmetic and order as k_reduce_partials
__global__ __launch_bounds__(256) void k_reduce_multi(ReduceMulti m, int nblk, int accumulate) {
  __shared__ float red[4][64];
  reduce_multi_block(m, nblk, accumulate, red, (int)blockIdx.y);
}
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
  if (i < len) s = kahan_sum_strided(p + i, b0, b1, len);
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
  // Towards 256 rows per chunk while the launch keeps a block per CU: every chunk costs a block start-up and a 64 KB
  // partial (written, then read by the reduction), which at the reference's batch size (4 608 atom rows, 18 problems)
  // outweighed the rows themselves - 648 chunks of 128 rows against 324 of 256: 0.623 -> 0.608 ms per step (round 6).
  // A launch of ONE problem over the same rows (the tape's weight gradients in a train-on-forces step) keeps its 72
  // chunks of 64 rows: with 18 chunks of 256 it ran on 18 CUs, 25 us instead of 13.
  while (c < 256 && (int64_t)nprob * ((R + 2 * c - 1) / (2 * c)) >= 256) c *= 2;
  *chunk = (int)c;
  *nblk = (int)((R + c - 1) / c);
  if (*nblk < 1) *nblk = 1;
}

extern "C" int64_t geossl_tn_workspace_floats(int64_t R, int M, int N, int nprob) {
  return tn_workspace_floats(R, M, N, nprob);
}

// Column GEMM (weight gradient) of plain row-major operands: k_wgrad_split<., ., PlainOps> (wgrad.h).
extern "C" int geossl_linear_wgrad(const GeosslTnBatch* batch, int nprob, int64_t R, int M, int N, int lda, int ldb,
                                   int ldw, float* workspace, int accumulate, hipStream_t stream) {
  return geossl_linear_wgrad_dyn(batch, nprob, R, M, N, lda, ldb, ldw, workspace, accumulate, nullptr, stream);
}

extern "C" int geossl_linear_wgrad_dyn(const GeosslTnBatch* batch, int nprob, int64_t R, int M, int N, int lda, int ldb,
                                       int ldw, float* workspace, int accumulate, const int32_t* dyn_R,
                                       hipStream_t stream) {
  if (lda < M || ldb < N || ldw < N || (lda & 3) || (ldb & 3) || (M & 3) || (N & 3)) return (int)hipErrorInvalidValue;
  if (nprob <= 0 || R <= 0) return 0;
  if (nprob > GEOSSL_TN_MAX) return (int)hipErrorInvalidValue;
  {
    const int NCM = (M + 31) / 32, NCN = (N + 31) / 32;
    PlainOps ops;
    ops.batch = *batch;
    ops.lda = lda;
    ops.ldb = ldb;
    WgradOut out;
    for (int z = 0; z < GEOSSL_TN_MAX; ++z) {
      out.dW[z] = batch->dW[z];
      out.db[z] = batch->db[z];
      out.dd[z] = nullptr;
    }
#define WG(a, b) if (NCM == a && NCN == b) return launch_wgrad_split<a, b>(ops, nprob, R, M, N, out, ldw, 1, workspace, accumulate, stream, dyn_R)
    WG(4, 4); WG(4, 2); WG(2, 4); WG(2, 2); WG(1, 1); WG(1, 2); WG(2, 1); WG(4, 1); WG(1, 4);
#undef WG
  }
  return (int)hipErrorInvalidValue;  // M, N <= 128 in blocks of 32, 64 or 128 columns
}

extern "C" int geossl_abi_version(void) { return GEOSSL_ABI_VERSION; }
