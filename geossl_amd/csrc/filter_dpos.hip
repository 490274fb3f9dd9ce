// Gradient of the loss with respect to the edge lengths through the continuous-filter convolution of every
// interaction block, and from there to the atom positions (the d/d pos of finetune_md17.py:46 through
// schnet.py:91-93,186-187,205-207).  First order only.
//
// Positions enter SchNet through the length d_p of each pair slot p = (i < j) alone (the radius graph is discrete):
//     Wf_l[p][c] = C(d_p) * O_l[p][c],   O_l = W2_l t + b2_l,   t = ssp(u),   u = W1_l rbf(d_p) + b1_l.
// With the upstream gradient of the filter rows (never stored; rebuilt from the per-layer atom tensors, as in
// filter_bwd.hip)
//     dO_l[p][c] = C(d_p) * ( flag0 * dagg_l[i][c] * x_l[j][c]  +  flag1 * dagg_l[j][c] * x_l[i][c] ),
// the two paths into d_p are the envelope and the Gaussian smearing:
//     dL/dd_p = sum_l [ C'(d_p) / C(d_p)^2 * sum_c dO_l[p][c] Wf_l[p][c]
//                       + sum_h dU_l[p][h] * v_l[p][h] ],
//     dU = (W2^T dO) * ssp'(u),      v = W1 rbf'(d_p),     rbf'_g(d) = 2 coeff (d - mu_g) rbf_g(d).
//
// Same organisation as the filter forward (filter_fwd.hip): fp32 results on the bf16 matrix pipe (split.h), both
// products evaluated TRANSPOSED with the pair rows on the lanes and the weights (W2^T, W1) as pre-split A fragments
// in LDS; a wave owns 32 pair rows of one layer end to end.  dO^T is built directly in B-fragment layout from
// 16-byte gathers of the atom rows (L2 resident); W2^T dO^T and W1 rbf'^T land in the same C layout (lane = pair
// row, register = hidden unit), so dU * v is a register-wise product followed by one cross-half shuffle.
// Output: dd[l][p], summed over l and scattered to the atoms by k_pair_position_grad (fixed order, no atomics).
#include "common.h"
#include "geossl_hip.h"
#include "split.h"

using namespace geossl;

namespace {

template <int NMB, int K1S>
__global__ __launch_bounds__(512) void k_filter_dpos(const float* __restrict__ pair_d,
                                                     const float* __restrict__ pair_c,
                                                     const uint8_t* __restrict__ pair_flag,
                                                     const int32_t* __restrict__ pair_i,
                                                     const int32_t* __restrict__ pair_j, int P,
                                                     GeosslFilterWeights w, GeosslFilterGradIn g, int G,
                                                     const float* __restrict__ offset, float coeff, float cutoff,
                                                     const float* __restrict__ T, const float* __restrict__ Wf,
                                                     float* __restrict__ dd) {
  constexpr int F = 32 * NMB, K2S = F / 16;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  u32x4* W2Tf = reinterpret_cast<u32x4*>(smem_raw);         // [NMB][K2S][3][64] A fragments of W2^T (rows = h, k = c)
  u32x4* W1f = W2Tf + NMB * K2S * 3 * 64;                   // [NMB][K1S][3][64] A fragments of W1   (rows = h, k = g)
  float* offs = reinterpret_cast<float*>(W1f + NMB * K1S * 3 * 64);  // [16*K1S] Gaussian centres, zero padded
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int l = blockIdx.y;
  {
    const float* __restrict__ w2 = w.w2[l];
    for (int i = tid; i < NMB * K2S * 64; i += 512) {
      const int ln = i & 63, ks = (i >> 6) % K2S, mb = i / (64 * K2S);
      // A[m = h][k = c], c = 16ks + kperm(e, half): the contraction index of this GEMM is the filter channel
      const float* col = w2 + 32 * mb + (ln & 31);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = col[(size_t)(16 * ks + kperm(e, ln >> 5)) * F];
      const Frag3 f = split8(v);
      u32x4* dst = W2Tf + ((size_t)(mb * K2S + ks) * 3) * 64 + ln;
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
        const int gg = 16 * ks + 8 * (ln >> 5) + e;
        v[e] = gg < G ? row[gg] : 0.0f;
      }
      const Frag3 f = split8(v);
      u32x4* dst = W1f + ((size_t)(mb * K1S + ks) * 3) * 64 + ln;
      dst[0] = f.h;
      dst[64] = f.m;
      dst[128] = f.l;
    }
    for (int i = tid; i < 16 * K1S; i += 512) offs[i] = i < G ? offset[i] : 0.0f;
  }
  __syncthreads();
  const float* __restrict__ x = g.x[l];
  const float* __restrict__ dagg = g.dagg[l];
  const size_t lbase = (size_t)l * P;
  const int nrb = (P + 31) / 32;
  for (int rb = blockIdx.x * 8 + wave; rb < nrb; rb += gridDim.x * 8) {
    const int row = 32 * rb + j;
    const bool live = row < P;
    const int rc = live ? row : P - 1;  // clamped: every address below is valid, dead rows are masked by m0 = m1 = 0
    const float d = pair_d[rc];
    const float cw = pair_c[rc];
    const unsigned fl = live ? pair_flag[rc] : 0u;
    const float m0 = (fl & 1u) ? cw : 0.0f, m1 = (fl & 2u) ? cw : 0.0f;
    const float* __restrict__ di = dagg + (size_t)pair_i[rc] * F + 4 * kh;
    const float* __restrict__ dj = dagg + (size_t)pair_j[rc] * F + 4 * kh;
    const float* __restrict__ xi = x + (size_t)pair_i[rc] * F + 4 * kh;
    const float* __restrict__ xj = x + (size_t)pair_j[rc] * F + 4 * kh;
    const float* __restrict__ wrow = Wf + (lbase + rc) * F + 4 * kh;
    // dO^T as B fragments of the contraction over the filter channels (element e of k-step ks <-> channel
    // 16ks + kperm(e, kh): two 16-byte pieces per operand), and the envelope path sum_c dO * Wf
    Frag3 dof[K2S];
    float s1 = 0.0f;
#pragma unroll
    for (int ks = 0; ks < K2S; ++ks) {
      float v[8];
#pragma unroll
      for (int h4 = 0; h4 < 2; ++h4) {
        const int c = 16 * ks + 8 * h4;
        const float4 a = *reinterpret_cast<const float4*>(di + c), b = *reinterpret_cast<const float4*>(xj + c);
        const float4 cc = *reinterpret_cast<const float4*>(dj + c), e4 = *reinterpret_cast<const float4*>(xi + c);
        const float4 wf = *reinterpret_cast<const float4*>(wrow + c);
        v[4 * h4 + 0] = m0 * (a.x * b.x) + m1 * (cc.x * e4.x);
        v[4 * h4 + 1] = m0 * (a.y * b.y) + m1 * (cc.y * e4.y);
        v[4 * h4 + 2] = m0 * (a.z * b.z) + m1 * (cc.z * e4.z);
        v[4 * h4 + 3] = m0 * (a.w * b.w) + m1 * (cc.w * e4.w);
        s1 += (v[4 * h4] * wf.x + v[4 * h4 + 1] * wf.y) + (v[4 * h4 + 2] * wf.z + v[4 * h4 + 3] * wf.w);
      }
      dof[ks] = split8(v);
    }
    // rbf'^T B fragments: lane (row, half kh) differentiates the 8 Gaussians of its k-step for its own row
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
        v[e] = (2.0f * coeff * diff) * exp_neg(coeff * (diff * diff));  // d/dd of schnet.py:206-207
      }
      bfr[ks] = split8(v);
    }
    // per 32-wide block of hidden units: dt^T = W2^T dO^T, v^T = W1 rbf'^T (same C layout), dU = dt * ssp'(u)
    const float* __restrict__ trow = T + (lbase + rc) * F + 4 * kh;
    float s2 = 0.0f;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      float4 tq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) tq[q] = *reinterpret_cast<const float4*>(trow + 32 * mb + 8 * q);
      f32x16 at0, at1, av;
#pragma unroll
      for (int r = 0; r < 16; ++r) at0[r] = at1[r] = av[r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < K1S; ++ks) {
        Frag3 af;
        const u32x4* src = W1f + ((size_t)(mb * K1S + ks) * 3) * 64 + lane;
        af.h = src[0];
        af.m = src[64];
        af.l = src[128];
        mma6(av, af, bfr[ks]);
      }
#pragma unroll
      for (int ks = 0; ks < K2S; ++ks) {
        Frag3 af;
        const u32x4* src = W2Tf + ((size_t)(mb * K2S + ks) * 3) * 64 + lane;
        af.h = src[0];
        af.m = src[64];
        af.l = src[128];
        mma6x2(at0, at1, af, dof[ks]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float t4[4] = {tq[q].x, tq[q].y, tq[q].z, tq[q].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q + i;
          s2 += ((at0[r] + at1[r]) * dssp_from_out(t4[i])) * av[r];
        }
      }
    }
    // envelope: C(d) = (cos(pi d / r_c) + 1) / 2  (schnet.py:186);  sum_c dO Wf = C^2 sum_c dWf O
    const float cp = -0.5f * (GEOSSL_PI_F / cutoff) * sinf(d * GEOSSL_PI_F / cutoff);
    float tot = s2 + (cw > 0.0f ? cp * (s1 / (cw * cw)) : 0.0f);
    tot += __shfl_xor(tot, 32, 64);
    if (live && kh == 0) dd[lbase + row] = tot;
  }
}

// dpos[a] = sum over the other atoms b of the molecule of (sum_l dd[l][slot(a,b)]) * (pos_a - pos_b) / d(a,b)
// (the derivative of (pos[row] - pos[col]).norm(), schnet.py:93, for both directions of the pair at once); slots
// that are not edges carry dd = 0.  One thread per atom, fixed summation order.
__global__ void k_pair_position_grad(const float* __restrict__ pos, const float* __restrict__ pair_d,
                                     const float* __restrict__ dd, const int32_t* __restrict__ mol_ptr,
                                     const int32_t* __restrict__ pair_ptr, int B, int64_t P, int L,
                                     float* __restrict__ dpos) {
  const int m = blockIdx.x;
  if (m >= B) return;
  const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0, base = pair_ptr[m];
  for (int a = threadIdx.x; a < n; a += blockDim.x) {
    const float px = pos[3 * (size_t)(a0 + a)], py = pos[3 * (size_t)(a0 + a) + 1], pz = pos[3 * (size_t)(a0 + a) + 2];
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;
    for (int b = 0; b < n; ++b) {
      if (b == a) continue;
      const int lo = a < b ? a : b, hi = a < b ? b : a;
      const int slot = base + lo * n - lo * (lo + 1) / 2 - lo - 1 + hi;
      float s = 0.0f;
      for (int l = 0; l < L; ++l) s += dd[(size_t)l * P + slot];
      const float dist = pair_d[slot];
      if (s != 0.0f && dist > 0.0f) {
        const float k = s / dist;
        gx += k * (px - pos[3 * (size_t)(a0 + b)]);
        gy += k * (py - pos[3 * (size_t)(a0 + b) + 1]);
        gz += k * (pz - pos[3 * (size_t)(a0 + b) + 2]);
      }
    }
    dpos[3 * (size_t)(a0 + a)] = gx;
    dpos[3 * (size_t)(a0 + a) + 1] = gy;
    dpos[3 * (size_t)(a0 + a) + 2] = gz;
  }
}

}  // namespace

extern "C" int geossl_cfconv_filter_dpos(const float* pair_d, const float* pair_c, const uint8_t* pair_flag,
                                         const int32_t* pair_i, const int32_t* pair_j, int64_t P,
                                         const GeosslFilterWeights* w, const GeosslFilterGradIn* g, int L, int F,
                                         int G, const float* offset, float coeff, float cutoff, const float* T,
                                         const float* Wf, float* dd, hipStream_t stream) {
  if (P <= 0 || L <= 0) return 0;
  if (L > GEOSSL_MAX_L || (F != 32 && F != 64 && F != 128) || G > 64 || G < 1 || T == nullptr || Wf == nullptr)
    return (int)hipErrorInvalidValue;
  const int nrb = (int)((P + 31) / 32);
  int per_layer = 256 / L;  // one 8-wave block per CU, a layer per block (its weights stay in LDS)
  if (per_layer < 1) per_layer = 1;
  if (per_layer > (nrb + 7) / 8) per_layer = (nrb + 7) / 8;
  dim3 grid(per_layer, L);
#define LAUNCH(NMB, K1S)                                                                                          \
  do {                                                                                                            \
    const size_t lds = (size_t)(NMB * (2 * NMB) + NMB * K1S) * 3 * 1024 + (16 * K1S) * 4;                         \
    allow_big_lds(&k_filter_dpos<NMB, K1S>);                                                                      \
    hipLaunchKernelGGL((k_filter_dpos<NMB, K1S>), grid, dim3(512), lds, stream, pair_d, pair_c, pair_flag, pair_i, \
                       pair_j, (int)P, *w, *g, G, offset, coeff, cutoff, T, Wf, dd);                              \
  } while (0)
#define LAUNCH_F(NMB)                   \
  do {                                  \
    if (G <= 16) LAUNCH(NMB, 1);        \
    else if (G <= 32) LAUNCH(NMB, 2);   \
    else if (G <= 48) LAUNCH(NMB, 3);   \
    else LAUNCH(NMB, 4);                \
  } while (0)
  if (F == 128) LAUNCH_F(4); else if (F == 64) LAUNCH_F(2); else LAUNCH_F(1);
#undef LAUNCH_F
#undef LAUNCH
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_pair_position_grad(const float* pos, const float* pair_d, const float* dd,
                                         const int32_t* mol_ptr, const int32_t* pair_ptr, int64_t B, int64_t P, int L,
                                         float* dpos, hipStream_t stream) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(k_pair_position_grad, dim3((unsigned)B), dim3(64), 0, stream, pos, pair_d, dd, mol_ptr, pair_ptr,
                     (int)B, P, L, dpos);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
