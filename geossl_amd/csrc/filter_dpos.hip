// Gradient of the loss with respect to the edge lengths through the continuous-filter convolution of every
// interaction block, and from there to the atom positions (the d/d pos of finetune_md17.py:46 through
// schnet.py:91-93,186-187,205-207).  First order only.
//
// Positions enter SchNet through the length d_p of each pair slot p = (i < j) alone (the radius graph is discrete):
//     Wf_l[p][c] = C(d_p) * O_l[p][c],   O_l = W2_l t + b2_l,   t = ssp(u),   u = W1_l rbf(d_p) + b1_l.
// The derivative of a filter row along its one input is propagated FORWARD through the filter network (a
// Jacobian-vector product with the scalar tangent d), which keeps every product in the orientation the forward
// kernel uses and needs no transposition of gathered data:
//     rbf'_g(d) = 2 coeff (d - mu_g) rbf_g(d),   z = W2 ( ssp'(u) * (W1 rbf'(d)) ) = dO/dd,
//     J[p][c]  = dWf[p][c]/dd = C(d) z[c] + C'(d)/C(d) * Wf[p][c],
// and contracted with the upstream gradient of the row (never stored; rebuilt from the per-layer atom tensors as in
// filter_bwd.hip):
//     dL/dd_p = sum_l sum_c ( flag0 * dagg_l[i][c] * x_l[j][c] + flag1 * dagg_l[j][c] * x_l[i][c] ) * J_l[p][c].
//
// Same organisation as the filter forward (filter_fwd.hip): fp32 results on the bf16 matrix pipe (split.h), both
// products evaluated TRANSPOSED with the pair rows on the lanes and the weights as pre-split A fragments in LDS (the
// very fragments of the forward); a wave owns 32 pair rows of one layer end to end; ssp'(u) comes from the saved
// activations T.  z lands in C layout (lane = pair row, register = 4 consecutive channels per group), which is also
// the layout of 16-byte gathers of the atom rows (L2 resident) and of the filter row, so the contraction is
// register-wise followed by one cross-half shuffle.
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
  u32x4* W2f = reinterpret_cast<u32x4*>(smem_raw);          // [NMB][K2S][3][64] A fragments of W2 (rows = c, k = h)
  u32x4* W1f = W2f + NMB * K2S * 3 * 64;                    // [NMB][K1S][3][64] A fragments of W1 (rows = h, k = g)
  float* offs = reinterpret_cast<float*>(W1f + NMB * K1S * 3 * 64);  // [16*K1S] Gaussian centres, zero padded
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int l = blockIdx.y;
  {
    const float* __restrict__ w2 = w.w2[l];
    for (int i = tid; i < NMB * K2S * 64; i += 512) {
      const int ln = i & 63, ks = (i >> 6) % K2S, mb = i / (64 * K2S);
      // contraction-index permutation kperm (split.h): elements 0..3 <- hidden units 4kh.., 4..7 <- 8+4kh..
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
  // row blocks dealt wave-index-major: the waves that get one block more than the others are then spread one per SIMD
  // over all blocks instead of filling the first blocks (a SIMD's two waves share its matrix pipe)
  for (int rb = blockIdx.x + gridDim.x * wave; rb < nrb; rb += gridDim.x * 8) {
    const int row = 32 * rb + j;
    const bool live = row < P;
    const int rc = live ? row : P - 1;  // clamped: every address below is valid, dead rows are masked by f0 = f1 = 0
    const float d = pair_d[rc];
    const float cw = pair_c[rc];
    const unsigned fl = live ? pair_flag[rc] : 0u;
    const float* __restrict__ di = dagg + (size_t)pair_i[rc] * F + 4 * kh;
    const float* __restrict__ dj = dagg + (size_t)pair_j[rc] * F + 4 * kh;
    const float* __restrict__ xi = x + (size_t)pair_i[rc] * F + 4 * kh;
    const float* __restrict__ xj = x + (size_t)pair_j[rc] * F + 4 * kh;
    const float* __restrict__ wrow = Wf + (lbase + rc) * F + 4 * kh;
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
    // hidden block mb: du/dd = W1 rbf' (C layout: lane = pair row, register = hidden unit), times ssp'(u) from the
    // saved activations, split in registers into the two B fragments (k-steps 2mb, 2mb+1) of the second product,
    // which are consumed at once by the z accumulators of ALL output blocks - nothing of the block stays live
    const float* __restrict__ trow = T + (lbase + rc) * F + 4 * kh;
    f32x16 z[NMB];
#pragma unroll
    for (int u = 0; u < NMB; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) z[u][r] = 0.0f;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      float4 tq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) tq[q] = *reinterpret_cast<const float4*>(trow + 32 * mb + 8 * q);
      f32x16 a1;
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < K1S; ++ks) {
        Frag3 af;
        const u32x4* src = W1f + ((size_t)(mb * K1S + ks) * 3) * 64 + lane;
        af.h = src[0];
        af.m = src[64];
        af.l = src[128];
        mma6(a1, af, bfr[ks]);
      }
      Frag3 tb[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const float4 t0 = tq[2 * half], t1 = tq[2 * half + 1];
        const float v[8] = {a1[8 * half] * dssp_from_out(t0.x),     a1[8 * half + 1] * dssp_from_out(t0.y),
                            a1[8 * half + 2] * dssp_from_out(t0.z), a1[8 * half + 3] * dssp_from_out(t0.w),
                            a1[8 * half + 4] * dssp_from_out(t1.x), a1[8 * half + 5] * dssp_from_out(t1.y),
                            a1[8 * half + 6] * dssp_from_out(t1.z), a1[8 * half + 7] * dssp_from_out(t1.w)};
        tb[half] = split8(v);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int u = 0; u < NMB; ++u) {
          Frag3 af;
          const u32x4* src = W2f + ((size_t)(u * K2S + 2 * mb + half) * 3) * 64 + lane;
          af.h = src[0];
          af.m = src[64];
          af.l = src[128];
          mma6(z[u], af, tb[half]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // contraction with the upstream gradient of the filter row, one 32-channel block of gathers in flight at a time.
    // envelope: C(d) = (cos(pi d / r_c) + 1) / 2  (schnet.py:186);  Wf = C * O
    const float cp = -0.5f * (GEOSSL_PI_F / cutoff) * sinf(d * GEOSSL_PI_F / cutoff);
    const float kappa = cw > 0.0f ? cp / cw : 0.0f;
    const float f0 = (fl & 1u) ? 1.0f : 0.0f, f1 = (fl & 2u) ? 1.0f : 0.0f;
    float tot = 0.0f;
#pragma unroll
    for (int u = 0; u < NMB; ++u) {
      float4 a[4], b[4], cc[4], e4[4], wf[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 32 * u + 8 * q;
        a[q] = *reinterpret_cast<const float4*>(di + c);
        b[q] = *reinterpret_cast<const float4*>(xj + c);
        cc[q] = *reinterpret_cast<const float4*>(dj + c);
        e4[q] = *reinterpret_cast<const float4*>(xi + c);
        wf[q] = *reinterpret_cast<const float4*>(wrow + c);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float gx = f0 * (a[q].x * b[q].x) + f1 * (cc[q].x * e4[q].x);
        const float gy = f0 * (a[q].y * b[q].y) + f1 * (cc[q].y * e4[q].y);
        const float gz = f0 * (a[q].z * b[q].z) + f1 * (cc[q].z * e4[q].z);
        const float gw = f0 * (a[q].w * b[q].w) + f1 * (cc[q].w * e4[q].w);
        tot += (gx * (cw * z[u][4 * q] + kappa * wf[q].x) + gy * (cw * z[u][4 * q + 1] + kappa * wf[q].y)) +
               (gz * (cw * z[u][4 * q + 2] + kappa * wf[q].z) + gw * (cw * z[u][4 * q + 3] + kappa * wf[q].w));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
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
