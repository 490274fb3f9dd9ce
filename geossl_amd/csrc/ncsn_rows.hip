// Row passes of the denoising-distance-matching head NCSN_version_03 (NCSN.py:168-220) on the bf16 matrix pipe
// (split.h): forward loss per super-edge and the backward pass down to the gradient at h_u + h_v.
//
// Per super-edge row s = (u, v):   feat = h[u] + h[v]                                   (NCSN.py:201-203)
//     a1 = relu(o1_w [feat, emb] + o1_b),  a2 = relu(o2_w a1 + o2_b),  out = o3_w a2 + o3_b     (:204, MLP :9-31)
// with emb the 1 -> F -> 1 distance embedding of the perturbed distance (:196-197).
//
// Like the filter network (filter_fwd.hip) both dense layers are evaluated TRANSPOSED: weights are the A operand
// (pre-split, pre-permuted fragments in LDS, 144 KB at F = 128), the rows sit on the lanes, a wave owns 32 rows end
// to end with no block-level synchronisation, and the C layout of one product is the B-fragment layout of the
// next, so relu / masks / splitting happen in registers.  The backward row pass mirrors it with the transposed
// weights:  dz2 = g w3 [a2 > 0];  dz1 = (o2_w^T dz2) [a1 > 0];  dfeat = o1_w[:, :F]^T dz1;  demb = o1_w[:, F] . dz1.
#include "common.h"
#include "geossl_hip.h"
#include "split.h"

using namespace geossl;

namespace {

struct RowIn {
  int64_t u, v;
  float sigma, d, eps;
};

// NMB = F/32
template <int NMB>
__device__ __forceinline__ void ncsn_fwd_body(const float* __restrict__ h, const int64_t* __restrict__ batch,
                                              const int64_t* __restrict__ sei0, const int64_t* __restrict__ sei1,
                                              int S, const float* __restrict__ distance,
                                              const int64_t* __restrict__ noise_level,
                                              const float* __restrict__ dist_noise, const GeosslNcsnWeights& w,
                                              float anneal_power, float* __restrict__ loss_e,
                                              const GeosslNcsnSaved& sv, float* __restrict__ loss_part) {
  constexpr int F = 32 * NMB, H = F / 2, HMB = (H + 31) / 32, HP = 32 * HMB, KS = F / 16;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  u32x4* W1f = reinterpret_cast<u32x4*>(smem_raw);  // [NMB][KS][3][64]: A[m = n][k], k = 16ks + 8kh + e
  u32x4* W2f = W1f + NMB * KS * 3 * 64;             // [HMB][KS][3][64]: A[m][n], n = 16ks + kperm(e, kh)
  float* b1s = reinterpret_cast<float*>(W2f + HMB * KS * 3 * 64);  // [F]
  float* wls = b1s + F;    // [F]  last column of o1_w (multiplies the distance embedding)
  float* b2s = wls + F;    // [HP]
  float* w3s = b2s + HP;   // [HP]
  float* iw1 = w3s + HP;   // [F]
  float* ib1 = iw1 + F;    // [F]
  float* iw2 = ib1 + F;    // [F]
  float* lred = iw2 + F;   // [8]  loss sums of the block's waves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int nrb = (S + 31) / 32;
  float lsum = 0.0f;       // loss of this lane's rows (lanes of the lower half-wave)
  auto load_row = [&](int rb) {
    RowIn r;
    const int rowc = min(32 * rb + j, S - 1);
    r.u = sei0[rowc];
    r.v = sei1[rowc];
    r.sigma = w.sigmas[noise_level[batch[r.u]]];  // NCSN.py:187,191-192
    r.d = distance[rowc];
    r.eps = dist_noise[rowc];
    return r;
  };
  // Row blocks are dealt wave-index-major (all waves 0 of the grid first, then all waves 1, ...): when the row blocks
  // do not divide evenly (4896 on 2048 waves at the bench size), the waves with one block more are then waves 0-3 of
  // every block - one per SIMD - instead of all eight waves of the first blocks, whose SIMDs would carry 6 blocks
  // against 4 elsewhere (a SIMD's two waves share its matrix pipe).
  int rb = blockIdx.x + gridDim.x * wave;
  RowIn cur;
  if (rb < nrb) cur = load_row(rb);  // in flight while the weights are formatted
  // ---- one-time weight formatting
#ifdef NCSN_STAGE_UNROLL
#pragma unroll
#endif
  for (int i = tid; i < NMB * KS * 64; i += 512) {
    const int ln = i & 63, ks = (i >> 6) % KS, mb = i / (64 * KS);
    const float* row = w.o1_w + (size_t)(32 * mb + (ln & 31)) * (F + 1) + 16 * ks + 8 * (ln >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = row[e];
    const Frag3 f = split8(v);
    u32x4* dst = W1f + ((size_t)(mb * KS + ks) * 3) * 64 + ln;
    dst[0] = f.h;
    dst[64] = f.m;
    dst[128] = f.l;
  }
#ifdef NCSN_STAGE_UNROLL
#pragma unroll
#endif
  for (int i = tid; i < HMB * KS * 64; i += 512) {
    const int ln = i & 63, ks = (i >> 6) % KS, mb = i / (64 * KS);
    const int m = 32 * mb + (ln & 31);
    const float* row = w.o2_w + (size_t)min(m, H - 1) * F + 16 * ks + 4 * (ln >> 5);
    const f32x4 lo = *reinterpret_cast<const f32x4*>(row), hi = *reinterpret_cast<const f32x4*>(row + 8);
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if (m >= H) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.0f;
    }
    const Frag3 f = split8(v);
    u32x4* dst = W2f + ((size_t)(mb * KS + ks) * 3) * 64 + ln;
    dst[0] = f.h;
    dst[64] = f.m;
    dst[128] = f.l;
  }
  for (int i = tid; i < F; i += 512) {
    b1s[i] = w.o1_b[i];
    wls[i] = w.o1_w[(size_t)i * (F + 1) + F];
    iw1[i] = w.in_w1[i];
    ib1[i] = w.in_b1[i];
    iw2[i] = w.in_w2[i];
  }
  for (int i = tid; i < HP; i += 512) {
    b2s[i] = i < H ? w.o2_b[i] : 0.0f;
    w3s[i] = i < H ? w.o3_w[i] : 0.0f;
  }
  __syncthreads();
  const float ib2 = w.in_b2[0], b3 = w.o3_b[0];
  for (; rb < nrb; rb += gridDim.x * 8) {
    const int row = 32 * rb + j;
    const bool valid = row < S;
    const RowIn in = cur;
    // feature gather: h[u] + h[v], 8 consecutive features per lane and k-step; two k-steps of requests in flight
    f32x4 gq[2][4];
    auto gather = [&](int ks, f32x4 (&g)[4]) {
      const float* hu = h + (size_t)in.u * F + 16 * ks + 8 * kh;
      const float* hv = h + (size_t)in.v * F + 16 * ks + 8 * kh;
      g[0] = *reinterpret_cast<const f32x4*>(hu);
      g[1] = *reinterpret_cast<const f32x4*>(hu + 4);
      g[2] = *reinterpret_cast<const f32x4*>(hv);
      g[3] = *reinterpret_cast<const f32x4*>(hv + 4);
    };
    gather(0, gq[0]);
    if (rb + gridDim.x * 8 < nrb) cur = load_row(rb + gridDim.x * 8);  // next row block's scalars
    const float pd = add_rn(in.d, mul_rn(in.eps, in.sigma));  // :196
    // distance embedding (:197): MLP 1 -> F -> 1 with relu; each half-wave sums half of the hidden units
    float e = 0.0f;
#pragma unroll 4
    for (int k = kh * (F / 2); k < (kh + 1) * (F / 2); k += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(iw1 + k), b = *reinterpret_cast<const f32x4*>(ib1 + k);
      const f32x4 c = *reinterpret_cast<const f32x4*>(iw2 + k);
      e = fmaf(c.x, fmaxf(fmaf(a.x, pd, b.x), 0.0f), e);
      e = fmaf(c.y, fmaxf(fmaf(a.y, pd, b.y), 0.0f), e);
      e = fmaf(c.z, fmaxf(fmaf(a.z, pd, b.z), 0.0f), e);
      e = fmaf(c.w, fmaxf(fmaf(a.w, pd, b.w), 0.0f), e);
    }
    e += __shfl_xor(e, 32, 64);
    const float emb = e + ib2;
    // first layer, transposed: acc1[mb] = (o1_w [feat, emb] + o1_b)^T
    f32x16 acc1[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(b1s + 32 * mb + 8 * q + 4 * kh);
        const f32x4 wl = *reinterpret_cast<const f32x4*>(wls + 32 * mb + 8 * q + 4 * kh);
        acc1[mb][4 * q] = fmaf(emb, wl.x, b.x);
        acc1[mb][4 * q + 1] = fmaf(emb, wl.y, b.y);
        acc1[mb][4 * q + 2] = fmaf(emb, wl.z, b.z);
        acc1[mb][4 * q + 3] = fmaf(emb, wl.w, b.w);
      }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) gather(ks + 1, gq[(ks + 1) & 1]);
      const f32x4(&g)[4] = gq[ks & 1];
      const float v[8] = {g[0].x + g[2].x, g[0].y + g[2].y, g[0].z + g[2].z, g[0].w + g[2].w,
                          g[1].x + g[3].x, g[1].y + g[3].y, g[1].z + g[3].z, g[1].w + g[3].w};
      const Frag3 bf = split8(v);
      Frag3 af[NMB];
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) {
        const u32x4* src = W1f + ((size_t)(mb * KS + ks) * 3) * 64 + lane;
        af[mb].h = src[0];
        af[mb].m = src[64];
        af[mb].l = src[128];
      }
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) acc1[mb] = mfma_bf16(af[mb].l, bf.h, acc1[mb]);
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) acc1[mb] = mfma_bf16(af[mb].h, bf.l, acc1[mb]);
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) acc1[mb] = mfma_bf16(af[mb].m, bf.m, acc1[mb]);
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) acc1[mb] = mfma_bf16(af[mb].m, bf.h, acc1[mb]);
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) acc1[mb] = mfma_bf16(af[mb].h, bf.m, acc1[mb]);
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) acc1[mb] = mfma_bf16(af[mb].h, bf.h, acc1[mb]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // relu; a1 saved for the backward; split into the B fragments of the second layer
    Frag3 tb[KS];
    {
      float* arow = sv.a1 != nullptr ? sv.a1 + (size_t)row * F + 4 * kh : nullptr;
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          float v[8];
#pragma unroll
          for (int e2 = 0; e2 < 8; ++e2) v[e2] = fmaxf(acc1[mb][8 * half + e2], 0.0f);
          if (arow != nullptr && valid) {
            *reinterpret_cast<f32x4*>(arow + 32 * mb + 16 * half) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(arow + 32 * mb + 16 * half + 8) = f32x4{v[4], v[5], v[6], v[7]};
          }
          tb[2 * mb + half] = split8(v);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    // second layer, transposed; the third (H -> 1) is a dot over this lane's registers + the other half-wave
    float sc = 0.0f;
#pragma unroll
    for (int mb = 0; mb < HMB; ++mb) {
      f32x16 acc2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(b2s + 32 * mb + 8 * q + 4 * kh);
        acc2[4 * q] = b.x;
        acc2[4 * q + 1] = b.y;
        acc2[4 * q + 2] = b.z;
        acc2[4 * q + 3] = b.w;
      }
      Frag3 af, an;
      {
        const u32x4* src = W2f + ((size_t)(mb * KS) * 3) * 64 + lane;
        af.h = src[0]; af.m = src[64]; af.l = src[128];
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
          const u32x4* src = W2f + ((size_t)(mb * KS + ks + 1) * 3) * 64 + lane;
          an.h = src[0]; an.m = src[64]; an.l = src[128];
        }
        __builtin_amdgcn_sched_barrier(0);
        mma6(acc2, af, tb[ks]);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < KS) af = an;
      }
      float* a2row = sv.a2 != nullptr ? sv.a2 + (size_t)row * H + 32 * mb + 4 * kh : nullptr;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w3 = *reinterpret_cast<const f32x4*>(w3s + 32 * mb + 8 * q + 4 * kh);
        const f32x4 a = f32x4{fmaxf(acc2[4 * q], 0.0f), fmaxf(acc2[4 * q + 1], 0.0f), fmaxf(acc2[4 * q + 2], 0.0f),
                              fmaxf(acc2[4 * q + 3], 0.0f)};
        sc = fmaf(a.x, w3.x, sc);
        sc = fmaf(a.y, w3.y, sc);
        sc = fmaf(a.z, w3.z, sc);
        sc = fmaf(a.w, w3.w, sc);
        if (a2row != nullptr && valid && 32 * mb + 8 * q + 4 * kh < H) *reinterpret_cast<f32x4*>(a2row + 8 * q) = a;
      }
    }
    sc += __shfl_xor(sc, 32, 64);
    const float out = sc + b3;
    const float inv_sigma = 1.0f / in.sigma;
    const float score = out * inv_sigma;                                            // :205
    const float target = (-1.0f / (in.sigma * in.sigma)) * (pd - in.d);              // :199
    const float diff = score - target;
    const float pw = powf(in.sigma, anneal_power);
    if (valid && kh == 0) {
      const float le = (0.5f * (diff * diff)) * pw;  // :209
      loss_e[row] = le;
      lsum += le;
      if (sv.pd != nullptr) {
        sv.pd[row] = pd;
        sv.emb[row] = emb;
        sv.gscale[row] = diff * pw * inv_sigma;  // d loss_e / d out
      }
    }
  }
  // one loss partial per block (geossl_loss_reduce_partials sums them in block order): waves in order, fixed butterfly
  if (loss_part != nullptr) {
    lsum = wave_sum(lsum);
    if (lane == 0) lred[wave] = lsum;
    __syncthreads();
    if (tid == 0) {
      float t = 0.0f;
      for (int i = 0; i < 8; ++i) t += lred[i];
      loss_part[blockIdx.x] = t;
    }
    if (blockIdx.x == 0)
      for (int i = (int)gridDim.x + tid; i < GEOSSL_LOSS_PARTIALS; i += 512) loss_part[i] = 0.0f;
  }
}

template <int NMB>
__global__ __launch_bounds__(512) void k_ncsn_fwd(const float* __restrict__ h, const int64_t* __restrict__ batch,
                                                  const int64_t* __restrict__ sei0, const int64_t* __restrict__ sei1,
                                                  int S, const float* __restrict__ distance,
                                                  const int64_t* __restrict__ noise_level,
                                                  const float* __restrict__ dist_noise, GeosslNcsnWeights w,
                                                  float anneal_power, float* __restrict__ loss_e,
                                                  GeosslNcsnSaved sv, float* __restrict__ loss_part) {
  ncsn_fwd_body<NMB>(h, batch, sei0, sei1, S, distance, noise_level, dist_noise, w, anneal_power, loss_e, sv, loss_part);
}
// both heads of a DDM step in one launch: blockIdx.y = head (same super-edges, each head its own view, weights, noise)
struct NcsnFwdHead {
  const float* h;
  const float* distance;
  const int64_t* noise_level;
  const float* dist_noise;
  GeosslNcsnWeights w;
  GeosslNcsnSaved sv;
  float anneal_power;
  float* loss_e;
  float* loss_part;
};
struct NcsnFwdPair { NcsnFwdHead a[2]; };  // one kernel argument indexed by blockIdx.y (see k_ncsn_bwd_fused2)
template <int NMB>
__global__ __launch_bounds__(512) void k_ncsn_fwd2(NcsnFwdPair pr, const int64_t* __restrict__ batch,
                                                   const int64_t* __restrict__ sei0, const int64_t* __restrict__ sei1,
                                                   int S, const int32_t* __restrict__ dyn_S,
                                                   const int32_t* __restrict__ dyn_view) {
  const NcsnFwdHead& a = pr.a[blockIdx.y];
  // capacity launch: the real number of super-edges, and (dyn_view) both heads were handed the base of ONE
  // [view 0 ; view 1] feature tensor - head 1's rows start *dyn_view rows in
  const float* h = a.h;
  if (dyn_view != nullptr && blockIdx.y == 1) h += (size_t)(*dyn_view) * (32 * NMB);
  ncsn_fwd_body<NMB>(h, batch, sei0, sei1, dyn_count(S, dyn_S), a.distance, a.noise_level, a.dist_noise, a.w,
                     a.anneal_power, a.loss_e, a.sv, a.loss_part);
}

template <int NMB>
__global__ __launch_bounds__(512) void k_ncsn_bwd_rows(GeosslNcsnWeights w, GeosslNcsnSaved sv, int S,
                                                       const int64_t* __restrict__ divisor, float out_scale,
                                                       const float* __restrict__ gout, float* __restrict__ dz1,
                                                       float* __restrict__ dfeat, float* __restrict__ demb,
                                                       float* __restrict__ grow) {
  constexpr int F = 32 * NMB, H = F / 2, HP = 32 * ((H + 31) / 32), KS = F / 16, KH = H / 16;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  u32x4* W2t = reinterpret_cast<u32x4*>(smem_raw);  // [NMB][KH][3][64]: A[m = n][k = m'], m' = 16ks + 8kh + e
  u32x4* W1t = W2t + NMB * KH * 3 * 64;             // [NMB][KS][3][64]: A[m = k][k = n], n = 16ks + kperm(e, kh)
  float* wls = reinterpret_cast<float*>(W1t + NMB * KS * 3 * 64);  // [F]
  float* w3s = wls + F;                                            // [HP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  for (int i = tid; i < NMB * KH * 64; i += 512) {
    const int ln = i & 63, ks = (i >> 6) % KH, mb = i / (64 * KH);
    const float* col = w.o2_w + (size_t)(16 * ks + 8 * (ln >> 5)) * F + 32 * mb + (ln & 31);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = col[(size_t)e * F];
    const Frag3 f = split8(v);
    u32x4* dst = W2t + ((size_t)(mb * KH + ks) * 3) * 64 + ln;
    dst[0] = f.h;
    dst[64] = f.m;
    dst[128] = f.l;
  }
  for (int i = tid; i < NMB * KS * 64; i += 512) {
    const int ln = i & 63, ks = (i >> 6) % KS, mb = i / (64 * KS);
    const float* col = w.o1_w + 32 * mb + (ln & 31);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = col[(size_t)(16 * ks + kperm(e, ln >> 5)) * (F + 1)];
    const Frag3 f = split8(v);
    u32x4* dst = W1t + ((size_t)(mb * KS + ks) * 3) * 64 + ln;
    dst[0] = f.h;
    dst[64] = f.m;
    dst[128] = f.l;
  }
  for (int i = tid; i < F; i += 512) wls[i] = w.o1_w[(size_t)i * (F + 1) + F];
  for (int i = tid; i < HP; i += 512) w3s[i] = i < H ? w.o3_w[i] : 0.0f;
  __syncthreads();
  const float scale = out_scale * (gout != nullptr ? gout[0] : 1.0f) / (float)divisor[0];
  const int nrb = (S + 31) / 32;
  for (int rb = blockIdx.x + gridDim.x * wave; rb < nrb; rb += gridDim.x * 8) {  // wave-index-major, as in k_ncsn_fwd
    const int row = 32 * rb + j;
    const bool valid = row < S;
    const size_t rowc = (size_t)min(row, S - 1);
    const float gr = valid ? sv.gscale[rowc] * scale : 0.0f;  // d L / d out_row
    if (valid && kh == 0) grow[row] = gr;
    // dz2[m] = gr * w3[m] * [a2 > 0] as B fragments (lane = row, 8 consecutive m per k-step)
    Frag3 zb[KH];
    {
      const float* a2row = sv.a2 + rowc * H + 8 * kh;
      f32x4 a2v[KH][2];
#pragma unroll
      for (int ks = 0; ks < KH; ++ks) {
        a2v[ks][0] = *reinterpret_cast<const f32x4*>(a2row + 16 * ks);
        a2v[ks][1] = *reinterpret_cast<const f32x4*>(a2row + 16 * ks + 4);
      }
#pragma unroll
      for (int ks = 0; ks < KH; ++ks) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(w3s + 16 * ks + 8 * kh);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(w3s + 16 * ks + 8 * kh + 4);
        const f32x4 a0 = a2v[ks][0], a1 = a2v[ks][1];
        const float v[8] = {a0.x > 0.0f ? gr * w0.x : 0.0f, a0.y > 0.0f ? gr * w0.y : 0.0f,
                            a0.z > 0.0f ? gr * w0.z : 0.0f, a0.w > 0.0f ? gr * w0.w : 0.0f,
                            a1.x > 0.0f ? gr * w1.x : 0.0f, a1.y > 0.0f ? gr * w1.y : 0.0f,
                            a1.z > 0.0f ? gr * w1.z : 0.0f, a1.w > 0.0f ? gr * w1.w : 0.0f};
        zb[ks] = split8(v);
      }
    }
    // saved first-layer activation of this row (C-layout order), requested before the product
    f32x4 a1v[NMB][4];
    {
      const float* a1row = sv.a1 + rowc * F + 4 * kh;
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) a1v[mb][q] = *reinterpret_cast<const f32x4*>(a1row + 32 * mb + 8 * q);
    }
    f32x16 acc[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) {
      Frag3 af[NMB];
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) {
        const u32x4* src = W2t + ((size_t)(mb * KH + ks) * 3) * 64 + lane;
        af[mb].h = src[0];
        af[mb].m = src[64];
        af[mb].l = src[128];
      }
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) acc[mb] = mfma_bf16(af[mb].l, zb[ks].h, acc[mb]);
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) acc[mb] = mfma_bf16(af[mb].h, zb[ks].l, acc[mb]);
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) acc[mb] = mfma_bf16(af[mb].m, zb[ks].m, acc[mb]);
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) acc[mb] = mfma_bf16(af[mb].m, zb[ks].h, acc[mb]);
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) acc[mb] = mfma_bf16(af[mb].h, zb[ks].m, acc[mb]);
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) acc[mb] = mfma_bf16(af[mb].h, zb[ks].h, acc[mb]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // dz1 = da1 * [a1 > 0]: stored for the weight gradients, dotted with the embedding column, split for the next
    // product
    Frag3 tb[KS];
    float de = 0.0f;
    {
      float* zrow = dz1 + (size_t)row * F + 4 * kh;
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          float v[8];
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            const f32x4 a = a1v[mb][2 * half + q2];
            const f32x4 wl = *reinterpret_cast<const f32x4*>(wls + 32 * mb + 16 * half + 8 * q2 + 4 * kh);
            v[4 * q2] = a.x > 0.0f ? acc[mb][8 * half + 4 * q2] : 0.0f;
            v[4 * q2 + 1] = a.y > 0.0f ? acc[mb][8 * half + 4 * q2 + 1] : 0.0f;
            v[4 * q2 + 2] = a.z > 0.0f ? acc[mb][8 * half + 4 * q2 + 2] : 0.0f;
            v[4 * q2 + 3] = a.w > 0.0f ? acc[mb][8 * half + 4 * q2 + 3] : 0.0f;
            de = fmaf(v[4 * q2], wl.x, de);
            de = fmaf(v[4 * q2 + 1], wl.y, de);
            de = fmaf(v[4 * q2 + 2], wl.z, de);
            de = fmaf(v[4 * q2 + 3], wl.w, de);
          }
          if (valid) {
            *reinterpret_cast<f32x4*>(zrow + 32 * mb + 16 * half) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(zrow + 32 * mb + 16 * half + 8) = f32x4{v[4], v[5], v[6], v[7]};
          }
          tb[2 * mb + half] = split8(v);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    de += __shfl_xor(de, 32, 64);
    if (valid && kh == 0) demb[row] = de;  // demb[row] = sum_n dz1[row][n] * o1_w[n][F]
    // dfeat = dz1 @ o1_w[:, :F], transposed, two 32-feature blocks at a time
    float* frow = dfeat + (size_t)row * F + 4 * kh;
    constexpr int MP = NMB >= 2 ? 2 : 1;
#pragma unroll
    for (int mb0 = 0; mb0 < NMB; mb0 += MP) {
      f32x16 acc2[MP];
#pragma unroll
      for (int u = 0; u < MP; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[u][r] = 0.0f;
      Frag3 af[MP], an[MP];
#pragma unroll
      for (int u = 0; u < MP; ++u) {
        const u32x4* src = W1t + ((size_t)((mb0 + u) * KS) * 3) * 64 + lane;
        af[u].h = src[0]; af[u].m = src[64]; af[u].l = src[128];
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
#pragma unroll
          for (int u = 0; u < MP; ++u) {
            const u32x4* src = W1t + ((size_t)((mb0 + u) * KS + ks + 1) * 3) * 64 + lane;
            an[u].h = src[0]; an[u].m = src[64]; an[u].l = src[128];
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
        if (ks + 1 < KS) {
#pragma unroll
          for (int u = 0; u < MP; ++u) af[u] = an[u];
        }
      }
      if (valid) {
#pragma unroll
        for (int u = 0; u < MP; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(frow + 32 * (mb0 + u) + 8 * q) =
                f32x4{acc2[u][4 * q], acc2[u][4 * q + 1], acc2[u][4 * q + 2], acc2[u][4 * q + 3]};
      }
    }
  }
}

inline int row_blocks_grid(int64_t S) {
  const int nrb = (int)((S + 31) / 32);
  int nx = (nrb + 7) / 8;
  return nx < 256 ? nx : 256;  // one 8-wave block per CU
}

}  // namespace

extern "C" int64_t geossl_ddm_loss_fwd_workspace_floats(int F) { return GEOSSL_LOSS_PARTIALS; }

extern "C" int geossl_ddm_loss_fwd(const float* h, const int64_t* batch, const int64_t* sei0, const int64_t* sei1,
                                   int64_t S, const float* distance, const int64_t* noise_level,
                                   const float* distance_noise, const GeosslNcsnWeights* w, int F, float anneal_power,
                                   float* loss_e, const GeosslNcsnSaved* saved, float* workspace, hipStream_t stream) {
  if (S <= 0) return 0;
  if (F != 32 && F != 64 && F != 128) return (int)hipErrorInvalidValue;
  GeosslNcsnSaved sv = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (saved != nullptr) sv = *saved;
  const int NMB = F / 32, H = F / 2, HMB = (H + 31) / 32, KS = F / 16;
  const size_t lds = (size_t)(NMB + HMB) * KS * 3 * 1024 + (size_t)(5 * F + 2 * 32 * HMB + 8) * sizeof(float);
  dim3 grid(row_blocks_grid(S));
#define LAUNCH(NMBV)                                                                                              \
  do {                                                                                                            \
    allow_big_lds(&k_ncsn_fwd<NMBV>);                                                                             \
    hipLaunchKernelGGL((k_ncsn_fwd<NMBV>), grid, dim3(512), lds, stream, h, batch, sei0, sei1, (int)S, distance,  \
                       noise_level, distance_noise, *w, anneal_power, loss_e, sv, workspace);                     \
  } while (0)
  if (F == 128) LAUNCH(4); else if (F == 64) LAUNCH(2); else LAUNCH(1);
#undef LAUNCH
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_ddm_loss_fwd2(const GeosslNcsnHeadFwd* heads, const int64_t* batch, const int64_t* sei0,
                                    const int64_t* sei1, int64_t S, int F, hipStream_t stream) {
  return geossl_ddm_loss_fwd2_dyn(heads, batch, sei0, sei1, S, F, nullptr, nullptr, stream);
}

extern "C" int geossl_ddm_loss_fwd2_dyn(const GeosslNcsnHeadFwd* heads, const int64_t* batch, const int64_t* sei0,
                                        const int64_t* sei1, int64_t S, int F, const int32_t* dyn_S,
                                        const int32_t* dyn_view, hipStream_t stream) {
  if (S <= 0) return 0;
  if (heads == nullptr || (F != 32 && F != 64 && F != 128)) return (int)hipErrorInvalidValue;
  NcsnFwdPair pr;
  NcsnFwdHead* a = pr.a;
  for (int k = 0; k < 2; ++k) {
    const GeosslNcsnHeadFwd& hd = heads[k];
    if (hd.h == nullptr || hd.loss_e == nullptr) return (int)hipErrorInvalidValue;
    a[k].h = hd.h; a[k].distance = hd.distance; a[k].noise_level = hd.noise_level; a[k].dist_noise = hd.distance_noise;
    a[k].w = hd.w; a[k].sv = hd.saved; a[k].anneal_power = hd.anneal_power; a[k].loss_e = hd.loss_e;
    a[k].loss_part = hd.workspace;
  }
  const int NMB = F / 32, H = F / 2, HMB = (H + 31) / 32, KS = F / 16;
  const size_t lds = (size_t)(NMB + HMB) * KS * 3 * 1024 + (size_t)(5 * F + 2 * 32 * HMB + 8) * sizeof(float);
  dim3 grid((row_blocks_grid(S) + 1) / 2, 2);  // half of the chip per head: both heads' blocks resident together
#define LAUNCH(NMBV)                                                                                              \
  do {                                                                                                            \
    allow_big_lds(&k_ncsn_fwd2<NMBV>);                                                                            \
    hipLaunchKernelGGL((k_ncsn_fwd2<NMBV>), grid, dim3(512), lds, stream, pr, batch, sei0, sei1, (int)S,           \
                       dyn_S, dyn_view);                                                                          \
  } while (0)
  if (F == 128) LAUNCH(4); else if (F == 64) LAUNCH(2); else LAUNCH(1);
#undef LAUNCH
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_ddm_loss_bwd_rows(const GeosslNcsnWeights* w, const GeosslNcsnSaved* saved, int64_t S, int F,
                                        const int64_t* stats_divisor, float out_scale, const float* gout, float* dz1,
                                        float* dfeat, float* demb, float* grow, hipStream_t stream) {
  if (S <= 0) return 0;
  if (F != 32 && F != 64 && F != 128) return (int)hipErrorInvalidValue;
  const int NMB = F / 32, H = F / 2, KS = F / 16, KH = H / 16;
  const size_t lds = (size_t)NMB * (KS + KH) * 3 * 1024 + (size_t)(F + 32 * ((H + 31) / 32)) * sizeof(float);
  dim3 grid(row_blocks_grid(S));
#define LAUNCH(NMBV)                                                                                              \
  do {                                                                                                            \
    allow_big_lds(&k_ncsn_bwd_rows<NMBV>);                                                                        \
    hipLaunchKernelGGL((k_ncsn_bwd_rows<NMBV>), grid, dim3(512), lds, stream, *w, *saved, (int)S, stats_divisor,  \
                       out_scale, gout, dz1, dfeat, demb, grow);                                                  \
  } while (0)
  if (F == 128) LAUNCH(4); else if (F == 64) LAUNCH(2); else LAUNCH(1);
#undef LAUNCH
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
