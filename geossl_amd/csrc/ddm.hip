// Denoising-distance-matching head on super-edge rows: NCSN_version_03.forward (examples/NCSN.py:183-220) and its
// backward, plus the flat-buffer Adam step (examples/pretrain_GeoSSL.py:258-260,343).
//
// A wave owns 32 super-edge rows.  Forward: gather h[u]+h[v] into a wave-private swizzled LDS tile, the scalar
// distance embedding is folded into the accumulator init as a rank-1 term, GEMM1 (F -> F) runs against
// output_mlp.layers.0.weight held in LDS, the relu'd result overwrites the tile, GEMM2 (F -> F/2) takes its B
// fragments from a k-major copy of layers.1.weight in global memory (L1/L2 resident, 32 KB), the last layer
// (F/2 -> 1) and the loss are per-row VALU work.  All random draws are inputs.
#include "common.h"
#include "geossl_hip.h"
#include "tn.h"
#include "wgrad.h"

using namespace geossl;

namespace {

inline int grid1d(int64_t n, int block, int cap = 2048) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// k-major copy of layers.1.weight: W2T[k][m] = o2_w[m][k], padded to HP columns
// fixed-order two-stage sum
__global__ __launch_bounds__(256) void k_sum_partial(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
  __shared__ float red[256];
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = blockIdx.x * per, hi = min(n, lo + per);
  float s = 0.0f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void k_loss_final(const float* __restrict__ partial, int nblk, const int64_t* __restrict__ divisor,
                             float out_scale, float* __restrict__ loss, int accumulate) {
  if (blockIdx.x != 0) return;
  // one wave, fixed order: lane l sums partials l, l + 64, ... in sequence, then a butterfly over the lanes (a single
  // thread walking the 256 partials was 11 us of dependent loads per head)
  float s = 0.0f;
  for (int b = threadIdx.x; b < nblk; b += 64) s += partial[b];
  s = wave_sum(s);
  if (threadIdx.x != 0) return;
  const float v = (s / (float)divisor[0]) * out_scale;  // loss.mean() over max(edge2graph)+1 graphs, NCSN.py:210-212
  loss[0] = accumulate ? loss[0] + v : v;
}
// the two heads' means and their sum in one launch: loss = scale0 * sum(p0) / divisor + scale1 * sum(p1) / divisor, each
// sum in k_loss_final's order, then added (as loss_01 + loss_02 is, pretrain_GeoSSL.py:210)
__global__ void k_loss_final2(const float* __restrict__ p0, const float* __restrict__ p1, int nblk,
                              const int64_t* __restrict__ divisor, float scale0, float scale1, float* __restrict__ loss) {
  if (blockIdx.x != 0) return;
  float s0 = 0.0f, s1 = 0.0f;
  for (int b = threadIdx.x; b < nblk; b += 64) {
    s0 += p0[b];
    s1 += p1[b];
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (threadIdx.x != 0) return;
  const float v0 = (s0 / (float)divisor[0]) * scale0, v1 = (s1 / (float)divisor[0]) * scale1;
  loss[0] = v0 + v1;
}
#define GEOSSL_LOSS_BLOCKS 256

// ---------------------------------------------------------------------------------------------- backward
// output_mlp.layers.1: dW[m][n] = sum_s da2[s][m] a1[s][n] with da2 = grow * w3 * [a2 > 0] rebuilt from the saved a2
struct NcsnW2Ops {
  const float* grow;
  const float* w3;
  const float* a2;
  const float* a1;
  int F, H;
  static constexpr bool kDot = false;
  struct RawA { float a[2][8]; float g[2][8]; float w; };
  struct RawB { float v[2][8]; };
  __device__ __forceinline__ void prime_a(int, int col, int, int, int, RawA& r) const { r.w = w3[col]; }
  __device__ __forceinline__ void prime_b(int, int, int, int, int, RawB&) const {}
  __device__ __forceinline__ void request_a(int, int col, int row0, int row_end, int kh, RawA& r) const {
    request_col8(a2, H, col, row0, row_end, kh, r.a);
    request_col8(grow, 1, 0, row0, row_end, kh, r.g);
  }
  __device__ __forceinline__ void request_b(int, int col, int row0, int row_end, int kh, RawB& r) const {
    request_col8(a1, F, col, row0, row_end, kh, r.v);
  }
  __device__ __forceinline__ void finish_a(int, int, int row0, int row_end, int kh, RawA& r, float (&out)[2][8],
                                           float (&)[2][8]) const {
    float a[2][8], g[2][8];
    finish_col8(row0, row_end, kh, r.a, a);  // rows past row_end -> a = 0 -> da2 = 0
    finish_col8(row0, row_end, kh, r.g, g);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) out[ks][e] = a[ks][e] > 0.0f ? g[ks][e] * r.w : 0.0f;
  }
  __device__ __forceinline__ void finish_b(int, int, int row0, int row_end, int kh, RawB& r, float (&out)[2][8]) const {
    finish_col8(row0, row_end, kh, r.v, out);
  }
};

// output_mlp.layers.0: dW[n][k<F] = sum_s dz1[s][n] (h[u_s] + h[v_s])[k];  dW[n][F] = sum_s dz1[s][n] emb[s].
// The gathered operand needs the endpoints of a row before its data can be requested: the endpoint indices run one
// iteration ahead of the data requests (two ahead of the use).
struct NcsnW1Ops {
  const float* dz1;
  const float* h;
  const int64_t* sei0;
  const int64_t* sei1;
  const float* emb;
  int F;
  static constexpr bool kDot = true;
  struct RawA { float v[2][8]; float e[2][8]; };
  struct RawB { float hu[2][8]; float hv[2][8]; int u[2][8]; int v[2][8]; };
  __device__ __forceinline__ void load_idx(int row0, int row_end, int kh, RawB& r) const {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = min(row0 + 16 * ks + 8 * kh + e, row_end - 1);
        r.u[ks][e] = (int)sei0[row];
        r.v[ks][e] = (int)sei1[row];
      }
  }
  __device__ __forceinline__ void prime_a(int, int, int, int, int, RawA&) const {}
  __device__ __forceinline__ void prime_b(int, int, int row0, int row_end, int kh, RawB& r) const {
    load_idx(row0, row_end, kh, r);
  }
  __device__ __forceinline__ void request_a(int, int col, int row0, int row_end, int kh, RawA& r) const {
    request_col8(dz1, F, col, row0, row_end, kh, r.v);
    request_col8(emb, 1, 0, row0, row_end, kh, r.e);
  }
  __device__ __forceinline__ void request_b(int, int col, int row0, int row_end, int kh, RawB& r) const {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        r.hu[ks][e] = h[(size_t)r.u[ks][e] * F + col];
        r.hv[ks][e] = h[(size_t)r.v[ks][e] * F + col];
      }
    load_idx(min(row0 + 32, row_end - 1), row_end, kh, r);  // endpoints of the following request
  }
  __device__ __forceinline__ void finish_a(int, int, int row0, int row_end, int kh, RawA& r, float (&out)[2][8],
                                           float (&e)[2][8]) const {
    finish_col8(row0, row_end, kh, r.v, out);
    finish_col8(row0, row_end, kh, r.e, e);
  }
  __device__ __forceinline__ void finish_b(int, int, int row0, int row_end, int kh, RawB& r, float (&out)[2][8]) const {
    float a[2][8], b[2][8];
    finish_col8(row0, row_end, kh, r.hu, a);
    finish_col8(row0, row_end, kh, r.hv, b);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) out[ks][e] = a[ks][e] + b[ks][e];
  }
};

// The vector-shaped gradients of the head: layers.2.weight/bias of output_mlp and the whole input_distance_mlp.
// Block b reduces its row chunk; thread k owns hidden unit k.  partial layout per block: [o3_w H][in_w2 F][in_w1 F]
// [in_b1 F][o3_b 1][in_b2 1].
__global__ __launch_bounds__(128) void k_ncsn_small_partial(GeosslNcsnWeights w, GeosslNcsnSaved sv,
                                                            const float* __restrict__ grow,
                                                            const float* __restrict__ demb, int S, int F, int chunk,
                                                            float* __restrict__ partial) {
  const int H = F / 2, len = H + 3 * F + 2;
  const int lo = blockIdx.x * chunk, hi = min(S, lo + chunk);
  float* P = partial + (size_t)blockIdx.x * len;
  for (int k = threadIdx.x; k < F; k += blockDim.x) {
    const float w1 = w.in_w1[k], b1 = w.in_b1[k], w2 = w.in_w2[k];
    float s_o3 = 0.0f, s_w2 = 0.0f, s_w1 = 0.0f, s_b1 = 0.0f;
    for (int s0 = lo; s0 < hi; s0 += 8) {
      float de[8], pd[8], gr[8], a2v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // eight rows of loads in flight
        const int s = min(s0 + u, hi - 1);
        de[u] = demb[s];
        pd[u] = sv.pd[s];
        gr[u] = grow[s];
        a2v[u] = k < H ? sv.a2[(size_t)s * H + k] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (s0 + u < hi) {
          const float pre = fmaf(w1, pd[u], b1);
          s_w2 = fmaf(de[u], fmaxf(pre, 0.0f), s_w2);
          const float dp = pre > 0.0f ? de[u] * w2 : 0.0f;
          s_w1 = fmaf(dp, pd[u], s_w1);
          s_b1 += dp;
          s_o3 = fmaf(gr[u], a2v[u], s_o3);
        }
      }
    }
    if (k < H) P[k] = s_o3;
    P[H + k] = s_w2;
    P[H + F + k] = s_w1;
    P[H + 2 * F + k] = s_b1;
  }
  if (threadIdx.x == 0) {
    float sg = 0.0f, sd = 0.0f;
    for (int s = lo; s < hi; ++s) {
      sg += grow[s];
      sd += demb[s];
    }
    P[H + 3 * F] = sg;
    P[H + 3 * F + 1] = sd;
  }
}
// one 64-lane block per output scalar: lanes stride over the row chunks, fixed-order butterfly at the end
__global__ __launch_bounds__(64) void k_ncsn_small_reduce(const float* __restrict__ partial, int nblk, int F,
                                                          GeosslNcsnGrads g, int accumulate) {
  const int H = F / 2, len = H + 3 * F + 2;
  const int i = blockIdx.x;
  if (i >= len) return;
  float s = 0.0f;
  for (int b = threadIdx.x; b < nblk; b += 64) s += partial[(size_t)b * len + i];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    float* dst;
    if (i < H) dst = g.o3_w + i;
    else if (i < H + F) dst = g.in_w2 + (i - H);
    else if (i < H + 2 * F) dst = g.in_w1 + (i - H - F);
    else if (i < H + 3 * F) dst = g.in_b1 + (i - H - 2 * F);
    else if (i == H + 3 * F) dst = g.o3_b;
    else dst = g.in_b2;
    *dst = accumulate ? *dst + s : s;
  }
}

// dh[a] (+)= sum over incident super-edges (fixed order) of dfeat[s]; one wave per atom
__global__ __launch_bounds__(256) void k_incidence_gather(const float* __restrict__ dfeat0,
                                                          const int64_t* __restrict__ inc_ptr,
                                                          const int32_t* __restrict__ inc_idx, int N, int F,
                                                          float* __restrict__ dh0, int accumulate,
                                                          const float* __restrict__ dfeat1, float* __restrict__ dh1,
                                                          const int32_t* __restrict__ dyn_view) {
  // (blockIdx.y = 1: the second head of a two-head launch - same incidence lists, its own rows)
  const float* __restrict__ dfeat = blockIdx.y == 0 ? dfeat0 : dfeat1;
  float* __restrict__ dh = blockIdx.y == 0 ? dh0 : dh1;
  // capacity launch: *dyn_view = real atoms of a view; both heads were handed the base of ONE [view 0 ; view 1] gradient
  // tensor, head 1 writes its rows behind head 0's
  N = dyn_count(N, dyn_view);
  if (dyn_view != nullptr && blockIdx.y == 1) dh += (size_t)N * F;
  // A super-edge row is read twice, by its two atoms - atoms of ONE molecule, i.e. of neighbouring groups of four.
  // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): consecutive groups on consecutive
  // workgroups would fetch a molecule's rows into several L2s, from HBM every time.  Group g = (block mod 8) * per +
  // block / 8 keeps neighbouring groups on one XCD and close in time: the second read is an L2 hit.
  const int per = ((int)gridDim.x + 7) / 8;
  const int grp = ((int)blockIdx.x % 8) * per + (int)blockIdx.x / 8;
  const int a = grp * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (a >= N) return;
  const int64_t p0 = inc_ptr[a], p1 = inc_ptr[a + 1];
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // F <= 256
  if (accumulate) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (lane + 64 * v < F) acc[v] = dh[(size_t)a * F + lane + 64 * v];
  }
  for (int64_t pb = p0; pb < p1; pb += 64) {
    const int cnt = (int)min((int64_t)64, p1 - pb);
    const int mine = lane < cnt ? inc_idx[pb + lane] : 0;  // this chunk's super-edge ids, one per lane
    for (int q0 = 0; q0 < cnt; q0 += 8) {
      float val[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = __shfl(mine, min(q0 + u, cnt - 1), 64);
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (lane + 64 * v < F) val[u][v] = dfeat[(size_t)s * F + lane + 64 * v];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (q0 + u < cnt) {
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (lane + 64 * v < F) acc[v] += val[u][v];
        }
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v)
    if (lane + 64 * v < F) dh[(size_t)a * F + lane + 64 * v] = acc[v];
}

// ------------------------------------------------------------------------------------------------- Adam
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, int64_t n, float step_size, float w1, float beta2, float w2, float eps,
                       float wd, float bc2_sqrt, float grad_scale) {
  // torch.optim.Adam's default (foreach) device path, rounding step by rounding step - found by comparing forms against
  // it bit for bit (tools/probes/adam_probe.hip: 0 differing elements of param / exp_avg / exp_avg_sq over four steps):
  //   exp_avg.lerp_(grad, 1 - beta1)                          = fma(w1, grad - exp_avg, exp_avg)
  //   exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)  = fma(w2, grad * grad, exp_avg_sq * beta2)
  //   denom = exp_avg_sq.sqrt() / bias_correction2_sqrt + eps
  //   param.addcdiv_(exp_avg, denom, value=-lr / bc1)         = fma(-step_size, exp_avg / denom, param)
  // with w1 = float(1 - beta1), w2 = float(1 - beta2) formed in DOUBLE on the host like torch forms them (1.0f - 0.999f
  // is 4.7e-5 away from float(0.001): the form this kernel had until round 4)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = mul_rn(g[i], grad_scale);
    const float pi = p[i];
    if (wd != 0.0f) gi = fmaf(wd, pi, gi);
    const float m0 = m[i];
    const float mi = fmaf(w1, add_rn(gi, -m0), m0);
    const float vi = fmaf(w2, mul_rn(gi, gi), mul_rn(v[i], beta2));
    m[i] = mi;
    v[i] = vi;
    const float denom = add_rn(sqrtf(vi) / bc2_sqrt, eps);
    p[i] = fmaf(-step_size, mi / denom, pi);
  }
}

}  // namespace

namespace geossl {
int launch_ncsn_small_reduce(const float* partial, int nblk, int F, const GeosslNcsnGrads& g, int accumulate,
                             hipStream_t stream);
}

extern "C" int64_t geossl_loss_reduce_workspace_floats(int64_t S) { return GEOSSL_LOSS_BLOCKS; }

extern "C" int geossl_loss_reduce(const float* loss_e, int64_t S, const int64_t* stats_divisor, float out_scale,
                                  float* loss, float* workspace, int accumulate, hipStream_t stream) {
  hipLaunchKernelGGL(k_sum_partial, dim3(GEOSSL_LOSS_BLOCKS), dim3(256), 0, stream, loss_e, S, workspace);
  GEOSSL_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(64), 0, stream, workspace, GEOSSL_LOSS_BLOCKS, stats_divisor, out_scale,
                     loss, accumulate);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_loss_reduce_partials(const float* partial, const int64_t* stats_divisor, float out_scale,
                                           float* loss, int accumulate, hipStream_t stream) {
  hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(64), 0, stream, partial, GEOSSL_LOSS_PARTIALS, stats_divisor, out_scale,
                     loss, accumulate);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_loss_reduce_partials2(const float* partial0, const float* partial1, const int64_t* stats_divisor,
                                            float scale0, float scale1, float* loss, hipStream_t stream) {
  hipLaunchKernelGGL(k_loss_final2, dim3(1), dim3(64), 0, stream, partial0, partial1, GEOSSL_LOSS_PARTIALS, stats_divisor,
                     scale0, scale1, loss);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

static inline void small_plan(int64_t S, int* chunk, int* nblk) {
  int64_t c = (S + 2047) / 2048;  // eight blocks per CU: the row loop of a thread is a chain of dependent batches
  if (c < 64) c = 64;
  *chunk = (int)c;
  *nblk = (int)((S + c - 1) / c);
  if (*nblk < 1) *nblk = 1;
}

extern "C" int64_t geossl_ddm_loss_bwd_workspace_floats(int64_t S, int F) {
  int chunk, nblk;
  small_plan(S, &chunk, &nblk);
  const int64_t a = tn_workspace_floats(S, F, F, 1), b = (int64_t)nblk * (F / 2 + 3 * F + 2);
  return a > b ? a : b;
}

extern "C" int geossl_ddm_loss_bwd_weights(const float* h, const int64_t* sei0, const int64_t* sei1, int64_t S, int F,
                                           const GeosslNcsnWeights* w, const GeosslNcsnSaved* saved, const float* dz1,
                                           const float* demb, const float* grow, const GeosslNcsnGrads* grads,
                                           float* workspace, int accumulate, hipStream_t stream) {
  if (S <= 0) return 0;
  const int H = F / 2;
  WgradOut o;
  for (int z = 0; z < GEOSSL_TN_MAX; ++z) o.dW[z] = o.db[z] = o.dd[z] = nullptr;
  int rc;
  // output_mlp.layers.1: dW[m][k] = sum_s da2[s][m] a1[s][k]
  o.dW[0] = grads->o2_w;
  o.db[0] = grads->o2_b;
  NcsnW2Ops l2{grow, w->o3_w, saved->a2, saved->a1, F, H};
  if (F == 128) rc = launch_wgrad_split<2, 4>(l2, 1, S, H, F, o, F, 1, workspace, accumulate, stream);
  else if (F == 64) rc = launch_wgrad_split<1, 2>(l2, 1, S, H, F, o, F, 1, workspace, accumulate, stream);
  else rc = launch_wgrad_split<1, 1>(l2, 1, S, H, F, o, F, 1, workspace, accumulate, stream);
  if (rc) return rc;
  // output_mlp.layers.0: dW[n][k<F] = sum_s dz1[s][n] (h[u]+h[v])[s][k];  dW[n][F] = sum_s dz1[s][n] emb[s]
  o.dW[0] = grads->o1_w;
  o.db[0] = grads->o1_b;
  o.dd[0] = grads->o1_w + F;
  NcsnW1Ops l1{dz1, h, sei0, sei1, saved->emb, F};
  if (F == 128) rc = launch_wgrad_split<4, 4>(l1, 1, S, F, F, o, F + 1, F + 1, workspace, accumulate, stream);
  else if (F == 64) rc = launch_wgrad_split<2, 2>(l1, 1, S, F, F, o, F + 1, F + 1, workspace, accumulate, stream);
  else rc = launch_wgrad_split<1, 1>(l1, 1, S, F, F, o, F + 1, F + 1, workspace, accumulate, stream);
  if (rc) return rc;
  int chunk, nblk;
  small_plan(S, &chunk, &nblk);
  hipLaunchKernelGGL(k_ncsn_small_partial, dim3(nblk), dim3(128), 0, stream, *w, *saved, grow, demb, (int)S, F, chunk,
                     workspace);
  GEOSSL_CHECK_LAUNCH();
  return launch_ncsn_small_reduce(workspace, nblk, F, *grads, accumulate, stream);
}

namespace geossl {
int launch_ncsn_small_reduce(const float* partial, int nblk, int F, const GeosslNcsnGrads& g, int accumulate,
                             hipStream_t stream) {
  hipLaunchKernelGGL(k_ncsn_small_reduce, dim3(F / 2 + 3 * F + 2), dim3(64), 0, stream, partial, nblk, F, g, accumulate);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
}  // namespace geossl

extern "C" int geossl_incidence_gather(const float* dfeat, const int64_t* inc_ptr, const int32_t* inc_idx, int64_t N,
                                       int F, float* dh, int accumulate, hipStream_t stream) {
  if (N <= 0) return 0;
  const unsigned groups = (unsigned)((N + 3) / 4);
  hipLaunchKernelGGL(k_incidence_gather, dim3((groups + 7) / 8 * 8), dim3(256), 0, stream, dfeat, inc_ptr, inc_idx,
                     (int)N, F, dh, accumulate, (const float*)nullptr, (float*)nullptr, (const int32_t*)nullptr);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

namespace geossl {
// dh of both heads in one launch (geossl_ddm_loss_bwd_fused2, ncsn_bwd.hip)
int launch_incidence_gather2(const float* dfeat0, const float* dfeat1, const int64_t* inc_ptr, const int32_t* inc_idx,
                             int64_t N, int F, float* dh0, float* dh1, hipStream_t stream, const int32_t* dyn_view) {
  if (N <= 0) return 0;
  const unsigned groups = (unsigned)((N + 3) / 4);
  hipLaunchKernelGGL(k_incidence_gather, dim3((groups + 7) / 8 * 8, 2), dim3(256), 0, stream, dfeat0, inc_ptr, inc_idx,
                     (int)N, F, dh0, 0, dfeat1, dh1, dyn_view);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
}  // namespace geossl

extern "C" int geossl_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                                double beta1, double beta2, double eps, double weight_decay, int64_t step_count,
                                float grad_scale, hipStream_t stream) {
  if (n <= 0) return 0;
  // (hyperparameters arrive as the doubles Python holds; every derived scalar is formed in double and rounded once)
  const double bc1 = 1.0 - pow(beta1, (double)step_count);
  const double bc2 = 1.0 - pow(beta2, (double)step_count);
  hipLaunchKernelGGL(k_adam, dim3(grid1d(n, 256)), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, n,
                     (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                     (float)weight_decay, (float)sqrt(bc2), grad_scale);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
