// Backward of the continuous-filter network (InteractionBlock.mlp inside CFConv, schnet.py:141-145,186-187) with
// respect to its weights, for all interaction blocks in two launches.
//
// Upstream gradient per pair slot p = (i<j) of layer l (never stored per pair in HBM):
//     dO[p][n] = C(d_p) * ( flag0 * dagg_l[i][n] * x_l[j][n]  +  flag1 * dagg_l[j][n] * x_l[i][n] )
// Both kernels work on 64-row tiles of pair slots.  A tile touches the atoms of at most a few molecules, so the rows
// a_lo..a_hi of x_l and dagg_l are staged in LDS with coalesced 16-byte loads, and the tile of dO is built ONCE per
// block in LDS, n-major ([n][64 rows], row stride 65): it is the A operand of both products below with nothing but
// base + immediate-offset ds_reads in the MFMA loops.  ~75 KB of LDS -> two blocks per CU.
//
//   k_filter_bwd_a :  dt = dO W2 ; dU = dt * ssp'(.) ; dW1 += dU^T rbf(d) ; db1 += sum dU
//       column-split waves: wave w owns hidden units [32w, 32w+32) and keeps that slice of W2 as B fragments in
//       registers.  dt's accumulators (C layout: lane = hidden unit, register = pair row) ARE the A operand of the
//       dW1 product when the contraction slot of k-step s is the row held in register s — dU never leaves registers.
//   k_filter_bwd_b :  dW2 += dO^T T ; db2 += sum dO
//       wave w owns rows n in [32w, 32w+32) of dW2; A fragments are single ds_reads of the dO tile, B fragments are
//       16-byte global loads of the saved hidden activation T (each half-wave reads one full 512-byte row), with the
//       output columns of an accumulator block taken as {4j + c} so that no transposition is needed.
// Weight-gradient accumulators stay in registers across all tiles of a block; one partial per block, fixed-order
// reduction afterwards (no atomics).
#include "common.h"
#include "geossl_hip.h"
#include "tn.h"

using namespace geossl;

namespace {

constexpr int TR = 64;         // pair rows per tile
constexpr int TS = TR + 1;     // row stride of the n-major dO tile
constexpr int ATOM_CAP = 40;   // atoms staged per tile (two 18..20-atom molecules, or more smaller ones)

template <typename K>
inline void allow_big_lds(K kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
}

// LDS carve shared by both kernels
template <int F>
struct TileLds {
  static constexpr int AS = F + 1;
  float* dO;     // [F][TS]
  float* xs;     // [ATOM_CAP][AS]
  float* ds;     // [ATOM_CAP][AS]
  float* tdd;    // [TR] distances of the tile's rows
  int4* desc;    // [TR] {LDS offset of atom i, of atom j, C*flag0, C*flag1} (staged) or {i*F, j*F, ..} (global)
  int* s_amax;   // [1]
  __device__ explicit TileLds(float* smem) {
    dO = smem;
    xs = dO + F * TS;
    ds = xs + ATOM_CAP * AS;
    tdd = ds + ATOM_CAP * AS;
    desc = reinterpret_cast<int4*>(smem + ((F * TS + 2 * ATOM_CAP * AS + TR + 3) & ~3));
    s_amax = reinterpret_cast<int*>(desc + TR);
  }
  static size_t bytes() { return ((size_t)((F * TS + 2 * ATOM_CAP * AS + TR + 3) & ~3) + 4 * TR + 4) * sizeof(float); }
};

// rows [a_lo, a_lo+na) of src[N][F] -> dst[na][F+1], 16-byte global loads
template <int F, int NT>
__device__ __forceinline__ void stage_atoms(const float* __restrict__ src, int a_lo, int na, float* dst, int tid) {
  constexpr int Q = F / 4;
  const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)a_lo * F);
#pragma unroll 4
  for (int i = tid; i < na * Q; i += NT) {
    const int a = i / Q, q = i - a * Q;
    const float4 v = s4[i];
    float* d = dst + a * (F + 1) + 4 * q;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

// Builds the dO tile of rows [r0, r0+64) of layer l in LDS (NW waves).  Starts with a barrier (the previous tile must
// be fully consumed) and ends with one; the tile and tdd are valid afterwards.
// `alo` = first atom of the tile's window (= pair_i[r0]: pair slots are lexicographic inside a molecule, so the first
// row holds the smallest atom); it is fetched one tile ahead by the caller, which lets the row descriptors and a
// fixed ATOM_CAP-row window of x / dagg be requested together — one global round trip per tile instead of two.
template <int F, int NW>
__device__ __forceinline__ void build_dO_tile(const TileLds<F>& L, const float* __restrict__ pair_d,
                                              const float* __restrict__ pair_c, const uint8_t* __restrict__ pair_flag,
                                              const int32_t* __restrict__ pair_i, const int32_t* __restrict__ pair_j,
                                              int P, int N, int r0, int alo, const float* __restrict__ x,
                                              const float* __restrict__ dagg, int tid) {
  constexpr int AS = F + 1, NT = 64 * NW, Q = F / 4;
  constexpr int NPRE = (ATOM_CAP * Q + NT - 1) / NT;  // float4 per thread and array for the atom window
  const int lane = tid & 63, wave = tid >> 6;
  // ---- requests: atom window (speculative, fixed size) and row descriptors
  const int nwin = min(ATOM_CAP, N - alo);
  const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)alo * F);
  const float4* d4 = reinterpret_cast<const float4*>(dagg + (size_t)alo * F);
  float4 px[NPRE], pdg[NPRE];
#pragma unroll
  for (int u = 0; u < NPRE; ++u) {
    const int i = tid + NT * u;
    const bool ok = i < nwin * Q;
    px[u] = ok ? x4[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    pdg[u] = ok ? d4[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  int ai = 0, aj = 0;
  float m0 = 0.0f, m1 = 0.0f, dd = 0.0f;
  if (wave == 0) {  // one pair row per lane
    const int row = r0 + lane;
    const bool ok = row < P;
    const int q = ok ? row : P - 1;
    ai = pair_i[q];
    aj = pair_j[q];
    const unsigned fl = ok ? pair_flag[q] : 0u;
    const float c = pair_c[q];
    m0 = (fl & 1u) ? c : 0.0f;
    m1 = (fl & 2u) ? c : 0.0f;
    dd = pair_d[q];
  }
  __syncthreads();  // previous tile fully consumed: LDS may be overwritten
  int amax = aj + 1;
  if (wave == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = max(amax, __shfl_xor(amax, o, 64));
    const bool staged0 = amax - alo <= ATOM_CAP;
    L.tdd[lane] = dd;
    L.desc[lane] = staged0 ? make_int4((ai - alo) * AS, (aj - alo) * AS, __float_as_int(m0), __float_as_int(m1))
                           : make_int4(ai, aj, __float_as_int(m0), __float_as_int(m1));
    if (lane == 0) *L.s_amax = amax;
  }
#pragma unroll
  for (int u = 0; u < NPRE; ++u) {
    const int i = tid + NT * u;
    if (i < nwin * Q) {
      const int a = i / Q, q4 = i - a * Q;
      float* dx = L.xs + a * AS + 4 * q4;
      float* dg = L.ds + a * AS + 4 * q4;
      dx[0] = px[u].x; dx[1] = px[u].y; dx[2] = px[u].z; dx[3] = px[u].w;
      dg[0] = pdg[u].x; dg[1] = pdg[u].y; dg[2] = pdg[u].z; dg[3] = pdg[u].w;
    }
  }
  __syncthreads();
  const bool staged = *L.s_amax - alo <= ATOM_CAP;
  // lane = pair row, wave w takes n = w, w+NW, ...
  const int4 q = L.desc[lane];
  const float qm0 = __int_as_float(q.z), qm1 = __int_as_float(q.w);
  if (staged) {
    const float* di = L.ds + q.x;
    const float* dj = L.ds + q.y;
    const float* xi = L.xs + q.x;
    const float* xj = L.xs + q.y;
#pragma unroll 4
    for (int n = wave; n < F; n += NW) L.dO[n * TS + lane] = qm0 * (di[n] * xj[n]) + qm1 * (dj[n] * xi[n]);
  } else {
    // atom window larger than the LDS stage (a run of tiny molecules): operands straight from global memory
    const float* di = dagg + (size_t)q.x * F;
    const float* dj = dagg + (size_t)q.y * F;
    const float* xi = x + (size_t)q.x * F;
    const float* xj = x + (size_t)q.y * F;
    for (int n = wave; n < F; n += NW) L.dO[n * TS + lane] = qm0 * (di[n] * xj[n]) + qm1 * (dj[n] * xi[n]);
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------ kernel A
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void k_filter_bwd_a(const float* __restrict__ pair_d,
                                                             const float* __restrict__ pair_c,
                                                             const uint8_t* __restrict__ pair_flag,
                                                             const int32_t* __restrict__ pair_i,
                                                             const int32_t* __restrict__ pair_j, int P, int N,
                                                             GeosslFilterWeights w, GeosslFilterGradIn g, int G,
                                                             const float* __restrict__ offset, float coeff,
                                                             const float* __restrict__ T,
                                                             float* __restrict__ partial_w1,
                                                             float* __restrict__ partial_b1) {
  constexpr int F = 32 * NW, K2 = F / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const TileLds<F> L(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int l = blockIdx.y;
  const int col = 32 * wave + j;  // hidden unit owned by this lane
  // B fragments of dt = dO W2 for this wave's columns: bw2[kk] = W2[n = 2kk+kh][col]
  float bw2[K2];
  {
    const float* w2 = w.w2[l] + col;
#pragma unroll
    for (int kk = 0; kk < K2; ++kk) bw2[kk] = w2[(size_t)(2 * kk + kh) * F];
  }
  const float* __restrict__ x = g.x[l];
  const float* __restrict__ dagg = g.dagg[l];
  const size_t lbase = (size_t)l * P;
  f32x16 accw[2];  // dW1 rows [32w, 32w+32) x gaussians [0, 64)
  float bsum = 0.0f;
#pragma unroll
  for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[g2][r] = 0.0f;
  float offr[2];
#pragma unroll
  for (int g2 = 0; g2 < 2; ++g2) offr[g2] = (32 * g2 + j) < G ? offset[32 * g2 + j] : 0.0f;
  const int ntiles = (P + TR - 1) / TR;
  const int per = (ntiles + gridDim.x - 1) / gridDim.x;  // contiguous tile range per block (atom reuse in L2)
  const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
  int alo_next = t_begin < t_end ? pair_i[t_begin * TR] : 0;
  for (int t = t_begin; t < t_end; ++t) {
    const int r0 = t * TR;
    const int alo = alo_next;
    if (t + 1 < t_end) alo_next = pair_i[r0 + TR];  // one tile ahead: the next build needs it before anything else
    // saved hidden activation of the tile's rows for this lane's hidden unit, C layout (requested before the build)
    float tc[2][16];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(r0 + 32 * rb + c_row(r, lane), P - 1);
        tc[rb][r] = T[(lbase + row) * F + col];
      }
    build_dO_tile<F, NW>(L, pair_d, pair_c, pair_flag, pair_i, pair_j, P, N, r0, alo, x, dagg, tid);
    f32x16 acc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
    {
      const float* abase = L.dO + kh * TS + j;  // A[row = 32rb+j][n = 2kk+kh] = abase[kk*2*TS + 32*rb]
      float a_cur[2], a_nxt[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) a_cur[rb] = abase[32 * rb];
#pragma unroll
      for (int kk = 0; kk < K2; ++kk) {
        const int kn = kk + 1 < K2 ? kk + 1 : kk;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) a_nxt[rb] = abase[kn * 2 * TS + 32 * rb];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[rb], bw2[kk], acc[rb], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) a_cur[rb] = a_nxt[rb];
      }
    }
    // dU = dt * ssp'(pre) in place (C layout: lane = hidden unit, register = pair row); then
    // dW1[k][g] += sum_rows dU[row][k] * rbf(d_row)[g]: k-step (rb, s) contracts over the two rows held in register s
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[rb][r] *= dssp_from_out(tc[rb][r]);
        bsum += acc[rb][r];
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float dd = L.tdd[32 * rb + c_row(s, lane)];
        float bv[2];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const float diff = dd - offr[g2];
          bv[g2] = (32 * g2 + j) < G ? __expf(coeff * (diff * diff)) : 0.0f;
        }
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
          accw[g2] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[rb][s], bv[g2], accw[g2], 0, 0, 0);
      }
    }
  }
  // one partial per block: wave w writes rows [32w, 32w+32) of [F][G] and of [F]
  const size_t pb = (size_t)l * gridDim.x + blockIdx.x;
  float* Pw = partial_w1 + pb * F * G;
#pragma unroll
  for (int g2 = 0; g2 < 2; ++g2) {
    const int gg = 32 * g2 + j;
    if (gg < G) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Pw[(size_t)(32 * wave + c_row(r, lane)) * G + gg] = accw[g2][r];
    }
  }
  const float s = bsum + __shfl_xor(bsum, 32, 64);
  if (kh == 0) partial_b1[pb * F + col] = s;
}

// ------------------------------------------------------------------------------------------------ kernel B
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void k_filter_bwd_b(const float* __restrict__ pair_d,
                                                             const float* __restrict__ pair_c,
                                                             const uint8_t* __restrict__ pair_flag,
                                                             const int32_t* __restrict__ pair_i,
                                                             const int32_t* __restrict__ pair_j, int P, int N,
                                                             GeosslFilterGradIn g, const float* __restrict__ T,
                                                             float* __restrict__ partial_w2,
                                                             float* __restrict__ partial_b2) {
  constexpr int F = 32 * NW, NCB = F / 32;  // accumulator blocks per wave; block c holds columns {NCB*j + c}
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const TileLds<F> L(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int l = blockIdx.y;
  const float* __restrict__ x = g.x[l];
  const float* __restrict__ dagg = g.dagg[l];
  const size_t lbase = (size_t)l * P;
  const int nrow = 32 * wave + j;  // dW2 row (= filter output channel n) this lane feeds as A operand
  f32x16 acc[NCB];
#pragma unroll
  for (int c = 0; c < NCB; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
  float bsum = 0.0f;
  const int ntiles = (P + TR - 1) / TR;
  const int per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
  int alo_next = t_begin < t_end ? pair_i[t_begin * TR] : 0;
  for (int t = t_begin; t < t_end; ++t) {
    const int r0 = t * TR;
    const int alo = alo_next;
    if (t + 1 < t_end) alo_next = pair_i[r0 + TR];
    // B fragments of the whole tile first: row 2kk+kh of T, this lane's NCB consecutive columns.  They do not depend on
    // the dO tile, so the 32 loads fly while the tile is being built; the MFMA loop below then touches LDS only.
    float bv[TR / 2][NCB];
    {
      const float* tbase = T + lbase * F + NCB * j;
#pragma unroll
      for (int kk = 0; kk < TR / 2; ++kk) {
        const int row = min(r0 + 2 * kk + kh, P - 1);  // rows past P have dO = 0
        if constexpr (NCB == 4) {
          const float4 v = *reinterpret_cast<const float4*>(tbase + (size_t)row * F);
          bv[kk][0] = v.x; bv[kk][1] = v.y; bv[kk][2] = v.z; bv[kk][3] = v.w;
        } else if constexpr (NCB == 2) {
          const float2 v = *reinterpret_cast<const float2*>(tbase + (size_t)row * F);
          bv[kk][0] = v.x; bv[kk][1] = v.y;
        } else {
          bv[kk][0] = tbase[(size_t)row * F];
        }
      }
    }
    build_dO_tile<F, NW>(L, pair_d, pair_c, pair_flag, pair_i, pair_j, P, N, r0, alo, x, dagg, tid);
    const float* abase = L.dO + nrow * TS + kh;  // A[i = n][kslot kh] of k-step kk = dO[row 2kk+kh][n] = abase[2kk]
    float a_cur = abase[0];
#pragma unroll
    for (int kk = 0; kk < TR / 2; ++kk) {
      const float a_nxt = abase[2 * (kk + 1 < TR / 2 ? kk + 1 : kk)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < NCB; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, bv[kk][c], acc[c], 0, 0, 0);
      bsum += a_cur;
      a_cur = a_nxt;
    }
  }
  const size_t pb = (size_t)l * gridDim.x + blockIdx.x;
  float* Pw = partial_w2 + pb * F * F;
#pragma unroll
  for (int c = 0; c < NCB; ++c) {
    const int k = NCB * j + c;  // accumulator block c holds columns {NCB*j + c}
#pragma unroll
    for (int r = 0; r < 16; ++r) Pw[(size_t)(32 * wave + c_row(r, lane)) * F + k] = acc[c][r];
  }
  const float s = bsum + __shfl_xor(bsum, 32, 64);
  if (kh == 0) partial_b2[pb * F + nrow] = s;
}

inline int blocks_per_layer(int L, int ntiles) {
  int b = 512 / (L > 0 ? L : 1);  // two blocks per CU
  if (b < 1) b = 1;
  if (b > ntiles) b = ntiles;
  return b;
}

}  // namespace

extern "C" int64_t geossl_cfconv_filter_bwd_workspace_floats(int64_t P, int L, int F, int G) {
  const int ntiles = (int)((P + TR - 1) / TR);
  const int64_t nb = blocks_per_layer(L, ntiles);
  return (int64_t)L * nb * ((int64_t)F * G + F + (int64_t)F * F + F);
}

extern "C" int geossl_cfconv_filter_bwd(const float* pair_d, const float* pair_c, const uint8_t* pair_flag,
                                        const int32_t* pair_i, const int32_t* pair_j, int64_t P, int64_t N,
                                        const GeosslFilterWeights* w, const GeosslFilterGradIn* g, int L, int F, int G,
                                        const float* offset, float coeff, const float* T,
                                        const GeosslFilterGradOut* out, float* workspace, int accumulate,
                                        hipStream_t stream) {
  if (P <= 0 || L <= 0) return 0;
  if (L > GEOSSL_MAX_L || (F != 32 && F != 64 && F != 128) || G > 64) return (int)hipErrorInvalidValue;
  const int ntiles = (int)((P + TR - 1) / TR);
  const int nb = blocks_per_layer(L, ntiles);
  dim3 grid(nb, L);
  float* pw1 = workspace;                          // [L][nb][F][G]
  float* pb1 = pw1 + (size_t)L * nb * F * G;       // [L][nb][F]
  float* pw2 = pb1 + (size_t)L * nb * F;           // [L][nb][F][F]
  float* pb2 = pw2 + (size_t)L * nb * F * F;       // [L][nb][F]
#define LAUNCH(NW)                                                                                                  \
  do {                                                                                                              \
    const size_t lds = TileLds<32 * NW>::bytes();                                                                   \
    allow_big_lds(&k_filter_bwd_a<NW>);                                                                             \
    allow_big_lds(&k_filter_bwd_b<NW>);                                                                             \
    hipLaunchKernelGGL((k_filter_bwd_a<NW>), grid, dim3(64 * NW), lds, stream, pair_d, pair_c, pair_flag, pair_i,   \
                       pair_j, (int)P, (int)N, *w, *g, G, offset, coeff, T, pw1, pb1);                                      \
    hipLaunchKernelGGL((k_filter_bwd_b<NW>), grid, dim3(64 * NW), lds, stream, pair_d, pair_c, pair_flag, pair_i,   \
                       pair_j, (int)P, (int)N, *g, T, pw2, pb2);                                                            \
  } while (0)
  if (F == 128) LAUNCH(4); else if (F == 64) LAUNCH(2); else LAUNCH(1);
#undef LAUNCH
  GEOSSL_CHECK_LAUNCH();
  GeosslReduceBatch rb;
  auto reduce = [&](float* const* dst, const float* partial, int nblk, int len, int ncols) {
    for (int z = 0; z < GEOSSL_TN_MAX; ++z) rb.out[z] = z < L ? dst[z] : nullptr;
    hipLaunchKernelGGL(k_reduce_partials, dim3((len + 63) / 64, L), dim3(256), 0, stream, rb, partial, nblk, len, ncols,
                       ncols, 1, accumulate);
  };
  reduce(out->dw1, pw1, nb, F * G, G);
  reduce(out->db1, pb1, nb, F, F);
  reduce(out->dw2, pw2, nb, F * F, F);
  reduce(out->db2, pb2, nb, F, F);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
