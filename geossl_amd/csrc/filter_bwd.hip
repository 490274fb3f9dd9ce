// Backward of the continuous-filter network (InteractionBlock.mlp inside CFConv, schnet.py:141-145,186-187) with
// respect to its weights, for all interaction blocks in two launches.
//
// Upstream gradient per pair slot p = (i<j) of layer l (never stored per pair):
//     dO[p][n] = C(d_p) * ( flag0 * dagg_l[i][n] * x_l[j][n]  +  flag1 * dagg_l[j][n] * x_l[i][n] )
// A 128-row tile of pair slots touches the atoms of at most a few molecules, so each tile first stages the rows
// a_lo..a_hi of x_l and dagg_l in LDS (coalesced, once) and every MFMA operand that involves dO is then formed
// from LDS on the fly — no per-pair gathers from L2, no dO / dU round trip through HBM.
//
//   k_filter_bwd_a (grid: blocks x L):  dt = dO W2 ; dU = dt * ssp'(.) ; dW1 += dU^T rbf(d) ; db1 += sum dU
//       dt's accumulators (C layout: lane = hidden unit, reg = pair row) ARE the A operand of the dW1 product when
//       the contraction slot of k-step s is taken to be the row held in register s — no transpose, dU never leaves
//       the register file.  rbf(d) is recomputed in B-fragment layout.
//   k_filter_bwd_b (grid: blocks x L):  dW2 += dO^T T ; db2 += sum dO
//       wave w owns rows n in [32w, 32w+32) of dW2 and contracts over all 128 pair rows of the tile; A fragments
//       from the staged atoms, B fragments from a row-major LDS copy of the saved hidden activations T.
// Both keep their weight-gradient accumulators in registers across all tiles of the block and write ONE partial per
// wave / block; a fixed-order reduction finishes (no atomics).
#include "common.h"
#include "geossl_hip.h"
#include "tn.h"
#include <stdlib.h>

using namespace geossl;

namespace {

constexpr int ATOM_CAP = 48;  // atoms staged per 128-row tile (2 molecules of <= 24 atoms, or more smaller ones)

template <typename K>
inline void allow_big_lds(K kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
}

struct TileDesc {  // per pair row of the current tile, in LDS
  int* ti;         // atom i (absolute)
  int* tj;         // atom j (absolute)
  float* tm0;      // C(d) * flag0  (0 past the end)
  float* tm1;      // C(d) * flag1
  float* td;       // distance
};

// Fill the row tables for rows [r0, r0+128) and return the staged atom range through LDS ints.
__device__ __forceinline__ void load_tile_desc(const float* __restrict__ pair_d, const float* __restrict__ pair_c,
                                               const uint8_t* __restrict__ pair_flag,
                                               const int32_t* __restrict__ pair_i, const int32_t* __restrict__ pair_j,
                                               int P, int r0, const TileDesc& t, int* s_amax, int tid,
                                               int nrows = 128) {
  if (tid < nrows) {
    const int row = r0 + tid;
    const bool ok = row < P;
    const int r = ok ? row : P - 1;
    const int ai = pair_i[r], aj = pair_j[r];
    const unsigned fl = ok ? pair_flag[r] : 0u;
    const float c = pair_c[r];
    t.ti[tid] = ai;
    t.tj[tid] = aj;
    t.tm0[tid] = (fl & 1u) ? c : 0.0f;
    t.tm1[tid] = (fl & 2u) ? c : 0.0f;
    t.td[tid] = pair_d[r];
    atomicMax(s_amax, aj + 1);
  }
}

// rows [a_lo, a_lo+na) of src[N][F] -> dst[na][F+1]
template <int F>
__device__ __forceinline__ void stage_atoms(const float* __restrict__ src, int a_lo, int na, float* dst, int tid) {
  constexpr int Q = F / 4;
  const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)a_lo * F);
#pragma unroll 4
  for (int i = tid; i < na * Q; i += 256) {
    const int a = i / Q, q = i - a * Q;
    const float4 v = s4[i];
    float* d = dst + a * (F + 1) + 4 * q;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

// ------------------------------------------------------------------------------------------------ kernel A
// Software pipelined across tiles (one wave per SIMD cannot rely on other waves to hide memory latency): the row
// descriptors of tile t+1 are requested before the dt GEMM of tile t, and its atom window is requested (global ->
// registers) before the dW1 phase of tile t and written to LDS after it, so both global round trips run under MFMA
// work.  The atom window of tile t is only read by the dt GEMM, which is why the single LDS stage can be refilled
// while the dW1 phase of the same tile still runs.
struct RowDesc {
  int ai, aj;
  float m0, m1, d;
};
__device__ __forceinline__ RowDesc fetch_row_desc(const float* __restrict__ pair_d, const float* __restrict__ pair_c,
                                                  const uint8_t* __restrict__ pair_flag,
                                                  const int32_t* __restrict__ pair_i,
                                                  const int32_t* __restrict__ pair_j, int P, int row) {
  RowDesc r;
  const bool ok = row < P;
  const int q = ok ? row : P - 1;
  r.ai = pair_i[q];
  r.aj = pair_j[q];
  const unsigned fl = ok ? pair_flag[q] : 0u;
  const float c = pair_c[q];
  r.m0 = (fl & 1u) ? c : 0.0f;
  r.m1 = (fl & 2u) ? c : 0.0f;
  r.d = pair_d[q];
  return r;
}

template <int NC, int ABL = 0>
__global__ __launch_bounds__(256) void k_filter_bwd_a(const float* __restrict__ pair_d, const float* __restrict__ pair_c,
                                                      const uint8_t* __restrict__ pair_flag,
                                                      const int32_t* __restrict__ pair_i,
                                                      const int32_t* __restrict__ pair_j, int P, GeosslFilterWeights w,
                                                      GeosslFilterGradIn g, int G, const float* __restrict__ offset,
                                                      float coeff, const float* __restrict__ T,
                                                      float* __restrict__ partial_w1, float* __restrict__ partial_b1) {
  constexpr int F = 32 * NC, AS = F + 1, Q = F / 4;
  constexpr int NPRE = (2 * ATOM_CAP * Q + 255) / 256;  // float4 per thread to move one atom window (x and dagg)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W2s = smem;                      // [n][k] = w2[n][k]: B operand of dt = dO W2 (contraction over n)
  float* xs = W2s + F * F;                // [ATOM_CAP][F+1]
  float* ds = xs + ATOM_CAP * AS;         // [ATOM_CAP][F+1]
  float* offs = ds + ATOM_CAP * AS;       // [64]
  float* tabf = offs + 64;                // tm0, tm1, td: 3 x [128]
  int* tabi = reinterpret_cast<int*>(tabf + 3 * 128);  // ti, tj: 2 x [128], then s_amax[2]
  TileDesc td{tabi, tabi + 128, tabf, tabf + 128, tabf + 256};
  int* s_amax = tabi + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int l = blockIdx.y;
  load_weight_N(w.w2[l], F, F, F, W2s, F, F, F, tid, 256);
  if (tid < 64) offs[tid] = tid < G ? offset[tid] : 0.0f;
  if (tid < 2) s_amax[tid] = 0;
  const float* __restrict__ x = g.x[l];
  const float* __restrict__ dagg = g.dagg[l];
  const size_t lbase = (size_t)l * P;
  f32x16 accw[NC][2];  // dW1 partial of this wave: [hidden block][gaussian block]
  float bsum[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    bsum[c] = 0.0f;
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
      for (int r = 0; r < 16; ++r) accw[c][g2][r] = 0.0f;
  }
  const int ntiles = (P + 127) / 128;
  const int per = (ntiles + gridDim.x - 1) / gridDim.x;  // contiguous tile range per block (atom reuse in L2)
  const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
  if (t_begin >= t_end) {
    // nothing to do for this block: still publish zero partials below
  }
  // ---- prologue: descriptors and atom window of the first tile, synchronously
  RowDesc pd{0, 0, 0.0f, 0.0f, 0.0f};
  if (t_begin < t_end && tid < 128) pd = fetch_row_desc(pair_d, pair_c, pair_flag, pair_i, pair_j, P, t_begin * 128 + tid);
  __syncthreads();
  int a_lo = 0, na = 0;
  bool staged = false;
  if (t_begin < t_end) {
    if (tid < 128) {
      td.ti[tid] = pd.ai; td.tj[tid] = pd.aj; td.tm0[tid] = pd.m0; td.tm1[tid] = pd.m1; td.td[tid] = pd.d;
      atomicMax(&s_amax[t_begin & 1], pd.aj + 1);
    }
    __syncthreads();
    a_lo = td.ti[0];
    na = s_amax[t_begin & 1] - a_lo;
    staged = na <= ATOM_CAP;
    if (staged) {
      stage_atoms<F>(x, a_lo, na, xs, tid);
      stage_atoms<F>(dagg, a_lo, na, ds, tid);
    }
    if (t_begin + 1 < t_end && tid < 128)
      pd = fetch_row_desc(pair_d, pair_c, pair_flag, pair_i, pair_j, P, (t_begin + 1) * 128 + tid);
  }
  for (int t = t_begin; t < t_end; ++t) {
    const int r0b = t * 128;
    __syncthreads();  // tables(t) and atoms(t) are in LDS
    if (tid == 0) s_amax[(t + 1) & 1] = 0;  // slot of the next tile (last read two tiles ago)
    const int myr = wave * 32 + j;
    const int r0 = r0b + wave * 32;
    const int gi = td.ti[myr], gj = td.tj[myr];
    const float m0 = td.tm0[myr], m1 = td.tm1[myr];
    float dd16[16];  // distances of the rows this lane contracts over in the dW1 phase
#pragma unroll
    for (int s = 0; s < 16; ++s) dd16[s] = td.td[wave * 32 + c_row(s, lane)];
    // saved hidden activation of this wave's 32 rows in C layout (requested now, used after the GEMM)
    float tc[NC][16];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(r0 + c_row(r, lane), P - 1);
        tc[c][r] = ABL == 3 ? 0.5f : T[(lbase + row) * F + 32 * c + j];
      }
    f32x16 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = ABL == 1 ? 0.001f * r : 0.0f;
    if (ABL == 1) {
    } else if (staged) {
      const float* di = ds + (gi - a_lo) * AS;
      const float* dj = ds + (gj - a_lo) * AS;
      const float* xi = xs + (gi - a_lo) * AS;
      const float* xj = xs + (gj - a_lo) * AS;
      auto afrag = [&](int n) { return m0 * (di[n] * xj[n]) + m1 * (dj[n] * xi[n]); };
      float a_cur = afrag(kh), b_cur[NC], b_nxt[NC];
      {
        const float* bp = W2s + kh * F + j;
#pragma unroll
        for (int c = 0; c < NC; ++c) b_cur[c] = bp[32 * c];
      }
      constexpr int K2 = F / 2;
#pragma unroll 4
      for (int kk = 0; kk < K2; ++kk) {
        const int kn = 2 * min(kk + 1, K2 - 1) + kh;
        const float a_nxt = afrag(kn);
        const float* bp = W2s + kn * F + j;
#pragma unroll
        for (int c = 0; c < NC; ++c) b_nxt[c] = bp[32 * c];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[c], acc[c], 0, 0, 0);
        a_cur = a_nxt;
#pragma unroll
        for (int c = 0; c < NC; ++c) b_cur[c] = b_nxt[c];
      }
    } else {
      // tile spans more atoms than fit in LDS (many tiny molecules): operands straight from global memory
      const float* di = dagg + (size_t)gi * F;
      const float* dj = dagg + (size_t)gj * F;
      const float* xi = x + (size_t)gi * F;
      const float* xj = x + (size_t)gj * F;
      for (int kk = 0; kk < F / 2; ++kk) {
        const int n = 2 * kk + kh;
        const float a = m0 * (di[n] * xj[n]) + m1 * (dj[n] * xi[n]);
        const float* bp = W2s + n * F + j;
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[32 * c], acc[c], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave is done with tables(t) and atoms(t)
    // ---- publish the descriptors of tile t+1 and request its atom window (global -> registers)
    const bool more = t + 1 < t_end;
    float4 pre[NPRE];
    int a_lo_n = 0, na_n = 0;
    bool staged_n = false;
    if (more) {
      if (tid < 128) {
        td.ti[tid] = pd.ai; td.tj[tid] = pd.aj; td.tm0[tid] = pd.m0; td.tm1[tid] = pd.m1; td.td[tid] = pd.d;
        atomicMax(&s_amax[(t + 1) & 1], pd.aj + 1);
      }
      __syncthreads();
      a_lo_n = td.ti[0];
      na_n = s_amax[(t + 1) & 1] - a_lo_n;
      staged_n = na_n <= ATOM_CAP;
      if (staged_n && ABL != 4) {
        const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)a_lo_n * F);
        const float4* d4 = reinterpret_cast<const float4*>(dagg + (size_t)a_lo_n * F);
        const int nq = na_n * Q;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
          const int i = tid + 256 * u;
          pre[u] = i < nq ? x4[i] : (i < 2 * nq ? d4[i - nq] : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        }
      }
      if (t + 2 < t_end && tid < 128)
        pd = fetch_row_desc(pair_d, pair_c, pair_flag, pair_i, pair_j, P, (t + 2) * 128 + tid);
    }
    // ---- dU = dt * ssp'(pre) in place (C layout: lane = hidden unit 32c+j, register = pair row)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[c][r] *= dssp_from_out(tc[c][r]);
        bsum[c] += acc[c][r];
      }
    // dW1[k][g] += sum_rows dU[row][k] * rbf(d_row)[g]: k-step s contracts over the two rows held in register s
    // (row c_row(s, lane) for each half-wave); A fragment = acc[c][s] as is.
    if (ABL != 2)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float bv[2];
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        const int gg = 32 * g2 + j;
        const float diff = dd16[s] - offs[gg];
        bv[g2] = gg < G ? __expf(coeff * (diff * diff)) : 0.0f;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
          accw[c][g2] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[c][s], bv[g2], accw[c][g2], 0, 0, 0);
    }
    // ---- the atom window of tile t+1 lands in LDS (its previous content was last read before the barrier above)
    if (more && staged_n && ABL != 4) {
      const int nq = na_n * Q;
#pragma unroll
      for (int u = 0; u < NPRE; ++u) {
        const int i = tid + 256 * u;
        if (i < 2 * nq) {
          const int ii = i < nq ? i : i - nq;
          const int a = ii / Q, q4 = ii - a * Q;
          float* d = (i < nq ? xs : ds) + a * AS + 4 * q4;
          d[0] = pre[u].x; d[1] = pre[u].y; d[2] = pre[u].z; d[3] = pre[u].w;
        }
      }
    }
    a_lo = a_lo_n;
    na = na_n;
    staged = staged_n;
  }
  // one partial per wave: [F][G] and [F]
  const size_t pw = ((size_t)l * gridDim.x + blockIdx.x) * 4 + wave;
  float* Pw = partial_w1 + pw * F * G;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      const int gg = 32 * g2 + j;
      if (gg < G) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Pw[(size_t)(32 * c + c_row(r, lane)) * G + gg] = accw[c][g2][r];
      }
    }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float s = bsum[c] + __shfl_xor(bsum[c], 32, 64);
    if (kh == 0) partial_b1[pw * F + 32 * c + j] = s;
  }
  (void)na;
}

// ------------------------------------------------------------------------------------------------ kernel B
// 64-row tiles and a 40-atom window keep the block at ~75 KB of LDS: two blocks per CU, so one block's staging
// phase overlaps the other's MFMA phase.
constexpr int TRB = 64;       // pair rows per tile
constexpr int ATOM_CAP_B = 40;

template <int NC>
__global__ __launch_bounds__(256, 2) void k_filter_bwd_b(const float* __restrict__ pair_d,
                                                         const float* __restrict__ pair_c,
                                                         const uint8_t* __restrict__ pair_flag,
                                                         const int32_t* __restrict__ pair_i,
                                                         const int32_t* __restrict__ pair_j, int P,
                                                         GeosslFilterGradIn g, const float* __restrict__ T,
                                                         float* __restrict__ partial_w2,
                                                         float* __restrict__ partial_b2) {
  constexpr int F = 32 * NC, AS = F + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ts = smem;                       // [TRB rows][F] saved hidden activations of the tile (B operand)
  float* xs = Ts + TRB * F;               // [ATOM_CAP_B][F+1]
  float* ds = xs + ATOM_CAP_B * AS;       // [ATOM_CAP_B][F+1]
  float* tabf = ds + ATOM_CAP_B * AS;     // tm0, tm1, td
  int* tabi = reinterpret_cast<int*>(tabf + 3 * TRB);
  TileDesc td{tabi, tabi + TRB, tabf, tabf + TRB, tabf + 2 * TRB};
  int* s_amax = tabi + 2 * TRB;
  // packed per-row descriptor {LDS offset of atom i, of atom j, C*flag0, C*flag1}: one 16-byte broadcast read per
  // k-step instead of four 4-byte ones (the MFMA loop of this kernel is LDS-issue bound)
  int4* desc4 = reinterpret_cast<int4*>(smem + ((TRB * F + 2 * ATOM_CAP_B * AS + 5 * TRB + 4 + 3) & ~3));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int l = blockIdx.y;
  const float* __restrict__ x = g.x[l];
  const float* __restrict__ dagg = g.dagg[l];
  const size_t lbase = (size_t)l * P;
  const int ncol = 32 * wave + j;  // the dW2 row (= filter output channel n) this lane feeds as A operand
  f32x16 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
  float bsum = 0.0f;
  const int ntiles = (P + TRB - 1) / TRB;
  const int per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
  for (int t = t_begin; t < t_end; ++t) {
    const int r0b = t * TRB;
    __syncthreads();
    if (tid == 0) *s_amax = 0;
    __syncthreads();
    load_tile_desc(pair_d, pair_c, pair_flag, pair_i, pair_j, P, r0b, td, s_amax, tid, TRB);
    {  // T rows of the tile, row-major, 16-byte loads (rows past P: zero)
      const float4* t4 = reinterpret_cast<const float4*>(T + (lbase + r0b) * F);
      constexpr int Q = F / 4;
#pragma unroll
      for (int i = tid; i < TRB * Q; i += 256) {
        const int r = i / Q;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (r0b + r < P) v = t4[i];
        reinterpret_cast<float4*>(Ts)[i] = v;
      }
    }
    __syncthreads();
    const int a_lo = td.ti[0];
    const int na = *s_amax - a_lo;
    const bool staged = na <= ATOM_CAP_B;
    if (staged) {
      stage_atoms<F>(x, a_lo, na, xs, tid);
      stage_atoms<F>(dagg, a_lo, na, ds, tid);
      if (tid < TRB)
        desc4[tid] = make_int4((td.ti[tid] - a_lo) * AS, (td.tj[tid] - a_lo) * AS, __float_as_int(td.tm0[tid]),
                               __float_as_int(td.tm1[tid]));
    }
    __syncthreads();
    if (ncol < F) {
      if (staged) {
        // A fragment of k-step kk: dO[row = 2kk+kh][ncol]; the row's descriptor is uniform over the half-wave
        auto afrag = [&](int row) {
          const int4 q = desc4[row];
          const int oi = q.x + ncol, oj = q.y + ncol;
          return __int_as_float(q.z) * (ds[oi] * xs[oj]) + __int_as_float(q.w) * (ds[oj] * xs[oi]);
        };
        float a_cur = afrag(kh), b_cur[NC], b_nxt[NC];
        {
          const float* bp = Ts + kh * F + j;
#pragma unroll
          for (int c = 0; c < NC; ++c) b_cur[c] = bp[32 * c];
        }
#pragma unroll 4
        for (int kk = 0; kk < TRB / 2; ++kk) {
          const int rn = 2 * min(kk + 1, TRB / 2 - 1) + kh;
          const float a_nxt = afrag(rn);
          const float* bp = Ts + rn * F + j;
#pragma unroll
          for (int c = 0; c < NC; ++c) b_nxt[c] = bp[32 * c];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[c], acc[c], 0, 0, 0);
          bsum += a_cur;
          a_cur = a_nxt;
#pragma unroll
          for (int c = 0; c < NC; ++c) b_cur[c] = b_nxt[c];
        }
      } else {
        // window larger than the LDS stage (many tiny molecules in one tile): operands straight from global
        for (int kk = 0; kk < TRB / 2; ++kk) {
          const int row = 2 * kk + kh;
          const size_t oi = (size_t)td.ti[row] * F + ncol, oj = (size_t)td.tj[row] * F + ncol;
          const float a = td.tm0[row] * (dagg[oi] * x[oj]) + td.tm1[row] * (dagg[oj] * x[oi]);
          const float* bp = Ts + row * F + j;
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[32 * c], acc[c], 0, 0, 0);
          bsum += a;
        }
      }
    }
  }
  // one partial per block: wave w holds rows 32w..32w+31 of dW2 (C layout: lane = column k, register = row n)
  const size_t pb = (size_t)l * gridDim.x + blockIdx.x;
  float* Pw = partial_w2 + pb * F * F;
  if (32 * wave < F) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) Pw[(size_t)(32 * wave + c_row(r, lane)) * F + 32 * c + j] = acc[c][r];
    const float s = bsum + __shfl_xor(bsum, 32, 64);
    if (kh == 0) partial_b2[pb * F + ncol] = s;
  }
}

inline int blocks_per_layer(int L, int ntiles) {
  int b = 256 / (L > 0 ? L : 1);
  if (b < 1) b = 1;
  if (b > ntiles) b = ntiles;
  return b;
}

}  // namespace

extern "C" int64_t geossl_cfconv_filter_bwd_workspace_floats(int64_t P, int L, int F, int G) {
  const int ntiles = (int)((P + 127) / 128);
  const int64_t nb = blocks_per_layer(L, ntiles);
  return (int64_t)L * nb * (4 * ((int64_t)F * G + F) + 2 * ((int64_t)F * F + F));
}

extern "C" int geossl_cfconv_filter_bwd(const float* pair_d, const float* pair_c, const uint8_t* pair_flag,
                                        const int32_t* pair_i, const int32_t* pair_j, int64_t P,
                                        const GeosslFilterWeights* w, const GeosslFilterGradIn* g, int L, int F, int G,
                                        const float* offset, float coeff, const float* T,
                                        const GeosslFilterGradOut* out, float* workspace, int accumulate,
                                        hipStream_t stream) {
  if (P <= 0 || L <= 0) return 0;
  if (L > GEOSSL_MAX_L || (F != 32 && F != 64 && F != 128) || G > 64) return (int)hipErrorInvalidValue;
  const int ntiles = (int)((P + 127) / 128);
  const int nb = blocks_per_layer(L, ntiles);
  dim3 grid(nb, L);
  static const int abl = getenv("GEOSSL_ABLATE") ? atoi(getenv("GEOSSL_ABLATE")) : 0;
  float* pw1 = workspace;                                   // [L][nb*4][F][G]
  float* pb1 = pw1 + (size_t)L * nb * 4 * F * G;            // [L][nb*4][F]
  const int nbb = 2 * nb;                                   // kernel B runs two (smaller) blocks per CU
  float* pw2 = pb1 + (size_t)L * nb * 4 * F;                // [L][nbb][F][F]
  float* pb2 = pw2 + (size_t)L * nbb * F * F;               // [L][nbb][F]
  const size_t tab = (3 * 128 + 2 * 128 + 4) * sizeof(float);
  const size_t atoms = (size_t)2 * ATOM_CAP * (F + 1) * sizeof(float);
  const size_t lds_a = (size_t)F * F * sizeof(float) + atoms + 64 * sizeof(float) + tab;
  const size_t lds_b = (size_t)TRB * F * sizeof(float) + (size_t)2 * ATOM_CAP_B * (F + 1) * sizeof(float) +
                       (5 * TRB + 8) * sizeof(float) + (size_t)TRB * 16;
#define LAUNCH(NCV)                                                                                                   \
  do {                                                                                                                \
    allow_big_lds(&k_filter_bwd_a<NCV>);                                                                              \
    allow_big_lds(&k_filter_bwd_b<NCV>);                                                                              \
    if (abl == 1) { allow_big_lds(&k_filter_bwd_a<NCV, 1>); hipLaunchKernelGGL((k_filter_bwd_a<NCV, 1>), grid, dim3(256), lds_a, stream, pair_d, pair_c, pair_flag, pair_i, pair_j, (int)P, *w, *g, G, offset, coeff, T, pw1, pb1); } \
    else if (abl == 2) { allow_big_lds(&k_filter_bwd_a<NCV, 2>); hipLaunchKernelGGL((k_filter_bwd_a<NCV, 2>), grid, dim3(256), lds_a, stream, pair_d, pair_c, pair_flag, pair_i, pair_j, (int)P, *w, *g, G, offset, coeff, T, pw1, pb1); } \
    else if (abl == 3) { allow_big_lds(&k_filter_bwd_a<NCV, 3>); hipLaunchKernelGGL((k_filter_bwd_a<NCV, 3>), grid, dim3(256), lds_a, stream, pair_d, pair_c, pair_flag, pair_i, pair_j, (int)P, *w, *g, G, offset, coeff, T, pw1, pb1); } \
    else if (abl == 4) { allow_big_lds(&k_filter_bwd_a<NCV, 4>); hipLaunchKernelGGL((k_filter_bwd_a<NCV, 4>), grid, dim3(256), lds_a, stream, pair_d, pair_c, pair_flag, pair_i, pair_j, (int)P, *w, *g, G, offset, coeff, T, pw1, pb1); } \
    else if (abl == 9) { } \
    else hipLaunchKernelGGL((k_filter_bwd_a<NCV>), grid, dim3(256), lds_a, stream, pair_d, pair_c, pair_flag, pair_i,      \
                       pair_j, (int)P, *w, *g, G, offset, coeff, T, pw1, pb1);                                        \
    hipLaunchKernelGGL((k_filter_bwd_b<NCV>), dim3(nbb, L), dim3(256), lds_b, stream, pair_d, pair_c, pair_flag,     \
                       pair_i, pair_j, (int)P, *g, T, pw2, pb2);                                                              \
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
  reduce(out->dw1, pw1, nb * 4, F * G, G);
  reduce(out->db1, pb1, nb * 4, F, F);
  reduce(out->dw2, pw2, nbb, F * F, F);
  reduce(out->db2, pb2, nbb, F, F);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
