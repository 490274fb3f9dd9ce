// Continuous-filter convolution of ONE interaction block in one launch: filter network + neighbour aggregation
// (CFConv.forward, schnet.py:185-195, with InteractionBlock.mlp :141-145 and GaussianSmearing :205-207 folded in):
//     agg[i] = sum_{j in N(i)} x[j] * Wf(d_ij),     Wf(d) = ( ssp( rbf(d) A1^T + b1 ) A2^T + b2 ) * C(d).
// The filter rows never leave the chip: a pair slot's row of Wf is produced in MFMA accumulators, multiplied with
// the two atom rows it connects and added onto the molecule's atom rows in LDS.  Against geossl_cfconv_filter_fwd +
// geossl_cfconv_aggregate this removes the store of Wf and its re-read by the aggregation (160 MB each way per block
// at the bench size); the hidden activation t (saved for the backward) is still written.
//
// Work split (F = 128): a TEAM of four waves (one per SIMD) owns one molecule at a time and walks its pair slots in
// tiles of 32 rows.  Wave m holds, in registers for the whole launch, the fragments of A1's hidden block m (first GEMM,
// transposed: t^T = A1 rbf^T, pair rows on the lanes) and of A2's OUTPUT block m (second GEMM with the operands
// SWAPPED: Wf = t A2^T with the pair rows on the MFMA M axis, so that the lane is the output feature and the registers
// are the pair rows).  What the waves exchange per tile through LDS: the Gaussian fragments (each wave evaluates one
// k-step of them, every exp once per team) and the split t fragments (each wave publishes its 32 hidden units; all four
// read all of them as A operands) - 24 KB per tile where the wave-owns-the-tile kernel reads 96 KB of weights.
// In the swapped C layout lane (j, kh) holds, for its feature 32m + j, the 16 pair rows (r&3) + 8(r>>2) + 4kh: the
// scatter is plain vector code - two LDS reads (the partner rows of x), two multiplies and two ds_add_f32 per slot -
// onto an accumulator tile [atom][32 features] that belongs to this wave and this half alone (the two halves own
// separate copies, summed when the molecule is done), so every sum is formed by one instruction stream in program
// order: results are bit-reproducible.  The products are the reference's (fl(fl(w*C)*x)); only the ORDER of a target's
// sum differs from a sequential index_add (tests: 1e-6 of the tensor scale against the unfused kernels).
//
// Pipeline over the team's tile sequence, one block barrier (LDS only) per tile:
//     phase k:   second GEMM + scatter of tile k   |  first GEMM + ssp + publish of tile k+1  |  Gaussians of tile k+2
// Everything in a phase is unconditional (dummy tiles with dead rows at both ends of the sequence; LDS starts zeroed),
// so the body is one basic block apart from the two molecule-boundary branches.
//
// FROM_T = true is the same kernel fed from a saved t instead of the first GEMM: the transposed aggregation of the
// backward pass (dx = A^T(Wf) dagg) without a stored Wf.
#include "common.h"
#include "geossl_hip.h"
#include "split.h"

using namespace geossl;

namespace {

constexpr int FU_F = 128, FU_NMB = 4, FU_K1S = 4, FU_K2S = 8;
constexpr int FU_W2_Q = FU_NMB * FU_K2S * 2 * 64;  // u32x4 units of the A2 fragments  [mb][ks][piece][lane]
constexpr int FU_W1_Q = FU_NMB * FU_K1S * 2 * 64;  // u32x4 units of the A1 fragments
constexpr int FU_TAIL = 128 + 128 + 64 + 16;       // b1, b2, Gaussian centres (padded with 0), {inv1, k2, st}
constexpr size_t FU_IMAGE_BYTES = (size_t)(FU_W2_Q + FU_W1_Q) * 16 + FU_TAIL * 4;

// ---- operand image of every layer, one launch per pass: fragments on two fp16 pieces (split.h), power-of-two scales:
// A1, A2 by their largest magnitude, the Gaussians by 2^14, t by a BOUND of its magnitude that needs no look at the
// data: |ssp(u)| <= max(|u|, log 2) and |u_h| <= |b1_h| + sum_g |A1_hg| because 0 <= rbf <= 1 (the bound costs a few
// of the 17 bits of headroom the two-piece format has above its 22 significant ones; one scale for all rows keeps the
// undo factor out of the per-row epilogue).
__global__ __launch_bounds__(512) void k_fused_prepare(GeosslFilterWeights w, int G, const float* __restrict__ offset,
                                                       uint8_t* __restrict__ images) {
  constexpr int F = FU_F, NMB = FU_NMB, K1S = FU_K1S, K2S = FU_K2S;
  __shared__ float red[64];
  const int l = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  u32x4* W2f = reinterpret_cast<u32x4*>(images + (size_t)l * FU_IMAGE_BYTES);
  u32x4* W1f = W2f + FU_W2_Q;
  float* tail = reinterpret_cast<float*>(W1f + FU_W1_Q);
  const float* __restrict__ w1 = w.w1[l];
  const float* __restrict__ w2 = w.w2[l];
  const float* __restrict__ b1 = w.b1[l];
  float m1 = 0.0f, m2 = 0.0f, bu = 0.0f;
  for (int i = tid; i < F * G; i += 512) m1 = fmaxf(m1, fabsf(w1[i]));
  for (int i = tid; i < F * F; i += 512) m2 = fmaxf(m2, fabsf(w2[i]));
  if (tid < F) {
    float s = fabsf(b1[tid]);
    for (int g = 0; g < G; ++g) s += fabsf(w1[(size_t)tid * G + g]);
    bu = s;
  }
  m1 = wave_max(m1);
  m2 = wave_max(m2);
  bu = wave_max(bu);
  if (lane == 0) {
    red[wave] = m1;
    red[16 + wave] = m2;
    red[32 + wave] = bu;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    m1 = fmaxf(m1, red[i]);
    m2 = fmaxf(m2, red[16 + i]);
    bu = fmaxf(bu, red[32 + i]);
  }
  int e1, e2, et;
  const float s1 = pow2_scale_to_2p14(m1, e1), s2 = pow2_scale_to_2p14(m2, e2);
  const float st = pow2_scale_to_2p14(fmaxf(bu, 1.0f), et);
  for (int i = tid; i < NMB * K2S * 64; i += 512) {
    const int ln = i & 63, ks = (i >> 6) % K2S, mb = i / (64 * K2S);
    // contraction-index permutation kperm (split.h): elements 0..3 <- features 4kh.., 4..7 <- features 8+4kh..
    const float* row = w2 + (size_t)(32 * mb + (ln & 31)) * F + 16 * ks + 4 * (ln >> 5);
    const float4 lo = *reinterpret_cast<const float4*>(row), hi = *reinterpret_cast<const float4*>(row + 8);
    const float v[8] = {lo.x * s2, lo.y * s2, lo.z * s2, lo.w * s2, hi.x * s2, hi.y * s2, hi.z * s2, hi.w * s2};
    const Frag2 f = split8h(v);
    u32x4* dst = W2f + ((size_t)(mb * K2S + ks) * 2) * 64 + ln;
    dst[0] = f.h;
    dst[64] = f.l;
  }
  for (int i = tid; i < NMB * K1S * 64; i += 512) {
    const int ln = i & 63, ks = (i >> 6) % K1S, mb = i / (64 * K1S);
    const float* row = w1 + (size_t)(32 * mb + (ln & 31)) * G;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = 16 * ks + 8 * (ln >> 5) + e;
      v[e] = g < G ? row[g] * s1 : 0.0f;
    }
    const Frag2 f = split8h(v);
    u32x4* dst = W1f + ((size_t)(mb * K1S + ks) * 2) * 64 + ln;
    dst[0] = f.h;
    dst[64] = f.l;
  }
  for (int i = tid; i < F; i += 512) {
    tail[i] = b1[i];
    tail[128 + i] = w.b2[l][i];
  }
  for (int i = tid; i < 64; i += 512) tail[256 + i] = i < G ? offset[i] : 0.0f;
  if (tid == 0) {
    tail[320] = 1.0f / (s1 * 16384.0f);  // undoes the scales of A1 and of the Gaussians (powers of two: exact)
    tail[321] = 1.0f / (s2 * st);        // undoes the scales of A2 and of t
    tail[322] = st;
  }
}

struct FusedArgs {
  const float* pair_d;
  const float* pair_c;
  const uint8_t* pair_flag;
  const int32_t* pair_i;
  const int32_t* pair_j;
  const int32_t* mol_ptr;
  const int32_t* pair_ptr;
  const int32_t* order;
  int B;
  const uint8_t* image;
  float coeff;
  const float* x;
  float* out;
  float* T;   // FROM_T: read; else written when not null
  float* Wf;  // written when STORE_WF
  int swap;
};

struct Cur {  // a team's position in its tile sequence (wave-uniform)
  int ord, tile, nt, a0, n, base, np, valid;
};

template <int MAXN, bool FROM_T, bool STORE_WF>
__global__ __launch_bounds__(256, MAXN <= 18 ? 2 : 1) void k_cfconv_fused(FusedArgs A) {
  constexpr int F = FU_F, K1S = FU_K1S, K2S = FU_K2S;
  constexpr int WREG = 640 + 3 * MAXN * 128;  // bytes of a wave's private region: row tables, x rows, two accumulator tiles
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u32x4* tfr = reinterpret_cast<u32x4*>(smem);         // [2][K2S][2][64] split t fragments of a tile
  u32x4* rfr = tfr + 2 * K2S * 2 * 64;                 // [2][K1S][2][64] split Gaussian fragments of a tile
  float* shc = reinterpret_cast<float*>(rfr + 2 * K1S * 2 * 64);  // [128] b1, [64] Gaussian centres
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
  uint8_t* wbase = reinterpret_cast<uint8_t*>(shc + 192) + wave * WREG;
  float* tab = reinterpret_cast<float*>(wbase);        // [5][32]: c1, c2, cw, byte offset of atom i, of atom j
  float* xs = reinterpret_cast<float*>(wbase + 640);   // [MAXN][32] the molecule's rows of x, this wave's 32 features
  float* acc = xs + MAXN * 32;                         // [2 halves][MAXN][32]
  {
    uint32_t* z = reinterpret_cast<uint32_t*>(smem);
    constexpr int words = (2 * K2S * 2 * 64 + 2 * K1S * 2 * 64) * 4 + 192 + 4 * WREG / 4;
    for (int i = tid; i < words; i += 256) z[i] = 0u;
  }
  const u32x4* __restrict__ W2f = reinterpret_cast<const u32x4*>(A.image);
  const u32x4* __restrict__ W1f = W2f + FU_W2_Q;
  const float* __restrict__ tail = reinterpret_cast<const float*>(W1f + FU_W1_Q);
  // ---- stationary operands of this wave
  u32x4 w2h[K2S], w2l[K2S];
#pragma unroll
  for (int ks = 0; ks < K2S; ++ks) {
    w2h[ks] = W2f[((size_t)(wave * K2S + ks) * 2) * 64 + lane];
    w2l[ks] = W2f[((size_t)(wave * K2S + ks) * 2 + 1) * 64 + lane];
  }
  u32x4 w1h[FROM_T ? 1 : K1S], w1l[FROM_T ? 1 : K1S];
  if constexpr (!FROM_T) {
#pragma unroll
    for (int ks = 0; ks < K1S; ++ks) {
      w1h[ks] = W1f[((size_t)(wave * K1S + ks) * 2) * 64 + lane];
      w1l[ks] = W1f[((size_t)(wave * K1S + ks) * 2 + 1) * 64 + lane];
    }
  }
  const float b2f = tail[128 + 32 * wave + j];
  const float inv1 = tail[320], k2 = tail[321], st = tail[322];
  __syncthreads();  // the zero fill is done
  for (int i = tid; i < 192; i += 256) shc[i] = i < 128 ? tail[i] : tail[256 + (i - 128)];
  __syncthreads();

  const int stride = gridDim.x;
  const float* __restrict__ x = A.x;
  float* __restrict__ out = A.out;
  auto next_mol = [&](Cur& c) {  // the molecule at c.ord of the team's sequence, skipping those without pair slots
    for (;;) {
      if (c.ord >= A.B) {
        c = Cur{c.ord, 0, 1, 0, 0, 0, 0, 0};
        return;
      }
      const int mol = A.order != nullptr ? A.order[c.ord] : c.ord;
      const int a0 = A.mol_ptr[mol], n = A.mol_ptr[mol + 1] - a0;
      if (n >= 2) {
        const int np = n * (n - 1) / 2;
        c = Cur{c.ord, 0, (np + 31) / 32, a0, n, A.pair_ptr[mol], np, 1};
        return;
      }
      if (n == 1 && kh == 0) out[(size_t)a0 * F + 32 * wave + j] = 0.0f;  // an atom without partners (schnet.py:190)
      c.ord += stride;
    }
  };
  auto advance = [&](Cur& c) {
    if (c.valid && ++c.tile == c.nt) {
      c.ord += stride;
      next_mol(c);
    }
  };
  auto row_of = [&](const Cur& c) { return c.base + min(32 * c.tile + j, max(c.np - 1, 0)); };  // clamped: always valid
  auto live_of = [&](const Cur& c) { return c.valid && 32 * c.tile + j < c.np; };

  Cur S0{0, 0, 1, 0, 0, 0, 0, 0}, S1 = S0, S2{(int)blockIdx.x, 0, 1, 0, 0, 0, 0, 0};
  next_mol(S2);
  // per-row data of the tile in stage 0 (fetched while it was in stage 1) and requests in flight
  float cw0 = 0.0f;
  int fl0 = 0, pi0 = 0, pj0 = 0;
  bool live0 = false;
  float xr[(MAXN + 1) / 2];  // x rows of the molecule whose first tile is in stage 1 (two atoms per register)
#pragma unroll
  for (int i = 0; i < (MAXN + 1) / 2; ++i) xr[i] = 0.0f;
  float d2 = 0.0f;           // distance of this lane's row of the tile in stage 2
  f32x4 tn[FROM_T ? 4 : 1];  // FROM_T: the saved t pieces of the tile in stage 2
#pragma unroll
  for (int q = 0; q < (FROM_T ? 4 : 1); ++q) tn[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  auto request_stage2 = [&]() {
    const int row = row_of(S2);
    if constexpr (FROM_T) {
      const float* tr = A.T + (size_t)row * F + 32 * wave + 4 * kh;
#pragma unroll
      for (int q = 0; q < 4; ++q) tn[q] = *reinterpret_cast<const f32x4*>(tr + 8 * q);
    } else {
      d2 = A.pair_d[row];
    }
  };
  request_stage2();
  f32x4 t1[FROM_T ? 4 : 1];  // FROM_T: the pieces of the tile in stage 1 (zero while the pipeline fills: dummy tiles stay finite)
#pragma unroll
  for (int q = 0; q < (FROM_T ? 4 : 1); ++q) t1[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  int phase = 0;
  while (S0.valid | S1.valid | S2.valid) {
    const int pb = phase & 1;  // buffers written in this phase; the other pair was written in the last one
    // ---- stage 0, part 1: row tables of tile S0 (c1 / c2 = envelope where the directed edge exists, else 0)
    {
      const int f_in = A.swap ? 2 : 1, f_out = A.swap ? 1 : 2;
      const float c1 = (live0 && (fl0 & f_in)) ? cw0 : 0.0f;   // edge j -> i: message x[j] Wf to atom i
      const float c2 = (live0 && (fl0 & f_out)) ? cw0 : 0.0f;  // edge i -> j
      if (kh == 0) {
        tab[j] = c1;
        tab[32 + j] = c2;
        tab[64 + j] = live0 ? cw0 : 0.0f;
        reinterpret_cast<int*>(tab)[96 + j] = live0 ? pi0 * 128 : 0;
        reinterpret_cast<int*>(tab)[128 + j] = live0 ? pj0 * 128 : 0;
      }
    }
    if (S0.valid && S0.tile == 0) {  // a new molecule: its rows of x (this wave's feature block) into LDS
#pragma unroll
      for (int i = 0; i < (MAXN + 1) / 2; ++i)
        if (2 * i + kh < MAXN) xs[(2 * i + kh) * 32 + j] = xr[i];
    }
    // ---- requests that are consumed in the next phase: per-row data of tile S1, x rows of its molecule
    const int row1 = row_of(S1);
    const bool live1 = live_of(S1);
    const float cw1 = A.pair_c[row1];
    const int fl1 = A.pair_flag[row1], pi1 = A.pair_i[row1] - S1.a0, pj1 = A.pair_j[row1] - S1.a0;
    if (S1.valid && S1.tile == 0) {
#pragma unroll
      for (int i = 0; i < (MAXN + 1) / 2; ++i)
        xr[i] = x[(size_t)(S1.a0 + min(2 * i + kh, S1.n - 1)) * F + 32 * wave + j];
    }
    // ---- stage 0: second GEMM (operands swapped): acc2[r] = Wf'[pair row c_row(r)][feature 32 wave + j], scaled
    f32x16 acc2;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[e] = 0.0f;
    {
      const u32x4* src = tfr + ((size_t)((pb ^ 1) * K2S) * 2) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < K2S; ++ks) {
        const u32x4 th = src[(ks * 2) * 64], tl = src[(ks * 2 + 1) * 64];
        acc2 = mfma_f16(tl, w2h[ks], acc2);
        acc2 = mfma_f16(th, w2l[ks], acc2);
        acc2 = mfma_f16(th, w2h[ks], acc2);
      }
    }
    // ---- stage 2: this wave's k-step of the Gaussians of tile S2 (scaled by 2^14), published for the whole team
    if constexpr (!FROM_T) {
      float v[8];
      const f32x4 o0 = *reinterpret_cast<const f32x4*>(shc + 128 + 16 * wave + 8 * kh);
      const f32x4 o1 = *reinterpret_cast<const f32x4*>(shc + 128 + 16 * wave + 8 * kh + 4);
      const float o[8] = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float diff = d2 - o[e];
        v[e] = exp_neg(A.coeff * (diff * diff)) * 16384.0f;  // schnet.py:206-207 (padded centres meet zero weights)
      }
      const Frag2 f = split8h(v);
      u32x4* dst = rfr + ((size_t)(pb * K1S + wave) * 2) * 64 + lane;
      dst[0] = f.h;
      dst[64] = f.l;
    }
    // ---- stage 1: first GEMM (transposed) for this wave's hidden block: acc1[r] = u[hidden c_row(r)][pair row j]
    float tv[16];
    if constexpr (!FROM_T) {
      f32x16 acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[e] = 0.0f;
      const u32x4* src = rfr + ((size_t)((pb ^ 1) * K1S) * 2) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < K1S; ++ks) {
        const u32x4 bh = src[(ks * 2) * 64], bl = src[(ks * 2 + 1) * 64];
        acc1 = mfma_f16(w1l[ks], bh, acc1);
        acc1 = mfma_f16(w1h[ks], bl, acc1);
        acc1 = mfma_f16(w1h[ks], bh, acc1);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(shc + 32 * wave + 8 * q + 4 * kh);
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[4 * q + e] = ssp(fmaf(acc1[4 * q + e], inv1, b[e]));
      }
      if (A.T != nullptr && live1) {  // t, saved for the backward: row-major, 16 bytes per store
        float* tr = A.T + (size_t)row1 * F + 32 * wave + 4 * kh;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(tr + 8 * q) = f32x4{tv[4 * q], tv[4 * q + 1], tv[4 * q + 2], tv[4 * q + 3]};
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[4 * q + e] = t1[q][e];
    }
    // ---- stage 0, part 2: messages of tile S0 onto the molecule's atom rows.  Eight slots at a time: all table and x
    // reads of the batch first, then the arithmetic, then the sixteen ds_add_f32 - LDS operations of a wave complete in
    // order and the compiler keeps reads behind earlier atomics (they may alias), so a slot-by-slot loop pays two LDS
    // round trips per slot.
    {
      const uint8_t* xs_l = reinterpret_cast<const uint8_t*>(xs) + 4 * j;
      uint8_t* acc_l = reinterpret_cast<uint8_t*>(acc) + kh * (MAXN * 128) + 4 * j;
      float* wrow = nullptr;
      if constexpr (STORE_WF) wrow = A.Wf + (size_t)(S0.base + 32 * S0.tile) * F + 32 * wave + j;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        int ioa[8], joa[8];
        float c1a[8], c2a[8], cwa[8], xi[8], xj[8];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int q = 2 * hb + qq;
          const f32x4 c1 = *reinterpret_cast<const f32x4*>(tab + 8 * q + 4 * kh);
          const f32x4 c2 = *reinterpret_cast<const f32x4*>(tab + 32 + 8 * q + 4 * kh);
          const int4 io = *reinterpret_cast<const int4*>(tab + 96 + 8 * q + 4 * kh);
          const int4 jo = *reinterpret_cast<const int4*>(tab + 128 + 8 * q + 4 * kh);
          if constexpr (STORE_WF) {
            const f32x4 cw = *reinterpret_cast<const f32x4*>(tab + 64 + 8 * q + 4 * kh);
#pragma unroll
            for (int e = 0; e < 4; ++e) cwa[4 * qq + e] = cw[e];
          }
          const int iov[4] = {io.x, io.y, io.z, io.w}, jov[4] = {jo.x, jo.y, jo.z, jo.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            ioa[4 * qq + e] = iov[e];
            joa[4 * qq + e] = jov[e];
            c1a[4 * qq + e] = c1[e];
            c2a[4 * qq + e] = c2[e];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          xi[u] = *reinterpret_cast<const float*>(xs_l + ioa[u]);
          xj[u] = *reinterpret_cast<const float*>(xs_l + joa[u]);
        }
        float mi[8], mj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float wv = fmaf(acc2[8 * hb + u], k2, b2f);   // (A2 t + b2)[row][feature]
          mi[u] = mul_rn(mul_rn(wv, c1a[u]), xj[u]);           // x[j] * (Wf' * C), schnet.py:187,194
          mj[u] = mul_rn(mul_rn(wv, c2a[u]), xi[u]);
          if constexpr (STORE_WF) {
            const int rl = 8 * (2 * hb + (u >> 2)) + 4 * kh + (u & 3);
            if (S0.valid && 32 * S0.tile + rl < S0.np) wrow[(size_t)rl * F] = mul_rn(wv, cwa[u]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#if !defined(FU_PROBE) || FU_PROBE == 0
          __hip_atomic_fetch_add(reinterpret_cast<float*>(acc_l + ioa[u]), mi[u], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(reinterpret_cast<float*>(acc_l + joa[u]), mj[u], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
#elif FU_PROBE == 1  // timing probe: integer atomics (wrong sums)
          __hip_atomic_fetch_add(reinterpret_cast<int*>(acc_l + ioa[u]), __float_as_int(mi[u]), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(reinterpret_cast<int*>(acc_l + joa[u]), __float_as_int(mj[u]), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
#elif FU_PROBE == 2  // timing probe: plain stores (wrong sums)
          *reinterpret_cast<volatile float*>(acc_l + ioa[u]) = mi[u];
          *reinterpret_cast<volatile float*>(acc_l + joa[u]) = mj[u];
#elif FU_PROBE == 4  // fixed point: m * 2^sx = q_hi + q_lo 2^-24, two integer atomics (timing of the conversion; scale fixed)
          {
            const float ti = mi[u] * 1048576.0f, tj = mj[u] * 1048576.0f;
            const float fi = floorf(ti), fj = floorf(tj);
            __hip_atomic_fetch_add(reinterpret_cast<int*>(acc_l + ioa[u]), (int)fi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(reinterpret_cast<int*>(acc_l + joa[u]), (int)fj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(acc_l + MAXN * 128 * (1 - 2 * kh) + ioa[u]),
                                   (unsigned)((ti - fi) * 16777216.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(acc_l + MAXN * 128 * (1 - 2 * kh) + joa[u]),
                                   (unsigned)((tj - fj) * 16777216.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
#elif FU_PROBE == 3  // timing probe: no scatter at all
          asm volatile("" ::"v"(mi[u]), "v"(mj[u]));
#endif
        }
      }
    }
    // ---- stage 1, part 2: t scaled, split and published as the A fragments of the next phase's second GEMM
    // (registers 0..7 of the C layout are k-step 2 wave, registers 8..15 k-step 2 wave + 1: split.h, kperm)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tv[8 * half + e] * st;
      const Frag2 f = split8h(v);
      u32x4* dst = tfr + ((size_t)(pb * K2S + 2 * wave + half) * 2) * 64 + lane;
      dst[0] = f.h;
      dst[64] = f.l;
    }
    // ---- the molecule of tile S0 is complete: its atom rows out (the two halves' copies summed), accumulators cleared
    if (S0.valid && S0.tile == S0.nt - 1) {
      for (int a = kh; a < S0.n; a += 2) {
        float* p0 = acc + a * 32 + j;
        float* p1 = p0 + MAXN * 32;
        out[(size_t)(S0.a0 + a) * F + 32 * wave + j] = *p0 + *p1;
        *p0 = 0.0f;
        *p1 = 0.0f;
      }
    }
    lds_barrier();
    // ---- rotate
    S0 = S1;
    cw0 = cw1;
    fl0 = fl1;
    pi0 = pi1;
    pj0 = pj1;
    live0 = live1;
    S1 = S2;
    if constexpr (FROM_T) {
#pragma unroll
      for (int q = 0; q < 4; ++q) t1[q] = tn[q];
    }
    advance(S2);
    request_stage2();
    ++phase;
  }
}

}  // namespace

extern "C" int64_t geossl_cfconv_fused_image_bytes(int L, int F, int G) {
  if (F != FU_F || G > 64 || G < 1 || L < 1 || L > GEOSSL_MAX_L) return 0;
  return (int64_t)L * (int64_t)FU_IMAGE_BYTES;
}

extern "C" int geossl_cfconv_fused_prepare(const GeosslFilterWeights* w, int L, int F, int G, const float* offset,
                                           void* images, hipStream_t stream) {
  if (geossl_cfconv_fused_image_bytes(L, F, G) == 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_fused_prepare, dim3(L), dim3(512), 0, stream, *w, G, offset, static_cast<uint8_t*>(images));
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

namespace {
template <int MAXN, bool FROM_T, bool STORE_WF>
int launch_fused(const FusedArgs& a, hipStream_t stream) {
  constexpr size_t lds = (size_t)(2 * FU_K2S * 2 * 64 + 2 * FU_K1S * 2 * 64) * 16 + 192 * 4 + 4 * (640 + 3 * MAXN * 128);
  allow_big_lds(&k_cfconv_fused<MAXN, FROM_T, STORE_WF>);
  int blocks = MAXN <= 18 ? 512 : 256;  // persistent teams: two per CU while the LDS allows it
  if (blocks > a.B) blocks = a.B;
  hipLaunchKernelGGL((k_cfconv_fused<MAXN, FROM_T, STORE_WF>), dim3(blocks), dim3(256), lds, stream, a);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" int geossl_cfconv_fused(const float* pair_d, const float* pair_c, const uint8_t* pair_flag,
                                   const int32_t* pair_i, const int32_t* pair_j, const int32_t* mol_ptr,
                                   const int32_t* pair_ptr, const int32_t* order, int64_t B, int max_n, int F,
                                   const void* image, float coeff, const float* x, float* out, float* T, int from_t,
                                   float* Wf, int swap, hipStream_t stream) {
  if (B <= 0) return 0;
  if (F != FU_F || max_n > 34 || (from_t && T == nullptr)) return (int)hipErrorInvalidValue;
  FusedArgs a{pair_d, pair_c, pair_flag, pair_i, pair_j, mol_ptr, pair_ptr, order, (int)B,
              static_cast<const uint8_t*>(image), coeff, x, out, T, Wf, swap};
#define FUSED_GO(MAXN)                                                                   \
  do {                                                                                   \
    if (from_t) return launch_fused<MAXN, true, false>(a, stream);                       \
    if (Wf != nullptr) return launch_fused<MAXN, false, true>(a, stream);                \
    return launch_fused<MAXN, false, false>(a, stream);                                  \
  } while (0)
  if (max_n <= 18) FUSED_GO(18);
  FUSED_GO(34);
#undef FUSED_GO
}
