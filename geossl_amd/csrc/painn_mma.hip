// PaiNN interaction with the filter on the matrix pipe (PaiNNInteraction.forward, Geom3D/models/painn.py:54-64 and the
// filter of painn.py:241-245; F = 128).
//
// The vector kernels of painn.hip evaluate the filter W_e = (phi_e Wf^T + b) fcut_e of every edge and feature as 3 x R
// fused multiply-adds per thread: 60 of the ~72 instructions an (edge, feature) pair costs at R = 20.  Here it is one
// small GEMM per tile of 32 edges, W[e][c F + f] = sum_k phi'_e[k] Wf'[c F + f][k] with the augmented operands
//     phi'_e = [phi_e * fcut_e, fcut_e, 0 ...] (K = 32),      Wf' = [Wf | b | 0 ...]
// (bias and cutoff ride in the contraction: no epilogue) on two fp16 pieces per operand (split.h), and what is left for
// the vector unit is the message arithmetic itself (~13 instructions per edge and feature).
//
// Work split: a team of four waves owns a molecule; wave m holds the fragments of the filter columns of its 32 features
// (three channels) in registers and accumulates those 32 columns.  The C layout puts the FEATURE on the lane and 16 edge
// rows in the registers, so the sum over the edges of a target atom is a sum over registers - if the rows of a register
// group belong to one atom.  The edge list is therefore laid out in GROUPS of four rows (layout.py: PainnGroups): the
// edges of an atom in incidence order, padded to a multiple of four; a tile is eight consecutive groups of the molecule.
// Per tile and lane: the four rows of a group are summed in registers, the halves of the wave exchange the group sums
// (v_permlane32_swap) so that each half holds the eight group sums of the two output components it is responsible
// for, and the runs of groups with one target are summed in group order (wave-uniform control flow).  A run that ends an
// atom is written out; the one run a tile boundary can cut stays in its registers and goes on in the next tile.
// Fixed order everywhere: results are bit-reproducible; they differ from the per-atom kernels' only in the order of an
// atom's sum (tests: 2e-6 of the tensor scale).
//
// Built WITHOUT packed fp32 arithmetic (build.py: -fno-slp-vectorize for this file; tests/test_round6_cpu.py checks the
// code object): the packed form the compiler chose for the mu-zero instantiation, `v_pk_mul_f32 .. op_sel:[0,1]`,
// occasionally lost its low result in lanes 48-63 with two waves per SIMD.  Same run time.
#include "common.h"
#include "geossl_hip.h"
#include "split.h"
#include "tn.h"

using namespace geossl;

namespace {

constexpr int PM_F = 128;

struct PainnMmaArgs {
  const float* q;        // forward: q [N][F];            backward: dq_out [N][F]
  const float* mu;       // forward: mu [N][3][F];        backward: dmu_out [N][3][F]
  const float* xc;       // context features [N][3F]
  const float* mu_src;   // backward: mu [N][3][F] (rows of the source atoms)
  const int64_t* idx_other;  // forward: idx_j (source of an edge), backward: idx_i (target)
  const int32_t* row_edge;   // [4 * groups] edge of a row, -1 = padding
  const int32_t* grp_atom;   // [groups] atom the rows of a group belong to
  const int32_t* mol_grp;    // [B + 1] first group of a molecule
  const int32_t* mol_grp_end;  // [B] end of a molecule's groups, or NULL: mol_grp[m + 1] (the lists lie back to back)
  const float* phi;
  const float* fcut;
  const float* dir;
  const float* Wf;
  const float* bf;
  const int32_t* mol_ptr;
  int B, max_n, N;
  float* out0;   // forward: q_out;   backward: dxc [N][3F]
  float* out1;   // forward: mu_out;  backward: dmu_in [N][3][F]
  float* pw;     // backward: filter-gradient partials [blocks][3F][R]
  float* pb;     // backward: [blocks][3F]
};

// Wf' B fragments of this wave's 32 features, channel c, k-step ks: lane (col, kh) holds Wf'[c F + 32 m + col][16 ks + 8 kh + e]
template <int R>
__device__ __forceinline__ void load_filter_fragments(const float* __restrict__ Wf, const float* __restrict__ bf, int m,
                                                      int lane, float sW, u32x4 (&wh)[3][2], u32x4 (&wl)[3][2]) {
  const int col = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int rowi = c * PM_F + 32 * m + col;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * ks + 8 * kh + e;
        v[e] = (k < R ? Wf[(size_t)rowi * R + min(k, R - 1)] : (k == R ? bf[rowi] : 0.0f)) * sW;
      }
      const Frag2 f = split8h(v);
      wh[c][ks] = f.h;
      wl[c][ks] = f.l;
    }
  }
}

// largest magnitude of Wf' (all 3F rows), block-wide
template <int R>
__device__ __forceinline__ float filter_max(const float* __restrict__ Wf, const float* __restrict__ bf, float* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = 0.0f;
  for (int i = tid; i < 3 * PM_F * R; i += 256) mx = fmaxf(mx, fabsf(Wf[i]));
  for (int i = tid; i < 3 * PM_F; i += 256) mx = fmaxf(mx, fabsf(bf[i]));
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return mx;
}

// One row of a tile as it travels through the load pipeline: the edge (stage 1), then its data (stage 2).
template <int R>
struct EdgeRow {
  f32x4 p[3];   // this lane's share of the radial basis: k = 8 kh .. 8 kh + 7 and (kh = 0) k = 16 .. 19
  float fc, d0, d1, d2;
  int other;
  // stage 2: requests for the data of edge `ec` (clamped: always a valid edge)
  __device__ __forceinline__ void request(const PainnMmaArgs& A, int ec, int kh) {
    static_assert(R % 4 == 0 && R + 1 <= 32, "R: a multiple of 4, at most 28");
    const float* row = A.phi + (size_t)ec * R;
#pragma unroll
    for (int h = 0; h < 3; ++h) {
      const int kg = h < 2 ? 8 * kh + 4 * h : 16 + 8 * kh;  // first index of the group of four (kh is a run-time value)
      p[h] = *reinterpret_cast<const f32x4*>(row + min(kg, R - 4));
    }
    fc = A.fcut[ec];
    d0 = A.dir[3 * ec];
    d1 = A.dir[3 * ec + 1];
    d2 = A.dir[3 * ec + 2];
    other = (int)A.idx_other[ec];
  }
  // phi' = [phi, 1, 0 ..] * 2^14 as A fragments: lane (row j, half kh), k = 16 ks + 8 kh + e.  The factor fcut of the
  // filter is applied to the PRODUCT (one multiplication per row and channel in the epilogue), like the reference does
  // (painn.py:241): a power-of-two scale keeps the split exact - with phi * fcut formed here the compiler fused the
  // multiplication into one of the two fp16 conversions of split8h and not the other, and a product that lands on an
  // fp16 tie came out with h and l taken from different roundings (seen: 2 units in 2659, 1.5e-4 of a filter row).
  __device__ __forceinline__ void fragments(int kh, bool valid, u32x4 (&ah)[2], u32x4 (&al)[2]) const {
    const float sc = valid ? 16384.0f : 0.0f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kg = 16 * ks + 8 * kh + 4 * h;
        const f32x4 q = ks == 0 ? p[h] : p[2];
        const bool in = kg < R && (ks == 0 || h == 0), isb = kg == R;
        v[4 * h + 0] = in ? q[0] * sc : (isb ? sc : 0.0f);
        v[4 * h + 1] = in ? q[1] * sc : 0.0f;
        v[4 * h + 2] = in ? q[2] * sc : 0.0f;
        v[4 * h + 3] = in ? q[3] * sc : 0.0f;
      }
      const Frag2 f = split8h(v);
      ah[ks] = f.h;
      al[ks] = f.l;
    }
  }
};

#ifdef PM_TIMING
#define PM_MARK(slot)                                                                         \
  do {                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    if (blockIdx.x == 3 && tid == 0 && pm_t < 64) pm_dbg[pm_t * 8 + (slot)] = clock64();      \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  } while (0)
__device__ long long pm_dbg[64 * 8];
#else
#define PM_MARK(slot) do {} while (0)
#endif

// the value of lane l ^ 32 (v_permlane32_swap: one vector instruction, no LDS crossbar round trip)
__device__ __forceinline__ float swap_halves(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // r[0]: upper half <- lower half of u; r[1]: lower <- upper
  return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

struct TilePos {  // a team's position in its sequence of tiles (wave-uniform)
  int mol, a0, n, g1, gb, valid;
};

// ------------------------------------------------------------------------------------------------- forward
// q_out[i] = q[i] + sum_e dq_e, mu_out[i] = mu[i] + sum_e (dmuR_e dir_e + dmumu_e mu[j_e]) over the edges e of target i,
// [dq, dmuR, dmumu]_e = W_e * x[j_e]   (painn.py:54-64)
// MZ: mu is identically zero (the FIRST interaction, painn.py:249; A.mu is NULL): its rows are neither staged nor
// gathered - half of the LDS gathers of a tile, which is what the kernel is bound by - and the message's dmumu * mu_j term
// and the residual mu[i] drop out as the exact zeros they are in the general form.
template <int R, bool MZ = false>
__global__ __launch_bounds__(256, 2) void k_painn_fwd_mma(PainnMmaArgs A) {
  constexpr int F = PM_F, ROWB = 3 * F * 4;  // bytes of a staged row
  static_assert(R <= 20, "EdgeRow holds the three groups of four an R <= 20 basis needs per lane");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, m = tid >> 6, j = lane & 31, kh = lane >> 5;
  float* xs = reinterpret_cast<float*>(smem);               // [max_n][3F]
  float* ms = xs + (size_t)A.max_n * 3 * F;                 // [max_n][3F]
  float* qs = ms + (size_t)A.max_n * 3 * F;                 // [max_n][F]
  float* tab = qs + (size_t)A.max_n * F + m * 160;          // per wave [5][32]: byte offset of the source row, dir x, y, z, kk * fcut
  float* red = qs + (size_t)A.max_n * F + 4 * 160;          // [4]
  const float wmax = filter_max<R>(A.Wf, A.bf, red);
  int eW;
  const float sW = pow2_scale_to_2p14(wmax, eW);
  const float kk = __builtin_amdgcn_ldexpf(1.0f, eW - 28);  // undoes 2^(14 - eW) and the 2^14 of phi'
  u32x4 wh[3][2], wl[3][2];
  load_filter_fragments<R>(A.Wf, A.bf, m, lane, sW, wh, wl);
  const int f = 32 * m + j;  // this lane's feature
  const int stride = gridDim.x;
  // ---- the tile sequence: molecules blockIdx.x, + gridDim.x, ...; eight groups per tile
  auto enter = [&](TilePos& t) {  // t.mol set: the first tile of that molecule (every atom has at least one group)
    for (;;) {
      if (t.mol >= A.B) {
        t = TilePos{t.mol, 0, 0, 0, 0, 0};
        return;
      }
      const int a0 = A.mol_ptr[t.mol], n = A.mol_ptr[t.mol + 1] - a0;
      const int g0 = A.mol_grp[t.mol], g1 = A.mol_grp_end != nullptr ? A.mol_grp_end[t.mol] : A.mol_grp[t.mol + 1];
      if (g1 > g0 && n <= A.max_n) {
        t = TilePos{t.mol, a0, n, g1, g0, 1};
        return;
      }
      t.mol += stride;  // (a molecule without atoms - or with more than the staged rows: left to the per-atom kernel)
    }
  };
  auto advance = [&](TilePos& t) {
    if (!t.valid) return;
    t.gb += 8;
    if (t.gb >= t.g1) {
      t.mol += stride;
      enter(t);
    }
  };
  auto edge_of = [&](const TilePos& t) {  // stage 1: the edge of this lane's row
    const int grow = t.gb + (j >> 2);
    return t.valid && grow < t.g1 ? A.row_edge[4 * t.gb + j] : -1;
  };
  TilePos T0{(int)blockIdx.x, 0, 0, 0, 0, 0};
  enter(T0);
  TilePos T1 = T0;
  advance(T1);
  TilePos T2 = T1;
  advance(T2);
  EdgeRow<R> r0, r1;
  int e0 = edge_of(T0), e1 = edge_of(T1), e2 = edge_of(T2);
  r0.request(A, max(e0, 0), kh);
  // group codes of a tile (atom * 2 + last group of its atom; -1 past the molecule): lane g < 8 fetches the code of
  // group g with an ordinary vector load and the codes are read back with v_readlane.  (As eight scalar loads they sat in
  // the LGKM queue next to the LDS traffic: scalar loads return out of order, so every LDS wait of the tile in work
  // became a wait for the NEXT tile's codes - a memory round trip per tile.)
  auto codes_of = [&](const TilePos& t) {
    const int g = lane & 7;
    return t.valid && t.gb + g < t.g1 ? A.grp_atom[t.gb + g] : -1;
  };
  int gc0 = codes_of(T0), gc1 = gc0;
  int cur_mol = -1;
  float t0 = 0.0f, t1 = 0.0f;  // running sums of the current target atom: components 2 kh, 2 kh + 1
  const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(A.out0, 0, (uint32_t)A.N * F * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(A.out1, 0, (uint32_t)A.N * 3 * F * 4u, 0x00020000);
#ifdef PM_TIMING
  int pm_t = -1;
#endif
  while (T0.valid) {
#ifdef PM_TIMING
    ++pm_t;
#endif
    PM_MARK(0);
    if (T0.mol != cur_mol) {  // a new molecule: its rows of x and mu into LDS
      cur_mol = T0.mol;
      __syncthreads();  // the previous molecule's rows are no longer read
      // (all requests of a pass in flight together: a load-store loop pays one memory round trip per iteration)
      const f32x4* xg = reinterpret_cast<const f32x4*>(A.xc + (size_t)T0.a0 * 3 * F);
      const f32x4* mg = MZ ? xg : reinterpret_cast<const f32x4*>(A.mu + (size_t)T0.a0 * 3 * F);   // (MZ: never read)
      const f32x4* qg = reinterpret_cast<const f32x4*>(A.q + (size_t)T0.a0 * F);
      const int cnt = T0.n * 3 * F / 4, cntq = T0.n * F / 4;
      for (int base = 0, baseq = 0; base < cnt; base += 8 * 256, baseq += 3 * 256) {
        f32x4 bx[8], bm[8], bq[3];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int u = min(base + k * 256 + tid, cnt - 1);
          bx[k] = xg[u];
          if constexpr (!MZ) bm[k] = mg[u];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) bq[k] = qg[min(baseq + k * 256 + tid, cntq - 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int u = base + k * 256 + tid;
          if (u < cnt) {
            reinterpret_cast<f32x4*>(xs)[u] = bx[k];
            if constexpr (!MZ) reinterpret_cast<f32x4*>(ms)[u] = bm[k];
          }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int u = baseq + k * 256 + tid;
          if (u < cntq) reinterpret_cast<f32x4*>(qs)[u] = bq[k];
        }
      }
      __syncthreads();
    }
    PM_MARK(1);
    // ---- requests one and two tiles ahead (consumed in the next iterations)
    r1.request(A, max(e1, 0), kh);
    gc1 = codes_of(T1);
    TilePos T3 = T2;
    advance(T3);
    const int e3 = edge_of(T3);
    PM_MARK(2);
    // ---- this tile: A fragments, row table
    const bool valid = e0 >= 0;
    u32x4 ah[2], al[2];
    r0.fragments(kh, valid, ah, al);
    if (kh == 0) {
      reinterpret_cast<int*>(tab)[j] = valid ? (r0.other - T0.a0) * ROWB : 0;
      tab[32 + j] = r0.d0;
      tab[64 + j] = r0.d1;
      tab[96 + j] = r0.d2;
      tab[128 + j] = valid ? kk * r0.fc : 0.0f;
    }
    int ga[8], glast[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int code = __builtin_amdgcn_readlane(gc0, g);
      ga[g] = code < 0 ? -1 : code >> 1;
      glast[g] = code & 1;
    }
    PM_MARK(3);
    // ---- filter of the 32 rows, this wave's columns: acc[c][r] = (phi' Wf'^T)[row c_row(r)][c F + f] (scaled)
    f32x16 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // (three independent accumulator chains, interleaved)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = mfma_f16(al[ks], wh[c][ks], acc[c]);
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = mfma_f16(ah[ks], wl[c][ks], acc[c]);
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = mfma_f16(ah[ks], wh[c][ks], acc[c]);
    }
    PM_MARK(4);
    // ---- messages, summed over the four rows of a group (the lane's groups are 2q + kh)
    float gs[4][4];
    const uint8_t* xs_l = reinterpret_cast<const uint8_t*>(xs) + 4 * f;
    const uint8_t* ms_l = reinterpret_cast<const uint8_t*>(ms) + 4 * f;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int4 jo = *reinterpret_cast<const int4*>(tab + 8 * q4 + 4 * kh);
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(tab + 32 + 8 * q4 + 4 * kh);
      const f32x4 d1 = *reinterpret_cast<const f32x4*>(tab + 64 + 8 * q4 + 4 * kh);
      const f32x4 d2 = *reinterpret_cast<const f32x4*>(tab + 96 + 8 * q4 + 4 * kh);
      const f32x4 kf = *reinterpret_cast<const f32x4*>(tab + 128 + 8 * q4 + 4 * kh);
      const int jov[4] = {jo.x, jo.y, jo.z, jo.w};
      float xv[4][3], mv[4][3];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          xv[e][c] = *reinterpret_cast<const float*>(xs_l + jov[e] + c * F * 4);
          if constexpr (!MZ) mv[e][c] = *reinterpret_cast<const float*>(ms_l + jov[e] + c * F * 4);
        }
      float sq = 0.0f, s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q4 + e;
        const float x0 = (acc[0][r] * kf[e]) * xv[e][0], x1 = (acc[1][r] * kf[e]) * xv[e][1],
                    x2 = (acc[2][r] * kf[e]) * xv[e][2];                                  // painn.py:241, :56
        sq += x0;                                                                         // :59
        if constexpr (MZ) {   // (x2 * 0 added to the rounded product x1 * d leaves it: the same value, two instructions less)
          s0 = add_rn(s0, mul_rn(x1, d0[e]));
          s1 = add_rn(s1, mul_rn(x1, d1[e]));
          s2 = add_rn(s2, mul_rn(x1, d2[e]));
        } else {
          s0 += x1 * d0[e] + x2 * mv[e][0];                                               // :60-61
          s1 += x1 * d1[e] + x2 * mv[e][1];
          s2 += x1 * d2[e] + x2 * mv[e][2];
        }
      }
      gs[q4][0] = sq;
      gs[q4][1] = s0;
      gs[q4][2] = s1;
      gs[q4][3] = s2;
    }
    PM_MARK(5);
    // ---- the halves exchange: half kh keeps components 2 kh, 2 kh + 1 of all eight groups
    float V[8][2];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const float mine = kh ? gs[q4][2 + c2] : gs[q4][c2];       // my group 2 q4 + kh, a component I keep
        const float send = kh ? gs[q4][c2] : gs[q4][2 + c2];       // the same group, a component of the other half
        const float got = swap_halves(send);                        // its group 2 q4 + (1 - kh), a component I keep
        V[2 * q4][c2] = kh ? got : mine;
        V[2 * q4 + 1][c2] = kh ? mine : got;
      }
    PM_MARK(6);
    // ---- runs of groups with one target, in group order.  A run that ends its atom is written out; only the LAST run of
    // a tile can be cut by the tile's end, and it goes on as the first run of the next tile: the running sums t0 / t1
    // simply stay in their registers across the tile boundary.  Every group issues its two stores unconditionally
    // (buffer addressing: the offset of a group that does not end its atom is out of range, the store is dropped): with
    // stores behind branches the number of requests in flight is unknown to the compiler and the wait for the next
    // tile's operands becomes a wait for everything - including the acknowledgement of these stores.
    float res0[8], res1[8];  // what the two components of this half start from: q / mu x (kh = 0), mu y / mu z (kh = 1)
#pragma unroll
    for (int g = 0; g < 8; ++g) {  // all LDS reads of the eight groups first, then arithmetic and stores
      const bool fin = ga[g] >= 0 && glast[g];
      const int al = fin ? ga[g] - T0.a0 : 0;
      if constexpr (MZ) {
        res0[g] = kh ? 0.0f : qs[(size_t)al * F + f];
        res1[g] = 0.0f;
      } else {
        const float* mrow = ms + (size_t)al * 3 * F + f;
        res0[g] = kh ? mrow[F] : qs[(size_t)al * F + f];
        res1[g] = kh ? mrow[2 * F] : mrow[0];
      }
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      t0 += V[g][0];  // (groups past the molecule's end have zero rows)
      t1 += V[g][1];
      const bool fin = ga[g] >= 0 && glast[g];  // uniform
      constexpr uint32_t OOR = 0xFFFFFF00u;     // beyond every buffer: the store is dropped
      const uint32_t oq = (uint32_t)(ga[g] * F + f) * 4u, om = (uint32_t)(ga[g] * 3 * F + f) * 4u;
      if (kh == 0) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, res0[g] + t0), rs_q, fin ? oq : OOR, 0, 0);  // :63
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, res1[g] + t1), rs_m, fin ? om : OOR, 0, 0);  // :64, x
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, res0[g] + t0), rs_m, fin ? om + F * 4u : OOR, 0, 0);      // y
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, res1[g] + t1), rs_m, fin ? om + 2 * F * 4u : OOR, 0, 0);  // z
      }
      t0 = fin ? 0.0f : t0;
      t1 = fin ? 0.0f : t1;
    }
    PM_MARK(7);
    // ---- rotate the pipeline
    T0 = T1;
    T1 = T2;
    T2 = T3;
    e0 = e1;
    e1 = e2;
    e2 = e3;
    r0 = r1;
    gc0 = gc1;
  }
}

inline size_t painn_mma_lds(int max_n) {
  return ((size_t)max_n * 7 * PM_F + 4 * 160 + 8) * sizeof(float);
}

}  // namespace

#ifdef PM_TIMING
extern "C" int geossl_painn_mma_debug_read(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pm_dbg), sizeof(long long) * 64 * 8);
}
#endif

extern "C" int geossl_painn_interaction_fwd_mma_dyn(const float* q, const float* mu, const float* xc, const int64_t* idx_j,
                                                    const int32_t* row_edge, const int32_t* grp_atom,
                                                    const int32_t* mol_grp, const float* phi, const float* fcut,
                                                    const float* dir, const float* Wf, const float* bf,
                                                    const int32_t* mol_ptr, int64_t B, int max_n, int64_t N, int F, int R,
                                                    float* q_out, float* mu_out, const int32_t* mol_grp_end,
                                                    hipStream_t stream) {
  if (N <= 0 || B <= 0) return 0;
  // molecules above the rows the LDS holds (44 atoms: geossl_painn_stage_cap(0, ..)) are skipped - the caller covers
  // them with geossl_painn_interaction_fwd_atoms
  if (max_n > 44) max_n = 44;
  const size_t lds = painn_mma_lds(max_n);
  if (F != PM_F || lds > 160 * 1024 || (R != 8 && R != 16 && R != 20) || N * 3 * PM_F * 4 >= ((int64_t)1 << 32))
    return (int)hipErrorInvalidValue;
  PainnMmaArgs a{};
  a.q = q; a.mu = mu; a.xc = xc; a.idx_other = idx_j; a.row_edge = row_edge; a.grp_atom = grp_atom; a.mol_grp = mol_grp;
  a.phi = phi; a.fcut = fcut; a.dir = dir; a.Wf = Wf; a.bf = bf; a.mol_ptr = mol_ptr; a.B = (int)B; a.max_n = max_n; a.N = (int)N;
  a.out0 = q_out; a.out1 = mu_out; a.mol_grp_end = mol_grp_end;
  const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
  const int nb = (int)(B < 256 * per_cu ? B : 256 * per_cu);
#define LAUNCH_FWD_MMA(RV)                                                                              \
  do {                                                                                                  \
    if (mu == nullptr) {  /* mu identically zero: the first interaction */                              \
      allow_big_lds(&k_painn_fwd_mma<RV, true>);                                                        \
      hipLaunchKernelGGL((k_painn_fwd_mma<RV, true>), dim3((unsigned)nb), dim3(256), lds, stream, a);   \
    } else {                                                                                            \
      allow_big_lds(&k_painn_fwd_mma<RV>);                                                              \
      hipLaunchKernelGGL((k_painn_fwd_mma<RV>), dim3((unsigned)nb), dim3(256), lds, stream, a);         \
    }                                                                                                   \
  } while (0)
  if (R == 20) LAUNCH_FWD_MMA(20); else if (R == 16) LAUNCH_FWD_MMA(16); else LAUNCH_FWD_MMA(8);
#undef LAUNCH_FWD_MMA
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_painn_interaction_fwd_mma(const float* q, const float* mu, const float* xc, const int64_t* idx_j,
                                                const int32_t* row_edge, const int32_t* grp_atom,
                                                const int32_t* mol_grp, const float* phi, const float* fcut,
                                                const float* dir, const float* Wf, const float* bf,
                                                const int32_t* mol_ptr, int64_t B, int max_n, int64_t N, int F, int R,
                                                float* q_out, float* mu_out, hipStream_t stream) {
  return geossl_painn_interaction_fwd_mma_dyn(q, mu, xc, idx_j, row_edge, grp_atom, mol_grp, phi, fcut, dir, Wf, bf, mol_ptr,
                                              B, max_n, N, F, R, q_out, mu_out, nullptr, stream);
}
