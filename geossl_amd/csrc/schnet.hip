// SchNet pair-row and atom-row kernels: continuous-filter network (forward, backward), neighbour
// aggregation, embedding, readout.  Replaces, for the hot path, GaussianSmearing + InteractionBlock.mlp +
// CFConv.message/propagate (schnet.py:141-145,185-195,205-207), Embedding (:89) and scatter readout (:115).
#include "common.h"
#include "aggregate_reg.h"
#include "geossl_hip.h"
#include "tn.h"

#include <cstdlib>

using namespace geossl;

namespace {

inline int grid1d(int64_t n, int block, int cap = 2048) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------- K2 (export)
__global__ void k_rbf(const float* __restrict__ d, int64_t E, const float* __restrict__ offset, int G, float coeff,
                      float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E * G; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / G;
    const int g = (int)(i - e * G);
    const float diff = d[e] - offset[g];
    out[i] = expf(coeff * (diff * diff));
  }
}

// ------------------------------------------------------------------------- ShiftedSoftplus on its own (schnet.py:210-216)
__global__ void k_ssp_fwd(const float* __restrict__ x, int64_t n, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = ssp(x[i]);
}
__global__ void k_ssp_bwd(const float* __restrict__ y, const float* __restrict__ dy, int64_t n, float* __restrict__ dx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = dy[i] * dssp_from_out(y[i]);
}

// ------------------------------------------------------------------------------------------ d aggregate / d filter
// out[p][c] = f0 * a[i][c] * b[j][c] + f1 * a[j][c] * b[i][c]   for pair slot p = (i < j), f0 / f1 its two edge flags
// (exchanged when swap): the gradient of the neighbour aggregation (k_aggregate) with respect to the filter rows, as a
// tensor - used where the filter gradient itself has to stay differentiable (second-order path); the first-order path
// never materialises it (filter_bwd.hip).
__global__ void k_pair_product(const float* __restrict__ a, const float* __restrict__ b,
                               const int32_t* __restrict__ pair_i, const int32_t* __restrict__ pair_j,
                               const uint8_t* __restrict__ pair_flag, int64_t P, int F, int swap,
                               float* __restrict__ out) {
  const int Q = F / 4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < P * Q; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = idx / Q;
    const int c = 4 * (int)(idx - p * Q);
    unsigned fl = pair_flag[p];
    if (swap) fl = ((fl & 1u) << 1) | ((fl >> 1) & 1u);
    const float f0 = (fl & 1u) ? 1.0f : 0.0f, f1 = (fl & 2u) ? 1.0f : 0.0f;
    const size_t oi = (size_t)pair_i[p] * F + c, oj = (size_t)pair_j[p] * F + c;
    const f32x4 ai = *reinterpret_cast<const f32x4*>(a + oi), aj = *reinterpret_cast<const f32x4*>(a + oj);
    const f32x4 bi = *reinterpret_cast<const f32x4*>(b + oi), bj = *reinterpret_cast<const f32x4*>(b + oj);
    *reinterpret_cast<f32x4*>(out + (size_t)p * F + c) = f0 * (ai * bj) + f1 * (aj * bi);
  }
}

// ---------------------------------------------------------------------------------------------------- K4
// One wave per molecule.  A lane owns VW = F/64 adjacent feature columns of every atom row of the molecule, so the
// LDS copy of x and the accumulators are lane-private (one barrier after staging, no atomics); each filter row is
// read once and applied in both directions.  Separate multiply and add (no FMA contraction) in ascending source
// order: the same rounding sequence as a sequential index_add over the canonical edge list.
//
// The walk over the n(n-1)/2 pair slots is serial per lane; everything about a slot that is the same for all lanes
// - its flags, the pair (a, b), row ends - is kept in scalar registers, and the two columns of a lane move as 8-byte
// accesses and packed fp32 operations (details at the walk).  With ragged molecules the blocks are
// started largest molecule first (`order`): the largest one bounds the launch from below.

template <int VW>
struct AggVec;
template <>
struct AggVec<1> {
  typedef float type;
};
template <>
struct AggVec<2> {
  typedef f32x2 type;
};

template <int VW>
__global__ __launch_bounds__(64) void k_aggregate(const float* __restrict__ x, const float* __restrict__ Wf,
                                                  const uint8_t* __restrict__ pair_flag,
                                                  const int32_t* __restrict__ mol_ptr,
                                                  const int32_t* __restrict__ pair_ptr,
                                                  const int32_t* __restrict__ order, int B, int F, int max_n, int swap,
                                                  float* __restrict__ out) {
#pragma clang fp contract(off)
  typedef typename AggVec<VW>::type V;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x >= B) return;
  const int m = order != nullptr ? order[blockIdx.x] : (int)blockIdx.x;
  const int lane = threadIdx.x, f = VW * lane;  // first of this lane's columns
  const bool col = f < F;
  const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0, base = pair_ptr[m], np = n * (n - 1) / 2;
  float* xs = smem;                // [max_n][F]
  float* acc = smem + max_n * F;   // [max_n][F]
  uint8_t* sfl = reinterpret_cast<uint8_t*>(smem + 2 * max_n * F);  // [np] edge flags of the pair slots
  for (int p = lane; p < np; p += 64) {
    unsigned fl = pair_flag[base + p];
    if (swap) fl = ((fl & 1u) << 1) | ((fl >> 1) & 1u);
    sfl[p] = (uint8_t)fl;
  }
  if (col)
    for (int i = 0; i < n; ++i) {
      *reinterpret_cast<V*>(xs + i * F + f) = *reinterpret_cast<const V*>(x + (size_t)(a0 + i) * F + f);
      *reinterpret_cast<V*>(acc + i * F + f) = V(0.0f);
    }
  __syncthreads();
  if (col && np > 0) {
      const float* __restrict__ wcol = Wf + (size_t)base * F + f;
    const float* xl = xs + f;
    float* al = acc + f;
    // Walk by row atom a, U consecutive partners b at a time (a chunk never crosses a row, so its U accumulator rows
    // are distinct): the flags of the chunk become two scalar bit masks (one LDS read + ballots), the U rows of x and
    // of the accumulators are requested together and the U filter rows of the NEXT chunk are already in flight - no
    // wait inside a chunk depends on another.  Slots past the end of a row are computed on clamped rows and dropped
    // by selects (not by multiplying with zero: the sums stay bit-exact).  The accumulator of the row atom lives in
    // registers; it starts from the LDS value, which already holds every earlier source a' < a, so the summation
    // order per target stays ascending in the source index.
    constexpr int U = 8, D = 4;  // D chunks (32 filter rows, 16 KB per wave) requested ahead of the one in work: a
                                 // wave's stream is bound by the memory round trip, and with ragged molecules
                                 // (LDS sized for the largest) only half as many waves are resident
    V w[D][U];
    auto request = [&](int a, int b0, V (&dst)[U]) {
      const int rs = a * n - a * (a + 1) / 2 - a - 1;   // slot of (a, b) = rs + b
#pragma unroll
      for (int u = 0; u < U; ++u)
        dst[u] = __builtin_nontemporal_load(
            reinterpret_cast<const V*>(wcol + (uint32_t)(rs + min(b0 + u, n - 1)) * (uint32_t)F));  // streamed once
    };
    auto advance = [&](int& a, int& b0) {
      b0 += U;
      if (b0 >= n) {
        ++a;
        b0 = a + 1;
      }
    };
    int a = 0, b0 = 1, ap = 0, bp = 1;  // chunk in work, next chunk to request
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (ap < n - 1) {
        request(ap, bp, w[k]);
        advance(ap, bp);
      }
    V xa = *reinterpret_cast<const V*>(xl), acc_a = V(0.0f);
    while (a < n - 1) {
#pragma unroll
      for (int k = 0; k < D; ++k) {
        if (a < n - 1) {
          const int rs = a * n - a * (a + 1) / 2 - a - 1;
          const unsigned flv = (lane < U && b0 + lane < n) ? (unsigned)sfl[rs + b0 + lane] : 0u;
          const unsigned long long m0 = __builtin_amdgcn_ballot_w64((flv & 1u) != 0u);  // edge b -> a
          const unsigned long long m1 = __builtin_amdgcn_ballot_w64((flv & 2u) != 0u);  // edge a -> b
          V xb[U], ab[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int bo = min(b0 + u, n - 1) * F;
            xb[u] = *reinterpret_cast<const V*>(xl + bo);
            ab[u] = *reinterpret_cast<const V*>(al + bo);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const V t0 = xb[u] * w[k][u];
            const V s0 = acc_a + t0;
            acc_a = ((m0 >> u) & 1ull) ? s0 : acc_a;
            const V t1 = xa * w[k][u];
            const V s1 = ab[u] + t1;
            if ((m1 >> u) & 1ull) *reinterpret_cast<V*>(al + (b0 + u) * F) = s1;   // uniform: a scalar branch
          }
          const int a_was = a;
          advance(a, b0);
          if (a != a_was) {  // row finished: publish its sum, move to the next row atom
            *reinterpret_cast<V*>(al + a_was * F) = acc_a;
            if (a < n - 1) {
              xa = *reinterpret_cast<const V*>(xl + a * F);
              acc_a = *reinterpret_cast<const V*>(al + a * F);
            }
          }
          if (ap < n - 1) {  // refill this buffer
            request(ap, bp, w[k]);
            advance(ap, bp);
          }
        }
      }
    }
  }
  if (col)
    for (int i = 0; i < n; ++i)
      *reinterpret_cast<V*>(out + (size_t)(a0 + i) * F + f) = *reinterpret_cast<const V*>(acc + i * F + f);
}

// all molecules of the launch fit one size class (uniform batches: the best register allocation for that class)
template <int NMAX>
#ifndef AGG_WPS_SMALL
#define AGG_WPS_SMALL 3
#endif
__global__ __launch_bounds__(64, (NMAX <= 20 ? AGG_WPS_SMALL : 3)) void k_aggregate_reg(
    const float* __restrict__ x, const float* __restrict__ Wf, const uint8_t* __restrict__ pair_flag,
    const int32_t* __restrict__ mol_ptr, const int32_t* __restrict__ pair_ptr, const int32_t* __restrict__ order, int B,
    int F, int swap, float* __restrict__ out) {
  if ((int)blockIdx.x >= B) return;
  const int m = order != nullptr ? order[blockIdx.x] : (int)blockIdx.x;
  const int lane = threadIdx.x, f = 2 * lane < F ? 2 * lane : -2;
  // F = 64: lanes 32..63 own no channels but stay (lane b also reads the flag of partner b, b up to 32): they work on
  // column 0 and store nothing
  const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0, base = pair_ptr[m];
  aggregate_reg_body<NMAX>(x, Wf, pair_flag, a0, n, base, lane, f, F, swap, out);
}

// ragged batches in ONE launch: every wave takes the unrolled walk of its own molecule's size class (molecules are
// started largest first, so neighbouring waves mostly run the same code)
__global__ __launch_bounds__(64, 2) void k_aggregate_reg_ragged(
    const float* __restrict__ x, const float* __restrict__ Wf, const uint8_t* __restrict__ pair_flag,
    const int32_t* __restrict__ mol_ptr, const int32_t* __restrict__ pair_ptr, const int32_t* __restrict__ order, int B,
    int F, int swap, float* __restrict__ out) {
  if ((int)blockIdx.x >= B) return;
  const int m = order != nullptr ? order[blockIdx.x] : (int)blockIdx.x;
  const int lane = threadIdx.x, f = 2 * lane < F ? 2 * lane : -2;  // -2: no channels (see k_aggregate_reg)
  const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0, base = pair_ptr[m];
  const int nu = __builtin_amdgcn_readfirstlane(n);
#define AGG_CLASS(NM) if (nu <= NM) { aggregate_reg_body<NM>(x, Wf, pair_flag, a0, n, base, lane, f, F, swap, out); return; }
  AGG_CLASS(8) AGG_CLASS(12) AGG_CLASS(16) AGG_CLASS(18) AGG_CLASS(20) AGG_CLASS(22) AGG_CLASS(24) AGG_CLASS(26)
  AGG_CLASS(28) AGG_CLASS(30) AGG_CLASS(33)
#undef AGG_CLASS
}

// ragged batches with a host-built work list (layout.py: MolLayout.agg_work): entry = molecule | part << 24, largest
// molecules first.  Molecules of fewer than AGG_K2_MIN atoms are one work item (from 21 atoms on with a deep ring of
// filter-row requests: the walk of a large molecule is a chain of memory round trips), AGG_K2_MIN .. AGG_K4_MIN-1 atoms
// two target groups, more atoms four (geossl_aggregate_parts gives the same mapping to the host).
#ifndef AGG_K2_MIN
#define AGG_K2_MIN 27
#endif
#ifndef AGG_K4_MIN
#define AGG_K4_MIN 31
#endif
#ifndef AGG_RING_BIG
#define AGG_RING_BIG 40
#endif
// (a molecule above 33 atoms: one work item per atom - the part field of a work word has eight bits)
// a size class above 20 atoms (molecules of LO .. NM atoms): whole, two or four target groups, by the molecule's size;
// only the forms a class can meet are instantiated
template <int NM>
__device__ __forceinline__ void aggregate_big(const float* __restrict__ x, const float* __restrict__ Wf,
                                              const uint8_t* __restrict__ pair_flag, int a0, int n, int base, int lane,
                                              int f, int F, int swap, float* __restrict__ out, int kparts, int part) {
  constexpr int LO = NM == 33 ? 31 : NM - 1;
#define AGG_PART(KK, PP) aggregate_reg_part<NM, KK, PP>(x, Wf, pair_flag, a0, n, base, lane, f, F, swap, out)
  if (kparts == 1) {
    if constexpr (LO < AGG_K2_MIN)
      aggregate_reg_body<NM, AGG_RING_BIG>(x, Wf, pair_flag, a0, n, base, lane, f, F, swap, out);
  } else if (kparts == 2) {
    if constexpr (NM >= AGG_K2_MIN && LO < AGG_K4_MIN) {
      if (part == 0) AGG_PART(2, 0); else AGG_PART(2, 1);
    }
  } else {
    if constexpr (NM >= AGG_K4_MIN) {
      if (part == 0) AGG_PART(4, 0); else if (part == 1) AGG_PART(4, 1);
      else if (part == 2) AGG_PART(4, 2); else AGG_PART(4, 3);
    }
  }
#undef AGG_PART
}

// The work list is laid out in EIGHT queues of equal length Q = nwork / 8 (padded with -1), one per XCD: workgroups are
// dealt round-robin to the 8 XCDs (each with its own L2), so workgroup b takes entry b / 8 of queue b mod 8 and the work
// items of ONE molecule - consecutive entries of one queue: the target atoms of a large molecule, the target groups of a
// 27..33-atom one, which read each other's filter rows - run on one XCD, close in time: the second read of a filter row
// is an L2 hit instead of a second trip to HBM.  The host (layout.aggregate_work_list) deals the molecules to the queues
// largest first in snake order.
__device__ __forceinline__ int work_item(const int32_t* __restrict__ work, int nwork, const int32_t* __restrict__ dyn_nwork) {
  const int nw = dyn_count(nwork, dyn_nwork), b = (int)blockIdx.x;
  if (b >= nw) return -1;
  return work[(b & 7) * (nw >> 3) + (b >> 3)];
}

__global__ __launch_bounds__(64, 2) void k_aggregate_reg_work(
    const float* __restrict__ x, const float* __restrict__ Wf, const uint8_t* __restrict__ pair_flag,
    const int32_t* __restrict__ mol_ptr, const int32_t* __restrict__ pair_ptr, const int32_t* __restrict__ work,
    int nwork, int F, int swap, float* __restrict__ out, const int32_t* __restrict__ dyn_nwork) {
  const int wk = work_item(work, nwork, dyn_nwork);
  if (wk == -1) return;
  const int m = wk & 0x00FFFFFF, part = __builtin_amdgcn_readfirstlane((wk >> 24) & 255);
  const int lane = threadIdx.x, f = 2 * lane < F ? 2 * lane : -2;  // -2: no channels (see k_aggregate_reg)
  const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0, base = pair_ptr[m];
  const int nu = __builtin_amdgcn_readfirstlane(n);
  if (nu > 33) {  // above the size classes: one work item per TARGET atom (aggregate_targets): a target's sum is a chain
    aggregate_targets(x, Wf, pair_flag, a0, n, base, lane, f, F, swap, out, part, part + 1);  // of memory round trips (32
    return;                                                                                  // partners each) - kept short
  }
  const int kparts = nu < AGG_K2_MIN ? 1 : (nu < AGG_K4_MIN ? 2 : 4);
#define AGG_CLASS(NM) if (nu <= NM) { aggregate_reg_body<NM>(x, Wf, pair_flag, a0, n, base, lane, f, F, swap, out); return; }
#define AGG_CLASS_BIG(NM) if (nu <= NM) { aggregate_big<NM>(x, Wf, pair_flag, a0, n, base, lane, f, F, swap, out, kparts, part); return; }
  AGG_CLASS(8) AGG_CLASS(12) AGG_CLASS(16) AGG_CLASS(18) AGG_CLASS(20)
  AGG_CLASS_BIG(22) AGG_CLASS_BIG(24) AGG_CLASS_BIG(26) AGG_CLASS_BIG(28) AGG_CLASS_BIG(30) AGG_CLASS_BIG(33)
#undef AGG_CLASS
#undef AGG_CLASS_BIG
}

// SMALL batches (the reference's batch size of 128): the walks above are chains of memory round trips - 14 .. 27 of them
// for a molecule of 18 .. 33 atoms - and a launch over a few hundred molecules lasts as long as its longest walk while
// most of the chip idles.  Here every work item is ONE target atom (work word = molecule | target << 24): a single round
// trip of <= 32 partner rows for the molecules of the size classes, the same fixed summation order.  Every filter row is
// read by both of its atoms (twice the bytes): this form is for launches that are latency-, not bandwidth-bound.
// (second launch bound = waves per SIMD the compiler must leave room for: 2 -> up to 256 registers; a target's 64 rows in
// flight need 128 of them)
__global__ __launch_bounds__(64, 2) void k_aggregate_targets(
    const float* __restrict__ x, const float* __restrict__ Wf, const uint8_t* __restrict__ pair_flag,
    const int32_t* __restrict__ mol_ptr, const int32_t* __restrict__ pair_ptr, const int32_t* __restrict__ work,
    int nwork, int F, int swap, float* __restrict__ out, const int32_t* __restrict__ dyn_nwork) {
  const int wk = work_item(work, nwork, dyn_nwork);
  if (wk == -1) return;
  const int m = wk & 0x00FFFFFF, a = __builtin_amdgcn_readfirstlane((wk >> 24) & 255);
  const int lane = threadIdx.x, f = 2 * lane < F ? 2 * lane : -2;
  const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0, base = pair_ptr[m];
  aggregate_targets(x, Wf, pair_flag, a0, n, base, lane, f, F, swap, out, a, a + 1);
}

// --------------------------------------------------------------------------------------------- embedding
__global__ void k_embedding_fwd(const int64_t* __restrict__ z, int64_t zs, const float* __restrict__ table, int C,
                                int64_t N, int F, float* __restrict__ out, int32_t* __restrict__ status,
                                const int32_t* __restrict__ dyn_N) {
  N = dyn_count((int)N, dyn_N);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * F; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = i / F;
    const int f = (int)(i - a * F);
    const int64_t c = z[a * zs];
    if (c < 0 || c >= C) {
      if (status != nullptr) *status = 1;
      out[i] = 0.0f;
    } else {
      out[i] = table[c * F + f];
    }
  }
}
// the same for F = 4 << SH4 (32, 64, 128): four columns per thread, shifts instead of the 64-bit division
template <int SH4>
__global__ __launch_bounds__(256) void k_embedding_fwd4(const int64_t* __restrict__ z, int64_t zs,
                                                        const float* __restrict__ table, int C, int N,
                                                        float* __restrict__ out, int32_t* __restrict__ status,
                                                        const int32_t* __restrict__ dyn_N) {
  N = dyn_count(N, dyn_N);
  const uint32_t total = (uint32_t)N << SH4;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t a = i >> SH4, f4 = i & ((1u << SH4) - 1u);
    const int64_t c = z[(int64_t)a * zs];
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (c < 0 || c >= C) {
      if (status != nullptr) *status = 1;
    } else {
      v = reinterpret_cast<const f32x4*>(table)[((uint32_t)c << SH4) + f4];
    }
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}

#define GEOSSL_EMB_CHUNKS 512
// block = one chunk of rows, split over NG groups of threads (rows g, g + NG, ... of the chunk to group g); thread f of a
// group owns feature column f of the group's [classes][F] accumulator table in LDS (thread-private columns: rows added
// in row order, no barriers), sixteen rows in flight per group; the group tables are then summed in group order.  Only
// the classes [cmin, cmax] a chunk has met leave the block (QM9 uses 5 of the 119 rows of the table: a chunk's whole table
// is 60 KB, its occupied band 4.5 KB), the band is recorded behind the partial tables for the second stage.
template <int NG>
__global__ void k_embedding_bwd_partial(const int64_t* __restrict__ z, int64_t zs, const float* __restrict__ dh,
                                        int64_t N, int F, int C, float* __restrict__ partial, int2* __restrict__ band,
                                        const int32_t* __restrict__ dyn_N) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [NG][C][F], then NG bands
  N = dyn_count((int)N, dyn_N);  // (chunks past the real rows leave an empty band)
  const int TF = blockDim.x / NG;  // threads of a group (F rounded up to whole waves)
  const int chunk = blockIdx.x, grp = threadIdx.x / TF, f = threadIdx.x - grp * TF;
  const int64_t per = (N + (int64_t)gridDim.x - 1) / (int64_t)gridDim.x;
  const int64_t lo = chunk * per, hi = min((int64_t)N, lo + per);
  float* tab = smem + (size_t)grp * C * F;
  int2* gband = reinterpret_cast<int2*>(smem + (size_t)NG * C * F);
  int cmin = C, cmax = -1;  // the same in every thread of a group
  if (f < F) {
    for (int c = 0; c < C; ++c) tab[c * F + f] = 0.0f;
    constexpr int U = 16;
    for (int64_t a0 = lo + grp; a0 < hi; a0 += (int64_t)U * NG) {
      float v[U];
      int cls[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t a = min(a0 + (int64_t)u * NG, hi - 1);
        v[u] = dh[a * F + f];
        cls[u] = (int)z[a * zs];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (a0 + (int64_t)u * NG < hi && cls[u] >= 0 && cls[u] < C) {
          tab[cls[u] * F + f] += v[u];
          cmin = min(cmin, cls[u]);
          cmax = max(cmax, cls[u]);
        }
    }
    if (f == 0) gband[grp] = make_int2(cmin, cmax);
  }
  if (NG > 1) {
    __syncthreads();
    if (grp != 0) return;
    for (int g = 0; g < NG; ++g) {
      cmin = min(cmin, gband[g].x);
      cmax = max(cmax, gband[g].y);
    }
  }
  if (f >= F) return;
  for (int c = cmin; c <= cmax; ++c) {
    float sum = smem[c * F + f];
    for (int g = 1; g < NG; ++g) sum += smem[((size_t)g * C + c) * F + f];  // group order
    partial[((size_t)chunk * C + c) * F + f] = sum;
  }
  if (f == 0) band[chunk] = make_int2(cmin, cmax);
}
// dtable[c][f] (+)= sum over the chunks whose band holds c, in chunk order: block = (class, 64 columns) x 4 slices of the
// chunk list (one wave each).  A wave first lists the chunks of its slice that hold the class (ballot over the bands,
// ascending), then sums their rows compensated like kahan_sum_strided; the four slice sums are combined in slice order.
__global__ __launch_bounds__(256) void k_embedding_bwd_reduce(const float* __restrict__ partial,
                                                              const int2* __restrict__ band, int nchunks, int C, int F,
                                                              float* __restrict__ dtable, int accumulate) {
  constexpr int PERMAX = GEOSSL_EMB_CHUNKS / 4;
  static_assert(PERMAX == 128, "two chunks per lane");
  const int PER = nchunks / 4;  // chunks per slice (nchunks: a multiple of 4, at most GEOSSL_EMB_CHUNKS)
  __shared__ float red[4][64];
  __shared__ int hits[4][PERMAX];
  const int c = blockIdx.y, lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + lane;
  int n = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int b = slice * PER + 64 * h + lane;
    const int2 r = 64 * h + lane < PER ? band[b] : make_int2(1, 0);
    const bool hit = c >= r.x && c <= r.y;
    const unsigned long long m = __ballot(hit);
    if (hit) hits[slice][n + __popcll(m & ((1ull << lane) - 1ull))] = b;
    n += __popcll(m);
  }
  __builtin_amdgcn_wave_barrier();
  float s = 0.0f;
  if (f < F) {
#pragma clang fp reassociate(off) contract(off)
    float k = 0.0f;
    const float* p = partial + (size_t)c * F + f;
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      const float y = p[(size_t)hits[slice][i] * C * F] - k;
      const float t = s + y;
      k = (t - s) - y;
      s = t;
    }
  }
  red[slice][lane] = s;
  __syncthreads();
  if (slice == 0 && f < F) {
    float v = accumulate ? dtable[(size_t)c * F + f] : 0.0f;
    v += red[0][lane];
    v += red[1][lane];
    v += red[2][lane];
    v += red[3][lane];
    dtable[(size_t)c * F + f] = v;
  }
}
// ----------------------------------------------------------------------------------------------- readout
__global__ void k_segment_reduce_fwd(const float* __restrict__ h, const int32_t* __restrict__ mol_ptr, int B, int F,
                                     int mean, float* __restrict__ out) {
  const int m = blockIdx.x;
  if (m >= B) return;
  const int a0 = mol_ptr[m], a1 = mol_ptr[m + 1];
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float s = 0.0f;
    for (int a = a0; a < a1; ++a) s += h[(size_t)a * F + f];
    if (mean) s = s / fmaxf((float)(a1 - a0), 1.0f);
    out[(size_t)m * F + f] = s;
  }
}
__global__ void k_segment_reduce_bwd(const float* __restrict__ dout, const int32_t* __restrict__ mol_ptr, int B, int F,
                                     int mean, float* __restrict__ dh, int accumulate) {
  const int m = blockIdx.x;
  if (m >= B) return;
  const int a0 = mol_ptr[m], a1 = mol_ptr[m + 1];
  const float cnt = fmaxf((float)(a1 - a0), 1.0f);
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float gv = dout[(size_t)m * F + f];
    if (mean) gv = gv / cnt;
    for (int a = a0; a < a1; ++a) {
      const size_t o = (size_t)a * F + f;
      dh[o] = accumulate ? dh[o] + gv : gv;
    }
  }
}

// ----------------------------------------------------------------------------------- F.normalize(h, dim=-1)
// pretrain_GeoSSL.py:193-195 (--normalize): y = h / max(||h||_2, eps), one wave per row; the norm is kept for the
// backward  dh = (g - y (g . y)) / max(||h||, eps)   (zero gradient through the clamp when ||h|| < eps, like ATen).
__global__ __launch_bounds__(256) void k_row_normalize_fwd(const float* __restrict__ h, int64_t N, int F, float eps,
                                                           float* __restrict__ y, float* __restrict__ norm) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* hr = h + row * F;
  float ss = 0.0f;
  for (int f = lane; f < F; f += 64) ss = fmaf(hr[f], hr[f], ss);
  ss = wave_sum(ss);
  const float nr = sqrtf(ss), den = fmaxf(nr, eps);
  for (int f = lane; f < F; f += 64) y[row * F + f] = hr[f] / den;
  if (lane == 0 && norm != nullptr) norm[row] = nr;
}
__global__ __launch_bounds__(256) void k_row_normalize_bwd(const float* __restrict__ g, const float* __restrict__ y,
                                                           const float* __restrict__ norm, int64_t N, int F, float eps,
                                                           float* __restrict__ dh) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* gr = g + row * F;
  const float* yr = y + row * F;
  float dot = 0.0f;
  for (int f = lane; f < F; f += 64) dot = fmaf(gr[f], yr[f], dot);
  dot = wave_sum(dot);
  const float nr = norm[row];
  const bool clamped = nr < eps;  // y = h / eps there: the norm does not depend on h
  const float inv = 1.0f / fmaxf(nr, eps);
  for (int f = lane; f < F; f += 64) dh[row * F + f] = (clamped ? gr[f] : gr[f] - yr[f] * dot) * inv;
}

inline int blocks_per_layer(int L, int ntiles) {
  int b = 256 / (L > 0 ? L : 1);
  if (b < 1) b = 1;
  if (b > ntiles) b = ntiles;
  return b;
}

}  // namespace

extern "C" int geossl_rbf_fwd(const float* d, int64_t E, const float* offset, int G, float coeff, float* out,
                              hipStream_t stream) {
  if (E <= 0) return 0;
  hipLaunchKernelGGL(k_rbf, dim3(grid1d(E * G, 256)), dim3(256), 0, stream, d, E, offset, G, coeff, out);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_ssp_fwd(const float* x, int64_t n, float* y, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_ssp_fwd, dim3(grid1d(n, 256)), dim3(256), 0, stream, x, n, y);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
extern "C" int geossl_ssp_bwd(const float* y, const float* dy, int64_t n, float* dx, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_ssp_bwd, dim3(grid1d(n, 256)), dim3(256), 0, stream, y, dy, n, dx);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_pair_product(const float* a, const float* b, const int32_t* pair_i, const int32_t* pair_j,
                                   const uint8_t* pair_flag, int64_t P, int F, int swap, float* out,
                                   hipStream_t stream) {
  if (P <= 0) return 0;
  if (F & 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_pair_product, dim3(grid1d(P * (F / 4), 256)), dim3(256), 0, stream, a, b, pair_i, pair_j,
                     pair_flag, P, F, swap, out);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_aggregate_parts(int n) {  // work items of an n-atom molecule in geossl_cfconv_aggregate_work
  if (n > 33) return n < 256 ? n : 255;
  if (n < AGG_K2_MIN) return 1;
  return n < AGG_K4_MIN ? 2 : 4;
}

extern "C" int geossl_cfconv_aggregate_work_dyn(const float* x, const float* Wf, const uint8_t* pair_flag,
                                                const int32_t* mol_ptr, const int32_t* pair_ptr, const int32_t* work,
                                                int64_t nwork, int max_n, int F, int swap, float* out,
                                                const int32_t* dyn_nwork, hipStream_t stream) {
  if (nwork <= 0) return 0;
  if (max_n > 255 || F > 128 || F <= 32) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_aggregate_reg_work, dim3((unsigned)nwork), dim3(64), 0, stream, x, Wf, pair_flag, mol_ptr,
                     pair_ptr, work, (int)nwork, F, swap, out, dyn_nwork);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

// work[i] = molecule | target atom << 24 for EVERY atom of the launch (geossl_amd/layout.py: aggregate_target_list)
extern "C" int geossl_cfconv_aggregate_targets_dyn(const float* x, const float* Wf, const uint8_t* pair_flag,
                                                   const int32_t* mol_ptr, const int32_t* pair_ptr, const int32_t* work,
                                                   int64_t nwork, int max_n, int F, int swap, float* out,
                                                   const int32_t* dyn_nwork, hipStream_t stream) {
  if (nwork <= 0) return 0;
  if (max_n > 255 || F > 128 || F <= 32) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_aggregate_targets, dim3((unsigned)nwork), dim3(64), 0, stream, x, Wf, pair_flag, mol_ptr, pair_ptr,
                     work, (int)nwork, F, swap, out, dyn_nwork);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_cfconv_aggregate_work(const float* x, const float* Wf, const uint8_t* pair_flag,
                                            const int32_t* mol_ptr, const int32_t* pair_ptr, const int32_t* work,
                                            int64_t nwork, int max_n, int F, int swap, float* out, hipStream_t stream) {
  return geossl_cfconv_aggregate_work_dyn(x, Wf, pair_flag, mol_ptr, pair_ptr, work, nwork, max_n, F, swap, out, nullptr,
                                          stream);
}

extern "C" int geossl_cfconv_aggregate(const float* x, const float* Wf, const uint8_t* pair_flag,
                                       const int32_t* mol_ptr, const int32_t* pair_ptr, const int32_t* order, int64_t B,
                                       int max_n, int F, int swap, float* out, hipStream_t stream) {
  if (B <= 0) return 0;
  if (max_n > 255 || F > 128) return (int)hipErrorInvalidValue;
  if (F > 32 && max_n <= 33 && !getenv("GEOSSL_AGG_LDS")) {
    // register form: one size class for the whole launch when the largest molecule has <= 20 atoms, else the kernel
    // that picks the class per molecule
#define AGG_REG(NM)                                                                                               \
  if (max_n <= NM) {                                                                                              \
    hipLaunchKernelGGL(k_aggregate_reg<NM>, dim3((unsigned)B), dim3(64), 0, stream, x, Wf, pair_flag, mol_ptr,    \
                       pair_ptr, order, (int)B, F, swap, out);                                                    \
    GEOSSL_CHECK_LAUNCH();                                                                                        \
    return 0;                                                                                                     \
  }
    AGG_REG(8) AGG_REG(12) AGG_REG(16) AGG_REG(18) AGG_REG(20)
#undef AGG_REG
    hipLaunchKernelGGL(k_aggregate_reg_ragged, dim3((unsigned)B), dim3(64), 0, stream, x, Wf, pair_flag, mol_ptr,
                       pair_ptr, order, (int)B, F, swap, out);
    GEOSSL_CHECK_LAUNCH();
    return 0;
  }
  const size_t lds = (size_t)2 * max_n * F * sizeof(float) + (size_t)(max_n * (max_n - 1) / 2) + 16;
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  if (F > 64) {
    allow_big_lds(&k_aggregate<2>);
    hipLaunchKernelGGL(k_aggregate<2>, dim3((unsigned)B), dim3(64), lds, stream, x, Wf, pair_flag, mol_ptr, pair_ptr,
                       order, (int)B, F, max_n, swap, out);
  } else {
    allow_big_lds(&k_aggregate<1>);
    hipLaunchKernelGGL(k_aggregate<1>, dim3((unsigned)B), dim3(64), lds, stream, x, Wf, pair_flag, mol_ptr, pair_ptr,
                       order, (int)B, F, max_n, swap, out);
  }
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_row_normalize_fwd(const float* h, int64_t N, int F, float eps, float* y, float* norm,
                                        hipStream_t stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_row_normalize_fwd, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, h, N, F, eps, y, norm);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
extern "C" int geossl_row_normalize_bwd(const float* g, const float* y, const float* norm, int64_t N, int F, float eps,
                                        float* dh, hipStream_t stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_row_normalize_bwd, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, g, y, norm, N, F, eps,
                     dh);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_embedding_fwd_dyn(const int64_t* z, int64_t z_stride, const float* table, int num_classes,
                                        int64_t N, int F, float* out, int32_t* status, const int32_t* dyn_N,
                                        hipStream_t stream) {
  if (N <= 0) return 0;
  if (N > 0x7FFFFFFF) return (int)hipErrorInvalidValue;
  const bool al16 = ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (al16 && (F == 32 || F == 64 || F == 128) && N * (F / 4) < ((int64_t)1 << 31)) {
    const dim3 grid(grid1d(N * (F / 4), 256)), block(256);
    if (F == 128) hipLaunchKernelGGL(k_embedding_fwd4<5>, grid, block, 0, stream, z, z_stride, table, num_classes, (int)N, out, status, dyn_N);
    else if (F == 64) hipLaunchKernelGGL(k_embedding_fwd4<4>, grid, block, 0, stream, z, z_stride, table, num_classes, (int)N, out, status, dyn_N);
    else hipLaunchKernelGGL(k_embedding_fwd4<3>, grid, block, 0, stream, z, z_stride, table, num_classes, (int)N, out, status, dyn_N);
    GEOSSL_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(k_embedding_fwd, dim3(grid1d(N * F, 256)), dim3(256), 0, stream, z, z_stride, table, num_classes, N,
                     F, out, status, dyn_N);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_embedding_fwd(const int64_t* z, int64_t z_stride, const float* table, int num_classes, int64_t N,
                                    int F, float* out, int32_t* status, hipStream_t stream) {
  return geossl_embedding_fwd_dyn(z, z_stride, table, num_classes, N, F, out, status, nullptr, stream);
}

extern "C" int64_t geossl_embedding_bwd_workspace_floats(int num_classes, int F) {
  return (int64_t)GEOSSL_EMB_CHUNKS * num_classes * F + 2 * GEOSSL_EMB_CHUNKS;  // partial tables, then the class bands
}

extern "C" int geossl_embedding_bwd(const int64_t* z, int64_t z_stride, const float* dh, int num_classes, int64_t N,
                                    int F, float* dtable, float* workspace, int accumulate, hipStream_t stream) {
  return geossl_embedding_bwd_dyn(z, z_stride, dh, num_classes, N, F, dtable, workspace, accumulate, nullptr, stream);
}

extern "C" int geossl_embedding_bwd_dyn(const int64_t* z, int64_t z_stride, const float* dh, int num_classes, int64_t N,
                                        int F, float* dtable, float* workspace, int accumulate, const int32_t* dyn_N,
                                        hipStream_t stream) {
  if (num_classes <= 0) return 0;
  if (N > 0x7FFFFFFF) return (int)hipErrorInvalidValue;
  if (F > 256 || (size_t)num_classes * F * sizeof(float) > 160 * 1024) return (int)hipErrorInvalidValue;
  int2* band = reinterpret_cast<int2*>(workspace + (size_t)GEOSSL_EMB_CHUNKS * num_classes * F);
  // Four row groups per chunk while their tables fit 48 KB of LDS (SchNet's 9 classes: 18 KB), else one.  Chunks of at
  // least 32 rows per group (a small batch: fewer chunks, a shorter list for the second stage), a multiple of 4.
  const int TF = (F + 63) / 64 * 64;
  const int ng = ((size_t)4 * num_classes * F * sizeof(float) <= 48 * 1024 && 4 * TF <= 1024) ? 4 : 1;
  int nchunks = (int)((N + 32 * ng - 1) / (32 * ng));
  nchunks = nchunks < 16 ? 16 : (nchunks > GEOSSL_EMB_CHUNKS ? GEOSSL_EMB_CHUNKS : (nchunks + 3) / 4 * 4);
  const size_t lds = (size_t)ng * num_classes * F * sizeof(float) + ng * sizeof(int2);
  if (ng == 4) {
    hipLaunchKernelGGL(k_embedding_bwd_partial<4>, dim3(nchunks), dim3(4 * TF), lds, stream, z, z_stride, dh, N, F,
                       num_classes, workspace, band, dyn_N);
  } else {
    allow_big_lds(&k_embedding_bwd_partial<1>);
    hipLaunchKernelGGL(k_embedding_bwd_partial<1>, dim3(nchunks), dim3(TF), lds, stream, z, z_stride, dh, N, F,
                       num_classes, workspace, band, dyn_N);
  }
  GEOSSL_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_embedding_bwd_reduce, dim3((F + 63) / 64, num_classes), dim3(256), 0, stream, workspace, band,
                     nchunks, num_classes, F, dtable, accumulate);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_segment_reduce_fwd(const float* h, const int32_t* mol_ptr, int64_t B, int F, int mean, float* out,
                                         hipStream_t stream) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(k_segment_reduce_fwd, dim3((unsigned)B), dim3(F < 256 ? ((F + 63) / 64 * 64) : 256), 0, stream, h,
                     mol_ptr, (int)B, F, mean, out);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_segment_reduce_bwd(const float* dout, const int32_t* mol_ptr, int64_t B, int F, int mean,
                                         float* dh, int accumulate, hipStream_t stream) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(k_segment_reduce_bwd, dim3((unsigned)B), dim3(F < 256 ? ((F + 63) / 64 * 64) : 256), 0, stream,
                     dout, mol_ptr, (int)B, F, mean, dh, accumulate);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
