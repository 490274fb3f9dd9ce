// Every index structure the PaiNN kernels read for the fused (clean | perturbed) batch of a DDM step, from the batch's
// precomputed radius_edge_index (Geom3D/datasets/datasets_3D_Radius.py:120, collated with node offsets by
// Geom3D/dataloaders/dataloaders_AtomTuple.py:64-65; the perturbed view keeps the clean view's graph,
// pretrain_GeoSSL.py:190-191) in ONE launch - what layout.EdgeLayout builds with two dozen small launches per batch
// object.  It is what lets a PaiNN step be a replayed graph over a shuffled loader: the captured kernels read these
// arrays from static (capacity-sized) buffers that this launch rewrites per step (geossl_amd/bucket.py).
//
// One 256-thread block per molecule; it writes the structures of BOTH views (view 1 = view 0 with the offsets N / E):
//   * the molecule's edge range [e0, e1) by a 256-ary search over the edge list's first row (edges are grouped by
//     molecule in batch order, both ends in one molecule: the collated output of the reference's dataset is);
//   * idx_i / idx_j of the two-view batch;
//   * the two incidence lists (edges by idx_i = row 0: the forward's scatter target, painn.py:59,61; edges by idx_j =
//     row 1: the backward's), each atom's edges in ascending edge order (fixed summation order downstream) - a stable
//     counting sort: every wave takes a contiguous quarter of the molecule's edges, counts per (quarter, atom) by
//     integer LDS atomics, offsets by a block scan, then each wave places its quarter 64 edges at a time, an edge's rank
//     among the equal keys of its chunk from ballots (one round per distinct key of the chunk);
//   * the row layout of the matrix-pipe interaction forward (painn_mma.hip): an atom's edges in incidence order padded
//     to groups of four rows; the groups of molecule m start at floor(first edge / 4) + first atom (a bound on the
//     groups of all molecules before it, so no scan over molecules is needed) and END at mol_grp_end[m].
// Integer work only; bit-reproducible.
#include "common.h"
#include "geossl_hip.h"

using namespace geossl;

namespace {

constexpr int PL_MAXN = 256;    // atoms per molecule

struct PainnLayoutArgs {
  const int64_t* src_i;   // [E] row 0 of radius_edge_index
  const int64_t* src_j;   // [E] row 1
  const int32_t* mol_ptr; // [B + 1] one-view molecule CSR
  int E, N, B, N2cap, pad;
  int64_t* idx_i2;        // [2E]
  int64_t* idx_j2;
  int64_t* iptr_i;        // [N2cap + 1]
  int32_t* ilist_i;       // [2E]
  int64_t* iptr_j;
  int32_t* ilist_j;
  int32_t* row_edge;      // [4 * groups]
  int32_t* grp_atom;      // [groups]
  int32_t* mol_grp;       // [2B + 1]
  int32_t* mol_grp_end;   // [2B]
  int32_t* status;        // set to 1 on an edge that leaves its molecule / a molecule above PL_MAXN atoms
};

// first e in [0, E) with src[e] >= key (src non-decreasing by molecule), for TWO keys at once, by all 256 threads of the
// block (the two searches share their rounds: the probes of a round are in flight together)
__device__ __forceinline__ void block_lower_bound2(const int64_t* __restrict__ src, int E, int64_t key0, int64_t key1,
                                                   int& out0, int& out1) {
  int lo[2] = {0, 0}, hi[2] = {E, E};
  bool done[2] = {E == 0, E == 0};
  const int64_t key[2] = {key0, key1};
  while (!(done[0] && done[1])) {
    int step[2], p[2];
    bool below[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int len = hi[q] - lo[q];
      step[q] = (len + 255) / 256;
      p[q] = lo[q] + (int)threadIdx.x * step[q];
      below[q] = !done[q] && p[q] < hi[q] && src[p[q]] < key[q];
    }
    const int c0 = __syncthreads_count(below[0] ? 1 : 0);
    const int c1 = __syncthreads_count(below[1] ? 1 : 0);
    const int c[2] = {c0, c1};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (done[q]) continue;
      if (c[q] == 0) {
        hi[q] = lo[q];
        done[q] = true;
      } else if (step[q] == 1) {
        lo[q] = hi[q] = lo[q] + c[q];
        done[q] = true;
      } else {
        const int nlo = lo[q] + (c[q] - 1) * step[q] + 1, nhi = min(hi[q], lo[q] + c[q] * step[q]);
        lo[q] = nlo;
        hi[q] = nhi;
        if (lo[q] >= hi[q]) done[q] = true;
      }
    }
  }
  out0 = lo[0];
  out1 = lo[1];
}

__global__ __launch_bounds__(256) void k_painn_edge_layout(PainnLayoutArgs A) {
  __shared__ int cnt[2][4][PL_MAXN];   // [side][wave][atom]: edges of the wave's quarter; then the quarter's first list slot
  __shared__ int pre_i[PL_MAXN], pre_j[PL_MAXN], pre_g[PL_MAXN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = blockIdx.x;
  const int a0 = A.mol_ptr[m], n = A.mol_ptr[m + 1] - a0;
  int e0, e1;
  block_lower_bound2(A.src_i, A.E, (int64_t)a0, (int64_t)a0 + n, e0, e1);
  const bool too_big = n > PL_MAXN;
  if (too_big && tid == 0) *A.status = 1;
  // The molecules' ranges [e0, e1) share their ends (the same key, the same search), so they tile [e0 of the first, e1 of
  // the last): an edge list that is not grouped by molecule in batch order leaves edges in front of or behind that span -
  // visited by no block, their slots would keep the previous step's data - or inside a wrong molecule's
  // range (caught below).
  if (tid == 0 && ((m == 0 && e0 != 0) || (m == (int)gridDim.x - 1 && e1 != A.E))) *A.status = 1;
  const int nn = too_big ? 0 : n;  // (a molecule above the limit gets empty lists; the status word reports it)
  const int N = A.N, E = A.E;
  for (int i = tid; i < 2 * 4 * PL_MAXN; i += 256) (&cnt[0][0][0])[i] = 0;
  __syncthreads();
  // the wave's quarter of the molecule's edges
  const int Em = e1 - e0, Q = (Em + 3) >> 2;
  const int q0 = e0 + min(Em, wave * Q), q1 = e0 + min(Em, (wave + 1) * Q);
  // ---- pass 1: two-view edge arrays + counts per (quarter, atom)
  int bad = 0;
  for (int e = q0 + lane; e < q1; e += 64) {
    const int64_t gi = A.src_i[e], gj = A.src_j[e];
    A.idx_i2[e] = gi;
    A.idx_j2[e] = gj;
    A.idx_i2[(size_t)E + e] = gi + N;
    A.idx_j2[(size_t)E + e] = gj + N;
    const int li = (int)(gi - a0), lj = (int)(gj - a0);
    if (li < 0 || li >= nn || lj < 0 || lj >= nn) {
      bad = 1;
    } else {
      atomicAdd(&cnt[0][wave][li], 1);
      atomicAdd(&cnt[1][wave][lj], 1);
    }
  }
  if (bad) *A.status = 1;
  __syncthreads();
  // ---- per atom: totals, exclusive prefixes over the molecule's atoms (edges by idx_i, by idx_j, groups of four rows -
  // at least one per atom), and the first list slot of every quarter
  int ci = 0, cj = 0;
  if (tid < nn) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      ci += cnt[0][w][tid];
      cj += cnt[1][w][tid];
    }
  }
  const int cg = tid < nn ? max(1, (ci + 3) >> 2) : 0;
  pre_i[tid] = ci;
  pre_j[tid] = cj;
  pre_g[tid] = cg;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {  // inclusive Hillis-Steele scans
    const int ai = tid >= o ? pre_i[tid - o] : 0, aj = tid >= o ? pre_j[tid - o] : 0, ag = tid >= o ? pre_g[tid - o] : 0;
    __syncthreads();
    pre_i[tid] += ai;
    pre_j[tid] += aj;
    pre_g[tid] += ag;
    __syncthreads();
  }
  const int G = pre_g[255];
  const int xi = pre_i[tid] - ci, xj = pre_j[tid] - cj, xg = pre_g[tid] - cg;  // exclusive
  __syncthreads();
  pre_i[tid] = xi;   // (from here on: the exclusive prefixes, read by the placement)
  pre_g[tid] = xg;
  if (tid < nn) {
    int si = xi, sj = xj;
#pragma unroll
    for (int w = 0; w < 4; ++w) {  // quarter w of atom tid starts behind the atom's edges of the quarters before it
      const int ti = cnt[0][w][tid], tj = cnt[1][w][tid];
      cnt[0][w][tid] = si;
      cnt[1][w][tid] = sj;
      si += ti;
      sj += tj;
    }
    A.iptr_i[a0 + tid] = (int64_t)e0 + xi;
    A.iptr_j[a0 + tid] = (int64_t)e0 + xj;
    A.iptr_i[N + a0 + tid] = (int64_t)E + e0 + xi;
    A.iptr_j[N + a0 + tid] = (int64_t)E + e0 + xj;
  }
  const int gbase0 = (e0 >> 2) + a0, gbase1 = ((E + e0) >> 2) + N + a0;
  if (tid == 0) {
    A.mol_grp[m] = gbase0;
    A.mol_grp_end[m] = gbase0 + G;
    A.mol_grp[A.B + m] = gbase1;
    A.mol_grp_end[A.B + m] = gbase1 + G;
  }
  if (m == A.B - 1) {  // the last molecule: the lists end here; atoms past the real count (a capacity) have none
    const int N2 = 2 * N;
    for (int a = N2 + tid; a <= A.N2cap; a += 256) {
      A.iptr_i[a] = 2 * (int64_t)E;
      A.iptr_j[a] = 2 * (int64_t)E;
    }
    if (tid == 0) A.mol_grp[2 * A.B] = ((2 * E) >> 2) + N2;
  }
  __syncthreads();
  // ---- pass 2: placement, a wave's quarter 64 edges at a time in ascending edge order.  An edge's slot in its atom's
  // list = (first slot of the wave's quarter for that atom, advanced chunk by chunk) + its rank among the chunk's edges
  // with the same key: one round per distinct key of the chunk (ballot of the lanes that hold it).
  volatile int* run_i = cnt[0][wave];
  volatile int* run_j = cnt[1][wave];
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int c0 = q0; c0 < q1; c0 += 64) {
    const int e = c0 + lane;
    const bool in = e < q1;
    int li = -1, lj = -1;
    if (in) {
      li = (int)(A.src_i[e] - a0);
      lj = (int)(A.src_j[e] - a0);
      if (li < 0 || li >= nn || lj < 0 || lj >= nn) li = lj = -1;
    }
    // rank of an edge among the chunk's edges with the same key + their number: one round of ballots per distinct key
    // (no memory access inside the rounds: a key's first free slot is read before them, advanced after them)
    auto ranks = [&](int key, int& rank, int& total) {
      unsigned long long rem = __ballot(key >= 0);
      rank = 0;
      total = 0;
      while (rem) {
        const int leader = __builtin_ctzll(rem);
        const int k = __builtin_amdgcn_readlane(key, leader);
        const unsigned long long mk = __ballot(key == k);
        if (key == k) {
          rank = __popcll(mk & lt);
          total = __popcll(mk);
        }
        rem &= ~mk;
      }
    };
    int ri, ti, rj, tj;
    const int bi = li >= 0 ? run_i[li] : 0, bj = lj >= 0 ? run_j[lj] : 0;   // first free slots (relative to e0)
    ranks(li, ri, ti);
    ranks(lj, rj, tj);
    if (li >= 0) {   // side i: the forward's lists and the group rows
      const int slot = bi + ri;
      A.ilist_i[(size_t)e0 + slot] = e;
      A.ilist_i[(size_t)E + e0 + slot] = E + e;
      const int r = slot - pre_i[li];                          // row of the atom's group layout
      A.row_edge[4 * ((size_t)gbase0 + pre_g[li]) + r] = e;
      A.row_edge[4 * ((size_t)gbase1 + pre_g[li]) + r] = E + e;
      if (ri == ti - 1) run_i[li] = bi + ti;                   // (the key's last edge of the chunk advances its slot)
    }
    if (lj >= 0) {   // side j: the backward's lists
      const int slot = bj + rj;
      A.ilist_j[(size_t)e0 + slot] = e;
      A.ilist_j[(size_t)E + e0 + slot] = E + e;
      if (rj == tj - 1) run_j[lj] = bj + tj;
    }
  }
  // ---- padding rows of every atom's last group, group codes (both views)
  if (tid < nn) {
    int32_t* rows0 = A.row_edge + 4 * ((size_t)gbase0 + xg);
    int32_t* rows1 = A.row_edge + 4 * ((size_t)gbase1 + xg);
    for (int r = ci; r < 4 * cg; ++r) {
      rows0[r] = -1;
      rows1[r] = -1;
    }
    for (int g = 0; g < cg; ++g) {
      const int last = g == cg - 1 ? 1 : 0;
      A.grp_atom[gbase0 + xg + g] = 2 * (a0 + tid) + last;
      A.grp_atom[gbase1 + xg + g] = 2 * (N + a0 + tid) + last;
    }
  }
}

}  // namespace

extern "C" int64_t geossl_painn_group_capacity(int64_t E2, int64_t N2) { return E2 / 4 + N2 + 1; }

extern "C" int geossl_painn_edge_layout(const int64_t* src_i, const int64_t* src_j, int64_t E, const int32_t* mol_ptr,
                                        int64_t N, int64_t B, int64_t N2cap, int64_t* idx_i2, int64_t* idx_j2,
                                        int64_t* iptr_i, int32_t* ilist_i, int64_t* iptr_j, int32_t* ilist_j,
                                        int32_t* row_edge, int32_t* grp_atom, int32_t* mol_grp, int32_t* mol_grp_end,
                                        int32_t* status, hipStream_t stream) {
  if (B <= 0) return 0;
  if (2 * E >= ((int64_t)1 << 30) || 2 * N >= ((int64_t)1 << 30) || N2cap < 2 * N) return (int)hipErrorInvalidValue;
  PainnLayoutArgs A{};
  A.src_i = src_i; A.src_j = src_j; A.mol_ptr = mol_ptr; A.E = (int)E; A.N = (int)N; A.B = (int)B; A.N2cap = (int)N2cap;
  A.idx_i2 = idx_i2; A.idx_j2 = idx_j2; A.iptr_i = iptr_i; A.ilist_i = ilist_i; A.iptr_j = iptr_j; A.ilist_j = ilist_j;
  A.row_edge = row_edge; A.grp_atom = grp_atom; A.mol_grp = mol_grp; A.mol_grp_end = mol_grp_end; A.status = status;
  hipLaunchKernelGGL(k_painn_edge_layout, dim3((unsigned)B), dim3(256), 0, stream, A);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
