// Every index structure the PaiNN kernels read for the fused (clean | perturbed) batch of a DDM step, from the batch's
// precomputed radius_edge_index (Geom3D/datasets/datasets_3D_Radius.py:120, collated with node offsets by
// Geom3D/dataloaders/dataloaders_AtomTuple.py:64-65; the perturbed view keeps the clean view's graph,
// pretrain_GeoSSL.py:190-191) in ONE launch - what layout.EdgeLayout builds with two dozen small launches per batch
// object.  It is what lets a PaiNN step be a replayed graph over a shuffled loader: the captured kernels read these
// arrays from static (capacity-sized) buffers that this launch rewrites per step (geossl_amd/bucket.py).
//
// One 256-thread block per molecule of the two-view batch (2B blocks; view v = block / B reads the molecule's edges of
// the one-view list and writes them with the offsets v N / v E):
//   * the molecule's edge range [e0, e1) by a 256-ary search over the edge list's first row (edges are grouped by
//     molecule in batch order, both ends in one molecule: the collated output of the reference's dataset is);
//   * idx_i / idx_j of the two-view batch;
//   * the two incidence lists (edges by idx_i = row 0: the forward's scatter target, painn.py:59,61; edges by idx_j =
//     row 1: the backward's), each atom's edges in ascending edge order (fixed summation order downstream): counts by
//     integer LDS atomics, offsets by a block scan, the lists by one thread per atom walking the molecule's edges (keys
//     staged in LDS);
//   * the row layout of the matrix-pipe interaction forward (painn_mma.hip): an atom's edges in incidence order padded
//     to groups of four rows; the groups of molecule m start at floor(first edge / 4) + first atom (a bound on the
//     groups of all molecules before it, so no scan over molecules is needed) and END at mol_grp_end[m].
// Integer work only; bit-reproducible.
#include "common.h"
#include "geossl_hip.h"

using namespace geossl;

namespace {

constexpr int PL_MAXN = 256;    // atoms per molecule
constexpr int PL_CHUNK = 4096;  // edges staged per pass (local atom indices as 16-bit keys)

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

// first e in [0, E) with src[e] >= key (src non-decreasing by molecule), by all 256 threads of the block
__device__ __forceinline__ int block_lower_bound(const int64_t* __restrict__ src, int E, int64_t key) {
  int lo = 0, hi = E;
  while (lo < hi) {
    const int len = hi - lo, step = (len + 255) / 256;
    const int p = lo + (int)threadIdx.x * step;
    const bool below = p < hi && src[p] < key;
    const int c = __syncthreads_count(below ? 1 : 0);
    if (c == 0) return lo;
    if (step == 1) return lo + c;
    const int nlo = lo + (c - 1) * step + 1, nhi = min(hi, lo + c * step);
    lo = nlo;
    hi = nhi;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_painn_edge_layout(PainnLayoutArgs A) {
  __shared__ int cnt_i[PL_MAXN], cnt_j[PL_MAXN];
  __shared__ int pre_i[PL_MAXN], pre_j[PL_MAXN], pre_g[PL_MAXN];
  __shared__ uint16_t key_i[PL_CHUNK], key_j[PL_CHUNK];
  const int tid = threadIdx.x;
  const int mm = blockIdx.x, v = mm >= A.B ? 1 : 0, m = mm - v * A.B;
  const int a0 = A.mol_ptr[m], n = A.mol_ptr[m + 1] - a0;
  const int offN = v * A.N, offE = v * A.E;
  const int e0 = block_lower_bound(A.src_i, A.E, (int64_t)a0);
  const int e1 = block_lower_bound(A.src_i, A.E, (int64_t)a0 + n);
  const bool too_big = n > PL_MAXN;
  if (too_big && tid == 0) *A.status = 1;
  const int nn = too_big ? 0 : n;  // (a molecule above the limit gets empty lists; the status word reports it)
  cnt_i[tid] = 0;
  cnt_j[tid] = 0;
  __syncthreads();
  // ---- two-view edge arrays + per-atom counts
  int bad = 0;
  for (int e = e0 + tid; e < e1; e += 256) {
    const int64_t gi = A.src_i[e], gj = A.src_j[e];
    A.idx_i2[(size_t)offE + e] = gi + offN;
    A.idx_j2[(size_t)offE + e] = gj + offN;
    const int li = (int)(gi - a0), lj = (int)(gj - a0);
    if (li < 0 || li >= nn || lj < 0 || lj >= nn) {
      bad = 1;
    } else {
      atomicAdd(&cnt_i[li], 1);
      atomicAdd(&cnt_j[lj], 1);
    }
  }
  if (bad) *A.status = 1;
  __syncthreads();
  // ---- exclusive prefixes over the molecule's atoms: edges by idx_i, by idx_j, groups of four rows (at least one per atom)
  const int ci = tid < nn ? cnt_i[tid] : 0, cj = tid < nn ? cnt_j[tid] : 0;
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
  const int gbase = ((offE + e0) >> 2) + offN + a0;
  if (tid < nn) {
    A.iptr_i[offN + a0 + tid] = (int64_t)offE + e0 + xi;
    A.iptr_j[offN + a0 + tid] = (int64_t)offE + e0 + xj;
  }
  if (tid == 0) {
    A.mol_grp[mm] = gbase;
    A.mol_grp_end[mm] = gbase + G;
  }
  if (mm == 2 * A.B - 1) {  // the last molecule: the lists end here; atoms past the real count (a capacity) have none
    const int N2 = 2 * A.N;
    for (int a = N2 + tid; a <= A.N2cap; a += 256) {
      A.iptr_i[a] = 2 * (int64_t)A.E;
      A.iptr_j[a] = 2 * (int64_t)A.E;
    }
    if (tid == 0) A.mol_grp[2 * A.B] = ((2 * A.E) >> 2) + N2;
  }
  // ---- the lists: thread a walks the molecule's edges in ascending order (keys from LDS, a chunk at a time)
  int pi = 0, pj = 0;  // entries written so far
  int32_t* rows = A.row_edge + 4 * ((size_t)gbase + xg);  // this atom's rows of the group layout
  for (int c0 = e0; c0 < e1; c0 += PL_CHUNK) {
    const int len = min(PL_CHUNK, e1 - c0);
    __syncthreads();
    for (int k = tid; k < len; k += 256) {
      const int li = (int)(A.src_i[c0 + k] - a0), lj = (int)(A.src_j[c0 + k] - a0);
      const bool ok = li >= 0 && li < nn && lj >= 0 && lj < nn;
      key_i[k] = ok ? (uint16_t)li : (uint16_t)0xFFFF;
      key_j[k] = ok ? (uint16_t)lj : (uint16_t)0xFFFF;
    }
    __syncthreads();
    if (tid < nn) {
      for (int k = 0; k < len; ++k) {
        const int e = offE + c0 + k;
        if (key_i[k] == tid) {
          A.ilist_i[(size_t)offE + e0 + xi + pi] = e;
          rows[pi] = e;
          ++pi;
        }
        if (key_j[k] == tid) {
          A.ilist_j[(size_t)offE + e0 + xj + pj] = e;
          ++pj;
        }
      }
    }
  }
  if (tid < nn) {
    for (int r = ci; r < 4 * cg; ++r) rows[r] = -1;  // padding rows of the atom's last group
    for (int g = 0; g < cg; ++g) A.grp_atom[gbase + xg + g] = 2 * (offN + a0 + tid) + (g == cg - 1 ? 1 : 0);
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
  hipLaunchKernelGGL(k_painn_edge_layout, dim3((unsigned)(2 * B)), dim3(256), 0, stream, A);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
