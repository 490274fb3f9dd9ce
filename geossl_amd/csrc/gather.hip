// Batch assembly on the device from a device-resident dataset of molecules.
//
// The reference's loader (examples/pretrain_GeoSSL.py:295-301: DataLoaderAtomTuple(dataset, batch_size, shuffle=True))
// collates a random subset of molecules on the host every step: BatchAtomTuple.from_data_list concatenates x /
// positions, writes batch = full((n_i,), i) and adds the cumulative node offset to super_edge_index /
// radius_edge_index (Geom3D/dataloaders/dataloaders_AtomTuple.py:46-73), after AtomTupleExtractor enumerated the atom
// tuples per molecule (:15-37).  Here the molecules live in HBM (concatenated x / positions, for PaiNN the concatenated
// per-molecule radius_edge_index of datasets_3D_Radius.py:120) and ONE launch writes a permutation slice straight into
// the static inputs of a replayed step graph: one block per chosen molecule
//   * gathers its atom rows (x, positions) and writes the batch vector,
//   * enumerates its super-edges (itertools.combinations / permutations order, node offset added) - the tuples are a
//     function of the atom count, nothing is read,
//   * writes the pair-slot atoms of the two-view batch (what geossl_pair_index_fill produces) and the atom -> incident
//     super-edge lists of the heads (what geossl_incidence_fill finds by searching; here in closed form),
//   * copies its radius edges with the node offset changed from the dataset's to the batch's;
// the blocks behind the last molecule clear a float buffer (the owner's flat gradient buffer).  All offsets come from
// the host's size table (cumulative sums over B integers, uploaded with the step's other pointer arrays): no read-back.
// HBM-bound integer work: a few hundred KB per step, one wave-coalesced store stream per array.
#include "common.h"
#include "geossl_hip.h"

namespace {

constexpr int GATHER_THREADS = 256;

// slot p of the lexicographic i<j enumeration of n atoms -> (a, b);  row(a) = a n - a (a + 1) / 2 - a - 1, slot = row(a) + b
__device__ __forceinline__ void pair_of_slot(int p, int n, int& a, int& b) {
  const float t = (float)(2 * n - 1);
  int g = (int)((t - sqrtf(fmaxf(t * t - 8.0f * (float)p, 0.0f))) * 0.5f);
  g = g < 0 ? 0 : (g > n - 2 ? n - 2 : g);
  // first slot of row a: a n - a (a + 1) / 2 (the float estimate is off by at most one)
  while (g > 0 && g * n - g * (g + 1) / 2 > p) --g;
  while (g < n - 2 && (g + 1) * n - (g + 1) * (g + 2) / 2 <= p) ++g;
  a = g;
  b = p - (g * n - g * (g + 1) / 2) + g + 1;
}

__global__ __launch_bounds__(GATHER_THREADS) void k_gather_molecules(GeosslGather g, int B) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= B) {  // ---- the blocks behind the molecules: clear the float buffer
    if (g.zero == nullptr) return;
    const int64_t nb = (int64_t)gridDim.x - B;
    const int64_t t0 = ((int64_t)blockIdx.x - B) * GATHER_THREADS + tid, nt = nb * GATHER_THREADS;
    const int64_t n4 = g.zero_count >> 2;
    f32x4* z4 = reinterpret_cast<f32x4*>(g.zero);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if ((reinterpret_cast<uintptr_t>(g.zero) & 15) == 0) {
      for (int64_t i = t0; i < n4; i += nt) z4[i] = zero4;
      for (int64_t i = (n4 << 2) + t0; i < g.zero_count; i += nt) g.zero[i] = 0.f;
    } else {
      for (int64_t i = t0; i < g.zero_count; i += nt) g.zero[i] = 0.f;
    }
    return;
  }
  const int m = blockIdx.x;
  const int a0 = g.mol_ptr[m], n = g.mol_ptr[m + 1] - a0;
  const int64_t s0 = g.src_off[m];
  // ---- atom rows
  {
    const int C = g.x_cols;
    const int64_t* xs = g.x_src + s0 * C;
    int64_t* xd = g.x_dst + (int64_t)a0 * C;
    for (int i = tid; i < n * C; i += GATHER_THREADS) xd[i] = xs[i];
    const float* ps = g.pos_src + s0 * 3;
    float* pd = g.pos_dst + (int64_t)a0 * 3;
    for (int i = tid; i < n * 3; i += GATHER_THREADS) pd[i] = ps[i];
    if (g.batch_dst != nullptr)
      for (int i = tid; i < n; i += GATHER_THREADS) g.batch_dst[a0 + i] = m;
  }
  const int P = n * (n - 1) / 2;
  // ---- pair slots of the two-view batch (geossl_pair_index_fill) and, for "combination", the super-edges
  const bool comb_sei = g.sei0 != nullptr && g.option == 0;
  if (g.pair_i != nullptr || comb_sei) {
    const int N1 = g.mol_ptr[B];
    const int pp0 = g.pair_i != nullptr ? g.pair_ptr2[m] : 0, pp1 = g.pair_i != nullptr ? g.pair_ptr2[B + m] : 0;
    const int64_t se0 = comb_sei ? (int64_t)g.se_ptr[m] : 0;
    for (int p = tid; p < P; p += GATHER_THREADS) {
      int a, b;
      pair_of_slot(p, n, a, b);
      if (g.pair_i != nullptr) {
        g.pair_i[pp0 + p] = a0 + a;
        g.pair_j[pp0 + p] = a0 + b;
        g.pair_i[pp1 + p] = N1 + a0 + a;
        g.pair_j[pp1 + p] = N1 + a0 + b;
      }
      if (comb_sei) {
        g.sei0[se0 + p] = a0 + a;
        g.sei1[se0 + p] = a0 + b;
      }
    }
  }
  if (g.sei0 != nullptr && g.option == 1) {  // itertools.permutations order: (a, b), b != a, column a (n - 1) + (b < a ? b : b - 1)
    const int64_t se0 = g.se_ptr[m];
    for (int q = tid; q < 2 * P; q += GATHER_THREADS) {
      const int a = q / (n - 1), r = q - a * (n - 1), b = r < a ? r : r + 1;
      g.sei0[se0 + q] = a0 + a;
      g.sei1[se0 + q] = a0 + b;
    }
  }
  // ---- atom -> incident super-edges, ordered by super-edge id (geossl_incidence_fill, sides = 3)
  if (g.inc_idx != nullptr && n > 1) {
    const int se0 = g.se_ptr[m];
    const int64_t base = g.inc_ptr[a0];
    if (g.option == 0) {
      const int cnt = n - 1;           // atom k lies on (e, k) for e < k, then on (k, e + 1) for e >= k
      for (int f = tid; f < n * cnt; f += GATHER_THREADS) {
        const int k = f / cnt, e = f - k * cnt;
        const int a = e < k ? e : k, b = e < k ? k : e + 1;
        g.inc_idx[base + f] = se0 + a * n - a * (a + 1) / 2 - a - 1 + b;
      }
    } else {
      const int cnt = 2 * (n - 1);     // (e, k) for e < k; the n - 1 tuples (k, .); (e', k) for e' > k
      for (int f = tid; f < n * cnt; f += GATHER_THREADS) {
        const int k = f / cnt, e = f - k * cnt;
        int id;
        if (e < k) id = e * (n - 1) + k - 1;
        else if (e < k + n - 1) id = k * (n - 1) + (e - k);
        else id = (e - (n - 1) + 1) * (n - 1) + k;
        g.inc_idx[base + f] = se0 + id;
      }
    }
  }
  // ---- radius edges of the molecule: the dataset's node offset replaced by the batch's
  if (g.e0_dst != nullptr) {
    const int d0 = g.e_ptr[m], cnt = g.e_ptr[m + 1] - d0;
    const int64_t es = g.e_src_off[m];
    const int64_t shift = (int64_t)a0 - s0;
    for (int i = tid; i < cnt; i += GATHER_THREADS) {
      g.e0_dst[d0 + i] = g.e0_src[es + i] + shift;
      g.e1_dst[d0 + i] = g.e1_src[es + i] + shift;
    }
  }
}

}  // namespace

extern "C" int geossl_gather_molecules(const GeosslGather* g, int64_t B, hipStream_t stream) {
  if (g == nullptr || B < 0 || B > (1 << 24)) return (int)hipErrorInvalidValue;
  if (B > 0 && (g->x_src == nullptr || g->pos_src == nullptr || g->src_off == nullptr || g->mol_ptr == nullptr ||
                g->x_dst == nullptr || g->pos_dst == nullptr || g->x_cols < 1 || (g->option != 0 && g->option != 1)))
    return (int)hipErrorInvalidValue;
  if ((g->sei0 != nullptr) != (g->sei1 != nullptr) || (g->pair_i != nullptr) != (g->pair_j != nullptr) ||
      (g->e0_dst != nullptr) != (g->e1_dst != nullptr))
    return (int)hipErrorInvalidValue;
  if ((g->sei0 != nullptr || g->inc_idx != nullptr) && g->se_ptr == nullptr) return (int)hipErrorInvalidValue;
  if (g->pair_i != nullptr && g->pair_ptr2 == nullptr) return (int)hipErrorInvalidValue;
  if (g->inc_idx != nullptr && g->inc_ptr == nullptr) return (int)hipErrorInvalidValue;
  if (g->e0_dst != nullptr && (g->e0_src == nullptr || g->e1_src == nullptr || g->e_src_off == nullptr || g->e_ptr == nullptr))
    return (int)hipErrorInvalidValue;
  if (g->zero_count < 0 || (g->zero == nullptr && g->zero_count > 0) || (reinterpret_cast<uintptr_t>(g->zero) & 3))
    return (int)hipErrorInvalidValue;
  int zero_blocks = 0;
  if (g->zero != nullptr && g->zero_count > 0) {
    const int64_t want = (g->zero_count / 4 + GATHER_THREADS - 1) / GATHER_THREADS;   // one float4 per thread and trip
    zero_blocks = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
  }
  if (B + zero_blocks == 0) return 0;
  hipLaunchKernelGGL(k_gather_molecules, dim3((unsigned)(B + zero_blocks)), dim3(GATHER_THREADS), 0, stream, *g, (int)B);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
