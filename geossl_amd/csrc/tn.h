// Two-stage deterministic reductions shared by the weight-gradient kernels (wgrad.h, filter_bwd.hip, painn.hip):
// stage 1 writes one partial per row chunk / block, stage 2 (k_reduce_partials, gemm.hip) sums the partials in chunk
// order — no atomics, bit-reproducible.
#pragma once
#include "common.h"
#include "geossl_hip.h"

namespace geossl {

// out_z[(i/ncols)*ld + (i%ncols)*cstride] = (accumulate ? out : 0) + sum_b partial[z][b][i]
__global__ void k_reduce_partials(GeosslReduceBatch batch, const float* __restrict__ partial, int nblk, int len,
                                  int ncols, int ld, int cstride, int accumulate);

// Several reductions of one producer kernel in ONE launch (they share nblk and the z dimension): segment g covers
// blockIdx.x in [xoff[g], xoff[g+1]).
struct ReduceSeg {
  const float* partial;  // [z][nblk][len]
  int len, ncols, ld, cstride;
  float* out[GEOSSL_TN_MAX];
};
struct ReduceMulti {
  int nseg;
  int xoff[7];
  ReduceSeg seg[6];
  ReduceMulti() : nseg(0) { xoff[0] = 0; }
  void add(const float* partial, int len, int ncols, int ld, int cstride, float* const* out, int nz) {
    ReduceSeg& g = seg[nseg];
    g.partial = partial;
    g.len = len;
    g.ncols = ncols;
    g.ld = ld;
    g.cstride = cstride;
    for (int z = 0; z < GEOSSL_TN_MAX; ++z) g.out[z] = z < nz ? out[z] : nullptr;
    xoff[nseg + 1] = xoff[nseg] + (len + 63) / 64;
    ++nseg;
  }
  int blocks() const { return xoff[nseg]; }
};
__global__ void k_reduce_multi(ReduceMulti m, int nblk, int accumulate);
// the work of one 256-thread block of k_reduce_multi (64 outputs x 4 slices of the partial list, compensated; the four
// slice sums are combined in slice order); kernels that append reductions of their own to the batch call it
__device__ __forceinline__ void reduce_multi_block(const ReduceMulti& m, int nblk, int accumulate, float (*red)[64],
                                                   int z) {
  int g = 0;
  while (g + 1 < m.nseg && (int)blockIdx.x >= m.xoff[g + 1]) ++g;
  const ReduceSeg& sg = m.seg[g];
  float* out = sg.out[z];
  if (out == nullptr) return;
  const int len = sg.len;
  const float* p = sg.partial + (size_t)z * nblk * len;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = ((int)blockIdx.x - m.xoff[g]) * 64 + lane;
  const int per = (nblk + 3) / 4, b0 = slice * per, b1 = min(nblk, b0 + per);
  float s = 0.0f;
  if (i < len) s = kahan_sum_strided(p + i, b0, b1, len);
  red[slice][lane] = s;
  __syncthreads();
  if (slice == 0 && i < len) {
    const size_t o = (size_t)(i / sg.ncols) * sg.ld + (size_t)(i % sg.ncols) * sg.cstride;
    float v = accumulate ? out[o] : 0.0f;
    v += red[0][lane];
    v += red[1][lane];
    v += red[2][lane];
    v += red[3][lane];
    out[o] = v;
  }
}

inline int64_t tn_workspace_floats(int64_t R, int M, int N, int nprob) {
  int chunk, nblk;
  geossl_tn_plan(R, nprob, &chunk, &nblk);
  return (int64_t)nprob * nblk * ((int64_t)M * N + 2 * M);
}

}  // namespace geossl
