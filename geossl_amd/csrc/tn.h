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

inline int64_t tn_workspace_floats(int64_t R, int M, int N, int nprob) {
  int chunk, nblk;
  geossl_tn_plan(R, nprob, &chunk, &nblk);
  return (int64_t)nprob * nblk * ((int64_t)M * N + 2 * M);
}

}  // namespace geossl
