// PaiNN edge and atom kernels (BASELINE config 5): replaces, for the hot path, the per-edge arithmetic of
// PaiNN.forward / PaiNNInteraction / PaiNNMixing (Geom3D/models/painn.py:32-66,91-114,230-253) and their autograd.
// The Dense layers around them run on geossl_linear / geossl_linear_wgrad.
//
// Edge work is organised per atom through two incidence lists built once per batch from radius_edge_index
// (idx_i = row 0, idx_j = row 1): edges by idx_i (forward scatter, painn.py:59,61) and edges by idx_j (backward).
// One block per atom, one thread per feature: the filter value W_ij = (phi(d) W^T + b) * fcut of the thread's three
// channels is recomputed from the 20 radial basis values on the fly (never stored per edge), messages are summed in
// edge order in registers — no atomics, bit-reproducible.
#include <cstdlib>

#include "common.h"
#include "geossl_hip.h"
#include "tn.h"

using namespace geossl;

namespace {

constexpr int RMAX = 32;  // max radial basis functions

inline int grid1d(int64_t n, int block, int cap = 4096) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// painn.py:232-239 + painn_utils.py:99-103,152-154
__global__ void k_painn_edge_geom(const float* __restrict__ pos, const int64_t* __restrict__ idx_i,
                                  const int64_t* __restrict__ idx_j, int64_t E, float cutoff,
                                  const float* __restrict__ offsets, const float* __restrict__ widths, int R,
                                  float* __restrict__ dir, float* __restrict__ fcut, float* __restrict__ phi,
                                  const int32_t* __restrict__ dyn_E) {
  E = dyn_count((int)E, dyn_E);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx_i[e], j = idx_j[e];
    const float rx = pos[3 * i] - pos[3 * j], ry = pos[3 * i + 1] - pos[3 * j + 1], rz = pos[3 * i + 2] - pos[3 * j + 2];
    const float d = sqrtf(norm2_rn(rx, ry, rz));
    dir[3 * e] = rx / d;
    dir[3 * e + 1] = ry / d;
    dir[3 * e + 2] = rz / d;
    const float c = 0.5f * (cosf(mul_rn(d, GEOSSL_PI_F) / cutoff) + 1.0f);
    fcut[e] = d < cutoff ? c : 0.0f;
    for (int r = 0; r < R; ++r) {
      const float w = widths[r];
      const float coeff = -0.5f / (w * w);
      const float diff = d - offsets[r];
      phi[e * R + r] = expf(coeff * (diff * diff));
    }
  }
}

// silu (F.silu) and its derivative
__global__ void k_silu_fwd(const float* __restrict__ u, int64_t n, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = u[i];
    y[i] = x / (1.0f + expf(-x));
  }
}
__global__ void k_silu_bwd(const float* __restrict__ u, const float* __restrict__ dy, int64_t n,
                           float* __restrict__ du) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = u[i], s = 1.0f / (1.0f + expf(-x));
    du[i] = dy[i] * (s * (1.0f + x * (1.0f - s)));
  }
}

// Per-wave stage of the edges of one atom: lane k fetches the data of the k-th edge of the chunk with ordinary vector
// loads - its radial-basis row, cutoff value, direction and the atom at the other end - and writes them as one row of
// EROW floats; the edge loop then reads a row with broadcast LDS reads.  (Fetched through the scalar unit inside the
// loop, every edge paid a chain of L2 round trips: index -> row -> LDS address; the loop ran at ~1.2 us per edge.)
constexpr int ECHUNK = 16;               // edges per chunk (an atom has ~16 incoming edges at 5 A)
template <int R>
struct EdgeStage {
  static constexpr int EROW = ((R + 5 + 3) / 4) * 4;  // phi[R], fcut, dir[3], other atom (int bits), padded to 16 bytes
  // lanes 0 .. cnt-1: edge inc_idx[pc + lane] -> row `lane` of the wave's stage
  static __device__ __forceinline__ void fill(float* __restrict__ stage, int lane, int cnt, int64_t pc,
                                              const int32_t* __restrict__ inc_idx, const int64_t* __restrict__ idx_other,
                                              const float* __restrict__ phi, const float* __restrict__ fcut,
                                              const float* __restrict__ dir) {
    if (lane < cnt) {
      const int e = inc_idx[pc + lane];
      float* row = stage + lane * EROW;
      const float* __restrict__ p = phi + (size_t)e * R;
      if constexpr (R % 4 == 0) {
#pragma unroll
        for (int r = 0; r < R; r += 4) *reinterpret_cast<f32x4*>(row + r) = *reinterpret_cast<const f32x4*>(p + r);
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) row[r] = p[r];
      }
      row[R] = fcut[e];
      row[R + 1] = dir[3 * e];
      row[R + 2] = dir[3 * e + 1];
      row[R + 3] = dir[3 * e + 2];
      row[R + 4] = __int_as_float((int)idx_other[e]);
    }
  }
};

// ------------------------------------------------------------------------------------ interaction, forward
// q_out[i] = q[i] + sum_e dq_e, mu_out[i] = mu[i] + sum_e (dmuR_e * dir_e + dmumu_e * mu[j_e]),  e over edges with
// idx_i[e] == i in ascending e;  [dq, dmuR, dmumu]_e = W_e * x[j_e]  (painn.py:54-64)
template <int R>
__global__ __launch_bounds__(128) void k_painn_interaction_fwd(
    const float* __restrict__ q, const float* __restrict__ mu, const float* __restrict__ xc,
    const int64_t* __restrict__ idx_j, const int64_t* __restrict__ inc_ptr, const int32_t* __restrict__ inc_idx,
    const float* __restrict__ phi, const float* __restrict__ fcut, const float* __restrict__ dir,
    const float* __restrict__ Wf, const float* __restrict__ bf, int N, int F, float* __restrict__ q_out,
    float* __restrict__ mu_out, const int32_t* __restrict__ atom_list, const int32_t* __restrict__ dyn_nlist) {
  // atom_list == NULL: atoms 0 .. N-1 (one per block); else the atoms atom_list[0 .. min(N, *dyn_nlist)) - the atoms of
  // the molecules that a molecule-staged launch skipped (more atoms than its LDS rows), blocks striding over the list
  const int f = threadIdx.x;
  if (f >= F) return;
  const int cnt = atom_list != nullptr ? dyn_count(N, dyn_nlist) : N;
  if ((int)blockIdx.x >= cnt) return;
  float w0[R], w1[R], w2[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    w0[r] = Wf[(size_t)f * R + r];
    w1[r] = Wf[(size_t)(F + f) * R + r];
    w2[r] = Wf[(size_t)(2 * F + f) * R + r];
  }
  const float b0 = bf[f], b1 = bf[F + f], b2 = bf[2 * F + f];
  // Round 5: the data of an atom's edges through the per-wave stage of the molecule kernels (one edge per lane, vector
  // loads, broadcast LDS reads in the loop) and the source rows of edge k + 1 requested while edge k is multiplied.
  // Fetched through the scalar unit inside the loop, an edge cost a chain of three dependent round trips (list entry
  // -> source atom -> rows): 2.5 us per edge, 80 us for an atom with 32 edges whatever the size of the launch - the
  // time of a whole launch at the reference's batch size (set C, 2 x 128 molecules: 106 us -> see DESIGN 3.2).
  __shared__ __attribute__((aligned(16))) float estage_s[2 * ECHUNK * EdgeStage<R>::EROW];
  float* estage = estage_s + (threadIdx.x >> 6) * (ECHUNK * EdgeStage<R>::EROW);
  const int lane_ = threadIdx.x & 63;
  for (int k = blockIdx.x; k < cnt; k += gridDim.x) {
  const int i = atom_list != nullptr ? atom_list[k] : k;
  float dq = 0.0f, dm0 = 0.0f, dm1 = 0.0f, dm2 = 0.0f;
  const int64_t p0 = inc_ptr[i], p1 = inc_ptr[i + 1];
  for (int64_t pc = p0; pc < p1; pc += ECHUNK) {
    const int cntc = (int)min((int64_t)ECHUNK, p1 - pc);
    EdgeStage<R>::fill(estage, lane_, cntc, pc, inc_idx, idx_j, phi, fcut, dir);
    float nx0, nx1, nx2, nm0, nm1, nm2;
    {
      const int j0 = __float_as_int(estage[R + 4]);
      const float* __restrict__ xj = xc + (size_t)j0 * 3 * F;
      const float* __restrict__ mj = mu + (size_t)j0 * 3 * F;
      nx0 = xj[f]; nx1 = xj[F + f]; nx2 = xj[2 * F + f];
      nm0 = mj[f]; nm1 = mj[F + f]; nm2 = mj[2 * F + f];
    }
    // (four edges' rows at a time, the next four in flight: 152 registers instead of 116, three waves per SIMD instead
    // of four - 80 against 71 us on the set-C launch; the issue floor of the filter arithmetic is ~45 us)
    for (int kk = 0; kk < cntc; ++kk) {
      const float* row = estage + kk * EdgeStage<R>::EROW;
      const float xj0 = nx0, xj1 = nx1, xj2 = nx2, mj0 = nm0, mj1 = nm1, mj2 = nm2;
      if (kk + 1 < cntc) {  // (uniform) the next edge's source rows: in flight during this edge's arithmetic
        const int jn = __float_as_int(row[EdgeStage<R>::EROW + R + 4]);
        const float* __restrict__ xj = xc + (size_t)jn * 3 * F;
        const float* __restrict__ mj = mu + (size_t)jn * 3 * F;
        nx0 = xj[f]; nx1 = xj[F + f]; nx2 = xj[2 * F + f];
        nm0 = mj[f]; nm1 = mj[F + f]; nm2 = mj[2 * F + f];
      }
      float W0 = b0, W1 = b1, W2 = b2;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float pr = row[r];
        W0 = fmaf(pr, w0[r], W0);
        W1 = fmaf(pr, w1[r], W1);
        W2 = fmaf(pr, w2[r], W2);
      }
      const float fc = row[R];
      W0 *= fc; W1 *= fc; W2 *= fc;                                  // painn.py:241
      const float x0 = W0 * xj0, x1 = W1 * xj1, x2 = W2 * xj2;       // :56
      dq += x0;                                                      // :59
      dm0 += x1 * row[R + 1] + x2 * mj0;                             // :60-61
      dm1 += x1 * row[R + 2] + x2 * mj1;
      dm2 += x1 * row[R + 3] + x2 * mj2;
    }
  }
  q_out[(size_t)i * F + f] = q[(size_t)i * F + f] + dq;            // :63
  float* mo = mu_out + (size_t)i * 3 * F;
  const float* mi = mu + (size_t)i * 3 * F;
  mo[f] = mi[f] + dm0;                                             // :64
  mo[F + f] = mi[F + f] + dm1;
  mo[2 * F + f] = mi[2 * F + f] + dm2;
  }
}

// ----------------------------------------------------------------------------------- interaction, backward
// Persistent blocks over source atoms j (edges with idx_j[e] == j, ascending e).  Per atom: gradient of the context
// features x[j] and of mu[j]; across all atoms of the block: the filter-network weight/bias gradient partials.
template <int R>
__global__ __launch_bounds__(128) void k_painn_interaction_bwd(
    const float* __restrict__ dq_out, const float* __restrict__ dmu_out, const float* __restrict__ mu,
    const float* __restrict__ xc, const int64_t* __restrict__ idx_i, const int64_t* __restrict__ inc_ptr,
    const int32_t* __restrict__ inc_idx, const float* __restrict__ phi, const float* __restrict__ fcut,
    const float* __restrict__ dir, const float* __restrict__ Wf, const float* __restrict__ bf, int N, int F,
    float* __restrict__ dxc, float* __restrict__ dmu_in, float* __restrict__ partial_w, float* __restrict__ partial_b,
    const int32_t* __restrict__ atom_list, const int32_t* __restrict__ dyn_nlist) {
  const int f = threadIdx.x;
  if (f >= F) return;
  const int cnt = atom_list != nullptr ? dyn_count(N, dyn_nlist) : N;  // (a list: see k_painn_interaction_fwd)
  float w0[R], w1[R], w2[R], g0[R], g1[R], g2[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    w0[r] = Wf[(size_t)f * R + r];
    w1[r] = Wf[(size_t)(F + f) * R + r];
    w2[r] = Wf[(size_t)(2 * F + f) * R + r];
    g0[r] = g1[r] = g2[r] = 0.0f;
  }
  const float b0 = bf[f], b1 = bf[F + f], b2 = bf[2 * F + f];
  float gb0 = 0.0f, gb1 = 0.0f, gb2 = 0.0f;
  __shared__ __attribute__((aligned(16))) float estage_s[2 * ECHUNK * EdgeStage<R>::EROW];
  float* estage = estage_s + (threadIdx.x >> 6) * (ECHUNK * EdgeStage<R>::EROW);
  const int lane_ = threadIdx.x & 63;
  for (int k = blockIdx.x; k < cnt; k += gridDim.x) {
    const int j = atom_list != nullptr ? atom_list[k] : k;
    const float* __restrict__ xj = xc + (size_t)j * 3 * F;
    const float* __restrict__ mj = mu + (size_t)j * 3 * F;
    const float xj0 = xj[f], xj1 = xj[F + f], xj2 = xj[2 * F + f];
    const float m0 = mj[f], m1 = mj[F + f], m2 = mj[2 * F + f];
    float dx0 = 0.0f, dx1 = 0.0f, dx2 = 0.0f, dmj0 = 0.0f, dmj1 = 0.0f, dmj2 = 0.0f;
    const int64_t p0 = inc_ptr[j], p1 = inc_ptr[j + 1];
    for (int64_t pc = p0; pc < p1; pc += ECHUNK) {   // (staged edges, next edge's rows in flight: see the forward kernel)
      const int cntc = (int)min((int64_t)ECHUNK, p1 - pc);
      EdgeStage<R>::fill(estage, lane_, cntc, pc, inc_idx, idx_i, phi, fcut, dir);
      float ngq, ng0, ng1, ng2;
      {
        const int i0 = __float_as_int(estage[R + 4]);
        const float* __restrict__ gm = dmu_out + (size_t)i0 * 3 * F;
        ngq = dq_out[(size_t)i0 * F + f];
        ng0 = gm[f]; ng1 = gm[F + f]; ng2 = gm[2 * F + f];
      }
      for (int kk = 0; kk < cntc; ++kk) {
        const float* row = estage + kk * EdgeStage<R>::EROW;
        const float gq = ngq, gm0 = ng0, gm1 = ng1, gm2 = ng2;
        if (kk + 1 < cntc) {
          const int in_ = __float_as_int(row[EdgeStage<R>::EROW + R + 4]);
          const float* __restrict__ gm = dmu_out + (size_t)in_ * 3 * F;
          ngq = dq_out[(size_t)in_ * F + f];
          ng0 = gm[f]; ng1 = gm[F + f]; ng2 = gm[2 * F + f];
        }
        float pr[R];
        float W0 = b0, W1 = b1, W2 = b2;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          pr[r] = row[r];
          W0 = fmaf(pr[r], w0[r], W0);
          W1 = fmaf(pr[r], w1[r], W1);
          W2 = fmaf(pr[r], w2[r], W2);
        }
        const float fc = row[R];
        W0 *= fc; W1 *= fc; W2 *= fc;
        const float s1 = gm0 * row[R + 1] + gm1 * row[R + 2] + gm2 * row[R + 3];
        const float s2 = gm0 * m0 + gm1 * m1 + gm2 * m2;
        dx0 = fmaf(gq, W0, dx0);
        dx1 = fmaf(s1, W1, dx1);
        dx2 = fmaf(s2, W2, dx2);
        const float x2 = W2 * xj2;
        dmj0 = fmaf(gm0, x2, dmj0);
        dmj1 = fmaf(gm1, x2, dmj1);
        dmj2 = fmaf(gm2, x2, dmj2);
        // W_c = (b_c + sum_r phi_r w_c[r]) * fcut
        const float t0 = gq * xj0 * fc, t1 = s1 * xj1 * fc, t2 = s2 * xj2 * fc;
        gb0 += t0; gb1 += t1; gb2 += t2;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          g0[r] = fmaf(t0, pr[r], g0[r]);
          g1[r] = fmaf(t1, pr[r], g1[r]);
          g2[r] = fmaf(t2, pr[r], g2[r]);
        }
      }
    }
    float* dxo = dxc + (size_t)j * 3 * F;
    dxo[f] = dx0; dxo[F + f] = dx1; dxo[2 * F + f] = dx2;
    float* dmo = dmu_in + (size_t)j * 3 * F;
    const float* __restrict__ gmj = dmu_out + (size_t)j * 3 * F;
    dmo[f] = gmj[f] + dmj0;           // residual mu_out = mu + dmu  plus the edges that read mu[j]
    dmo[F + f] = gmj[F + f] + dmj1;
    dmo[2 * F + f] = gmj[2 * F + f] + dmj2;
  }
  float* pw = partial_w + (size_t)blockIdx.x * 3 * F * R;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    pw[(size_t)f * R + r] = g0[r];
    pw[(size_t)(F + f) * R + r] = g1[r];
    pw[(size_t)(2 * F + f) * R + r] = g2[r];
  }
  float* pb = partial_b + (size_t)blockIdx.x * 3 * F;
  pb[f] = gb0; pb[F + f] = gb1; pb[2 * F + f] = gb2;
}

// ------------------------------------------------------------------- interaction, one block per molecule
// The per-atom kernels above read x[j] and mu[j] (forward) or the upstream gradients of atom i (backward) of every
// edge straight from global memory: 6 (4) coalesced 512-byte rows per edge, 1.75 GB per layer call for 1024 molecules
// in two views - they ran at the speed of those gathers.  An edge never leaves its molecule, so a block that owns a
// whole molecule stages the rows of its atoms in LDS once (3 KB per atom forward, 2 KB backward) and every edge reads
// LDS.  The atoms of the molecule are dealt to the block's thread groups (F threads each, one feature per thread);
// the sums of an atom are formed by one thread in the edge order of the incidence list - bit for bit the per-atom
// kernels' results.
template <int R>
__global__ __launch_bounds__(512) void k_painn_interaction_fwd_mol(
    const float* __restrict__ q, const float* __restrict__ mu, const float* __restrict__ xc,
    const int64_t* __restrict__ idx_j, const int64_t* __restrict__ inc_ptr, const int32_t* __restrict__ inc_idx,
    const float* __restrict__ phi, const float* __restrict__ fcut, const float* __restrict__ dir,
    const float* __restrict__ Wf, const float* __restrict__ bf, const int32_t* __restrict__ mol_ptr, int B, int max_n,
    int F, float* __restrict__ q_out, float* __restrict__ mu_out) {
  extern __shared__ __attribute__((aligned(16))) float sm_rows[];
  const int G = blockDim.x / F, g = threadIdx.x / F, f = threadIdx.x - g * F;  // F is a multiple of 64: g is wave-uniform
  float w0[R], w1[R], w2[R];  // this thread's rows of the filter network: once per (persistent) block
#pragma unroll
  for (int r = 0; r < R; ++r) {
    w0[r] = Wf[(size_t)f * R + r];
    w1[r] = Wf[(size_t)(F + f) * R + r];
    w2[r] = Wf[(size_t)(2 * F + f) * R + r];
  }
  const float b0 = bf[f], b1 = bf[F + f], b2 = bf[2 * F + f];
  float* estage = sm_rows + (size_t)max_n * 6 * F + (threadIdx.x >> 6) * (ECHUNK * EdgeStage<R>::EROW);  // this wave's
  for (int m = blockIdx.x; m < B; m += gridDim.x) {
  const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0;
  if (n > max_n) continue;  // (more atoms than LDS rows: the caller covers the molecule with the per-atom kernel)
  float* xs = sm_rows;                     // [n][3F]
  float* ms = sm_rows + (size_t)n * 3 * F;  // [n][3F]
  __syncthreads();  // the previous molecule's rows are no longer read
  {
    const f32x4* xg = reinterpret_cast<const f32x4*>(xc + (size_t)a0 * 3 * F);
    const f32x4* mg = reinterpret_cast<const f32x4*>(mu + (size_t)a0 * 3 * F);
    const int cnt = n * 3 * F / 4;
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
      reinterpret_cast<f32x4*>(xs)[t] = xg[t];
      reinterpret_cast<f32x4*>(ms)[t] = mg[t];
    }
  }
  __syncthreads();
  for (int ia = g; ia < n; ia += G) {
    const int i = a0 + ia;
    float dq = 0.0f, dm0 = 0.0f, dm1 = 0.0f, dm2 = 0.0f;
    const int64_t p0 = inc_ptr[i], p1 = inc_ptr[i + 1];
    const int lane_ = threadIdx.x & 63;
    for (int64_t pc = p0; pc < p1; pc += ECHUNK) {
     const int cnt = (int)min((int64_t)ECHUNK, p1 - pc);
     EdgeStage<R>::fill(estage, lane_, cnt, pc, inc_idx, idx_j, phi, fcut, dir);  // (same wave writes and reads: in order)
     for (int kk = 0; kk < cnt; ++kk) {
      const float* row = estage + kk * EdgeStage<R>::EROW;             // broadcast reads
      const int jl = __float_as_int(row[R + 4]) - a0;
      float W0 = b0, W1 = b1, W2 = b2;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float pr = row[r];
        W0 = fmaf(pr, w0[r], W0);
        W1 = fmaf(pr, w1[r], W1);
        W2 = fmaf(pr, w2[r], W2);
      }
      const float fc = row[R];
      W0 *= fc; W1 *= fc; W2 *= fc;                                  // painn.py:241
      const float* xj = xs + (size_t)jl * 3 * F;
      const float x0 = W0 * xj[f], x1 = W1 * xj[F + f], x2 = W2 * xj[2 * F + f];  // :56
      const float* mj = ms + (size_t)jl * 3 * F;
      dq += x0;                                                      // :59
      dm0 += x1 * row[R + 1] + x2 * mj[f];                           // :60-61
      dm1 += x1 * row[R + 2] + x2 * mj[F + f];
      dm2 += x1 * row[R + 3] + x2 * mj[2 * F + f];
     }
    }
    q_out[(size_t)i * F + f] = q[(size_t)i * F + f] + dq;            // :63
    float* mo = mu_out + (size_t)i * 3 * F;
    const float* mi = ms + (size_t)ia * 3 * F;
    mo[f] = mi[f] + dm0;                                             // :64
    mo[F + f] = mi[F + f] + dm1;
    mo[2 * F + f] = mi[2 * F + f] + dm2;
  }
  }
}

// First index p in [0, n) with src[p] >= key (src non-decreasing), found by all threads of the block together
// (blockDim.x-ary: the probes of a round are in flight together; two or three rounds instead of log2(n) dependent
// loads).  `scratch`: 16 ints of the block's (dynamic) LDS - __syncthreads_count would add static LDS to a kernel that
// asks for all 160 KB.
template <typename T>
__device__ __forceinline__ int block_lower_bound(const T* __restrict__ src, int n, int64_t key, int* scratch) {
  int lo = 0, hi = n;
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  while (lo < hi) {
    const int len = hi - lo, step = (len + (int)blockDim.x - 1) / (int)blockDim.x;
    const int p = lo + (int)threadIdx.x * step;
    const bool below = p < hi && (int64_t)src[p] < key;
    const int cw = __popcll(__ballot(below));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wave] = cw;
    __syncthreads();
    int c = 0;
    for (int w = 0; w < nw; ++w) c += scratch[w];
    if (c == 0) { hi = lo; break; }
    if (step == 1) { lo += c; break; }
    const int nlo = lo + (c - 1) * step + 1, nhi = min(hi, lo + c * step);
    lo = nlo;
    hi = nhi;
  }
  __syncthreads();
  return lo;
}

// backward: persistent blocks over molecules; the upstream gradients of the molecule's atoms staged in LDS; the
// filter-network gradient partials of the block's thread groups are summed through LDS: one partial per block
// MZ: the interaction's input mu is identically zero and nobody asks for its gradient (the FIRST interaction,
// painn.py:249: mu = zeros): no row of mu is read, the dmumu third of the filter (W2) is not evaluated - its products
// s2 = <dmu_out, mu_j> are exact zeros, so dxc's third channel, its filter-weight gradient and bias gradient are written /
// left as zeros like the general form computes them - and dmu_in is not written.  45 of the ~145 operations per edge and
// feature, the 3F-float row of mu per atom and the 3F-float row of dmu_in per atom go.
template <int R, bool MZ = false>
__global__ __launch_bounds__(512) void k_painn_interaction_bwd_mol(
    const float* __restrict__ dq_out, const float* __restrict__ dmu_out, const float* __restrict__ mu,
    const float* __restrict__ xc, const int64_t* __restrict__ idx_i, const int64_t* __restrict__ inc_ptr,
    const int32_t* __restrict__ inc_idx, const float* __restrict__ phi, const float* __restrict__ fcut,
    const float* __restrict__ dir, const float* __restrict__ Wf, const float* __restrict__ bf,
    const int32_t* __restrict__ mol_ptr, int B, int max_n, int F, float* __restrict__ dxc, float* __restrict__ dmu_in,
    float* __restrict__ partial_w, float* __restrict__ partial_b, int N, int balance) {
  extern __shared__ __attribute__((aligned(16))) float sm_rows[];
  const int G = blockDim.x / F, g = threadIdx.x / F, f = threadIdx.x - g * F;
  // balance: block b owns the atoms whose incidence lists lie in the b-th of gridDim.x equal shares of the edge array
  // (a contiguous atom range [at_lo, at_hi), found in the list offsets) and walks the molecules that range touches -
  // a molecule at a boundary is staged by both neighbours, each taking its own atoms.  Ragged batches: a block per
  // molecule in turn leaves the launch waiting for whoever drew the largest molecules (set C, 2 x 128 molecules: the
  // 72-atom molecule's block runs four times the average).  The assignment is a function of the lists alone, so the
  // block partials and their fixed-order sum stay reproducible.
  int at_lo = 0, at_hi = 0x7fffffff, m_first = blockIdx.x, m_step = gridDim.x;
  if (balance) {
    int* scratch = reinterpret_cast<int*>(sm_rows);
    const int64_t E = inc_ptr[N];
    const int64_t e_lo = E * (int64_t)blockIdx.x / (int64_t)gridDim.x, e_hi = E * ((int64_t)blockIdx.x + 1) / (int64_t)gridDim.x;
    at_lo = blockIdx.x == 0 ? 0 : block_lower_bound(inc_ptr, N + 1, e_lo, scratch);
    at_hi = blockIdx.x == gridDim.x - 1 ? N : block_lower_bound(inc_ptr, N + 1, e_hi, scratch);
    at_lo = min(at_lo, N);
    at_hi = min(at_hi, N);
    // the molecule that holds atom at_lo: the last m with mol_ptr[m] <= at_lo
    m_first = max(0, block_lower_bound(mol_ptr, B + 1, (int64_t)at_lo + 1, scratch) - 1);
    m_step = 1;
  }
  float w0[R], w1[R], w2[R], g0[R], g1[R], g2[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    w0[r] = Wf[(size_t)f * R + r];
    w1[r] = Wf[(size_t)(F + f) * R + r];
    w2[r] = Wf[(size_t)(2 * F + f) * R + r];
    g0[r] = g1[r] = g2[r] = 0.0f;
  }
  const float b0 = bf[f], b1 = bf[F + f], b2 = bf[2 * F + f];
  float gb0 = 0.0f, gb1 = 0.0f, gb2 = 0.0f;
  for (int m = m_first; m < B; m += m_step) {
    const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0;
    if (a0 >= at_hi) break;
    if (n > max_n) continue;  // (more atoms than LDS rows: the caller covers the molecule with the per-atom kernel)
    const int ja_lo = max(at_lo - a0, 0), ja_hi = min(at_hi - a0, n);   // this block's atoms of the molecule
    if (ja_lo >= ja_hi) continue;
    float* gqs = sm_rows;                  // [n][F]   dq_out rows
    float* gms = sm_rows + (size_t)n * F;  // [n][3F]  dmu_out rows
    float* estage = sm_rows + (size_t)max_n * 4 * F + (threadIdx.x >> 6) * (ECHUNK * EdgeStage<R>::EROW);  // this wave's
    __syncthreads();  // the previous molecule's rows are no longer read
    {
      const f32x4* a = reinterpret_cast<const f32x4*>(dq_out + (size_t)a0 * F);
      const f32x4* b4 = reinterpret_cast<const f32x4*>(dmu_out + (size_t)a0 * 3 * F);
      for (int t = threadIdx.x; t < n * F / 4; t += blockDim.x) reinterpret_cast<f32x4*>(gqs)[t] = a[t];
      for (int t = threadIdx.x; t < n * 3 * F / 4; t += blockDim.x) reinterpret_cast<f32x4*>(gms)[t] = b4[t];
    }
    __syncthreads();
    for (int ja = ja_lo + g; ja < ja_hi; ja += G) {
      const int j = a0 + ja;
      const float* __restrict__ xj = xc + (size_t)j * 3 * F;
      const float xj0 = xj[f], xj1 = xj[F + f], xj2 = xj[2 * F + f];
      float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f;
      if constexpr (!MZ) {
        const float* __restrict__ mj = mu + (size_t)j * 3 * F;
        m0 = mj[f]; m1 = mj[F + f]; m2 = mj[2 * F + f];
      }
      float dx0 = 0.0f, dx1 = 0.0f, dx2 = 0.0f, dmj0 = 0.0f, dmj1 = 0.0f, dmj2 = 0.0f;
      const int64_t p0 = inc_ptr[j], p1 = inc_ptr[j + 1];
      const int lane_ = threadIdx.x & 63;
      for (int64_t pc = p0; pc < p1; pc += ECHUNK) {
       const int cnt = (int)min((int64_t)ECHUNK, p1 - pc);
       EdgeStage<R>::fill(estage, lane_, cnt, pc, inc_idx, idx_i, phi, fcut, dir);
       for (int kk = 0; kk < cnt; ++kk) {
        const float* row = estage + kk * EdgeStage<R>::EROW;
        const int il = __float_as_int(row[R + 4]) - a0;
        float pr[R];
        float W0 = b0, W1 = b1, W2 = b2;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          pr[r] = row[r];
          W0 = fmaf(pr[r], w0[r], W0);
          W1 = fmaf(pr[r], w1[r], W1);
          if constexpr (!MZ) W2 = fmaf(pr[r], w2[r], W2);
        }
        const float fc = row[R];
        W0 *= fc; W1 *= fc;
        const float gq = gqs[(size_t)il * F + f];
        const float* gm = gms + (size_t)il * 3 * F;
        const float gm0 = gm[f], gm1 = gm[F + f], gm2 = gm[2 * F + f];
        // (every rounding spelled out: the two instantiations share these lines and must share their results bit for
        // bit - left to -ffp-contract=fast the compiler fused them differently in the two, 1e-7 of a gradient)
        const float s1 = fmaf(gm2, row[R + 3], fmaf(gm1, row[R + 2], mul_rn(gm0, row[R + 1])));
        dx0 = fmaf(gq, W0, dx0);
        dx1 = fmaf(s1, W1, dx1);
        // W_c = (b_c + sum_r phi_r w_c[r]) * fcut
        const float t0 = mul_rn(mul_rn(gq, xj0), fc), t1 = mul_rn(mul_rn(s1, xj1), fc);
        gb0 = add_rn(gb0, t0);
        gb1 = add_rn(gb1, t1);
        if constexpr (!MZ) {
          W2 *= fc;
          const float s2 = fmaf(gm2, m2, fmaf(gm1, m1, mul_rn(gm0, m0)));
          dx2 = fmaf(s2, W2, dx2);
          const float x2 = W2 * xj2;
          dmj0 = fmaf(gm0, x2, dmj0);
          dmj1 = fmaf(gm1, x2, dmj1);
          dmj2 = fmaf(gm2, x2, dmj2);
          const float t2 = mul_rn(mul_rn(s2, xj2), fc);
          gb2 = add_rn(gb2, t2);
#pragma unroll
          for (int r = 0; r < R; ++r) g2[r] = fmaf(t2, pr[r], g2[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          g0[r] = fmaf(t0, pr[r], g0[r]);
          g1[r] = fmaf(t1, pr[r], g1[r]);
        }
       }
      }
      float* dxo = dxc + (size_t)j * 3 * F;
      dxo[f] = dx0; dxo[F + f] = dx1; dxo[2 * F + f] = dx2;
      if constexpr (!MZ) {
        float* dmo = dmu_in + (size_t)j * 3 * F;
        const float* gmj = gms + (size_t)ja * 3 * F;
        dmo[f] = gmj[f] + dmj0;           // residual mu_out = mu + dmu  plus the edges that read mu[j]
        dmo[F + f] = gmj[F + f] + dmj1;
        dmo[2 * F + f] = gmj[2 * F + f] + dmj2;
      }
    }
  }
  // one partial per block: the groups' sums added in group order through LDS (3F (R + 1) floats <= the row stage)
  __syncthreads();
  float* red = sm_rows;  // [3F][R + 1]
  for (int gg = 0; gg < G; ++gg) {
    if (g == gg) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float* p0 = red + (size_t)f * (R + 1) + r;
        float* p1 = red + (size_t)(F + f) * (R + 1) + r;
        float* p2 = red + (size_t)(2 * F + f) * (R + 1) + r;
        if (gg == 0) { *p0 = g0[r]; *p1 = g1[r]; *p2 = g2[r]; }
        else { *p0 += g0[r]; *p1 += g1[r]; *p2 += g2[r]; }
      }
      float* q0 = red + (size_t)f * (R + 1) + R;
      float* q1 = red + (size_t)(F + f) * (R + 1) + R;
      float* q2 = red + (size_t)(2 * F + f) * (R + 1) + R;
      if (gg == 0) { *q0 = gb0; *q1 = gb1; *q2 = gb2; }
      else { *q0 += gb0; *q1 += gb1; *q2 += gb2; }
    }
    __syncthreads();
  }
  if (g == 0) {
    float* pw = partial_w + (size_t)blockIdx.x * 3 * F * R;
    float* pb = partial_b + (size_t)blockIdx.x * 3 * F;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int r = 0; r < R; ++r) pw[(size_t)(c * F + f) * R + r] = red[(size_t)(c * F + f) * (R + 1) + r];
      pb[c * F + f] = red[(size_t)(c * F + f) * (R + 1) + R];
    }
  }
}

// ------------------------------------------------------------------------------------------------ mixing
// mm = mu_channel_mix(mu) [N][3][2F] -> ctx = [q, |mu_V|] [N][2F], dot = sum_xyz mu_V*mu_W  (painn.py:100-104,110)
__global__ void k_painn_mix_pre_fwd(const float* __restrict__ q, const float* __restrict__ mm, int64_t N, int F,
                                    float eps, float* __restrict__ ctx, float* __restrict__ dot, const int32_t* __restrict__ dyn_N) {
  N = dyn_count((int)N, dyn_N);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N * F; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = t / F;
    const int f = (int)(t - a * F);
    const float* m = mm + a * 6 * F;
    const float v0 = m[f], v1 = m[2 * F + f], v2 = m[4 * F + f];
    const float w0 = m[F + f], w1 = m[3 * F + f], w2 = m[5 * F + f];
    ctx[a * 2 * F + f] = q[t];
    ctx[a * 2 * F + F + f] = sqrtf(v0 * v0 + v1 * v1 + v2 * v2 + eps);
    dot[t] = v0 * w0 + v1 * w1 + v2 * w2;
  }
}
// q' = q + dq_intra + dqmu_intra*dot ; mu' = mu + dmu_intra*mu_W   (painn.py:107-113)
__global__ void k_painn_mix_post_fwd(const float* __restrict__ q, const float* __restrict__ mu,
                                     const float* __restrict__ mm, const float* __restrict__ xx,
                                     const float* __restrict__ dot, int64_t N, int F, float* __restrict__ q_out,
                                     float* __restrict__ mu_out, const int32_t* __restrict__ dyn_N) {
  N = dyn_count((int)N, dyn_N);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N * F; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = t / F;
    const int f = (int)(t - a * F);
    const float* x = xx + a * 3 * F;
    const float* m = mm + a * 6 * F;
    q_out[t] = q[t] + x[f] + x[2 * F + f] * dot[t];
    if (mu_out == nullptr) continue;  // (the last block: mu' is not an input of anything, painn.py:262-269)
    const float dmi = x[F + f];
#pragma unroll
    for (int k = 0; k < 3; ++k) mu_out[a * 3 * F + k * F + f] = mu[a * 3 * F + k * F + f] + dmi * m[2 * k * F + F + f];
  }
}
// backward of mix_post: dxx [N][3F], dmm [N][3][2F] (without the |mu_V| term, added by mix_pre_bwd)
__global__ void k_painn_mix_post_bwd(const float* __restrict__ dq_new, const float* __restrict__ dmu_new,
                                     const float* __restrict__ mm, const float* __restrict__ xx,
                                     const float* __restrict__ dot, int64_t N, int F, float* __restrict__ dxx,
                                     float* __restrict__ dmm, const int32_t* __restrict__ dyn_N) {
  N = dyn_count((int)N, dyn_N);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N * F; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = t / F;
    const int f = (int)(t - a * F);
    const float* x = xx + a * 3 * F;
    const float* m = mm + a * 6 * F;
    const float gq = dq_new[t];
    const float ddot = gq * x[2 * F + f];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float gm = dmu_new != nullptr ? dmu_new[a * 3 * F + k * F + f] : 0.0f;   // (NULL: d mu' = 0, the last block)
      const float v = m[2 * k * F + f], w = m[2 * k * F + F + f];
      s += gm * w;
      dmm[a * 6 * F + 2 * k * F + f] = ddot * w;                    // d mu_V (dot term)
      dmm[a * 6 * F + 2 * k * F + F + f] = gm * x[F + f] + ddot * v;  // d mu_W
    }
    dxx[a * 3 * F + f] = gq;
    dxx[a * 3 * F + F + f] = s;
    dxx[a * 3 * F + 2 * F + f] = gq * dot[t];
  }
}
// backward of mix_pre: dq_in = dq_new + dctx[:, :F]; dmm_V += dctx[:, F:] * mu_V / |mu_V|
__global__ void k_painn_mix_pre_bwd(const float* __restrict__ dq_new, const float* __restrict__ dctx,
                                    const float* __restrict__ ctx, const float* __restrict__ mm, int64_t N, int F,
                                    float* __restrict__ dq_in, float* __restrict__ dmm, const int32_t* __restrict__ dyn_N) {
  N = dyn_count((int)N, dyn_N);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N * F; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = t / F;
    const int f = (int)(t - a * F);
    dq_in[t] = dq_new[t] + dctx[a * 2 * F + f];
    const float gvn = dctx[a * 2 * F + F + f] / ctx[a * 2 * F + F + f];
#pragma unroll
    for (int k = 0; k < 3; ++k) dmm[a * 6 * F + 2 * k * F + f] += gvn * mm[a * 6 * F + 2 * k * F + f];
  }
}

__global__ void k_add(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a[i] + b[i];
}

}  // namespace

extern "C" int geossl_painn_edge_geom_dyn(const float* pos, const int64_t* idx_i, const int64_t* idx_j, int64_t E,
                                          float cutoff, const float* offsets, const float* widths, int R, float* dir,
                                          float* fcut, float* phi, const int32_t* dyn_E, hipStream_t stream) {
  if (E <= 0) return 0;
  if (R > RMAX || (dyn_E != nullptr && E >= ((int64_t)1 << 31))) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_painn_edge_geom, dim3(grid1d(E, 256)), dim3(256), 0, stream, pos, idx_i, idx_j, E, cutoff, offsets,
                     widths, R, dir, fcut, phi, dyn_E);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
extern "C" int geossl_painn_edge_geom(const float* pos, const int64_t* idx_i, const int64_t* idx_j, int64_t E,
                                      float cutoff, const float* offsets, const float* widths, int R, float* dir,
                                      float* fcut, float* phi, hipStream_t stream) {
  return geossl_painn_edge_geom_dyn(pos, idx_i, idx_j, E, cutoff, offsets, widths, R, dir, fcut, phi, nullptr, stream);
}

extern "C" int geossl_silu_fwd(const float* u, int64_t n, float* y, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_silu_fwd, dim3(grid1d(n, 256)), dim3(256), 0, stream, u, n, y);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
extern "C" int geossl_silu_bwd(const float* u, const float* dy, int64_t n, float* du, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_silu_bwd, dim3(grid1d(n, 256)), dim3(256), 0, stream, u, dy, n, du);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

#define GEOSSL_PAINN_LIST_BLOCKS 256  // blocks of a launch over an atom LIST (striding over it)
#define GEOSSL_PAINN_DISPATCH_R(KERNEL, ...)                                   \
  do {                                                                         \
    if (R == 20) hipLaunchKernelGGL((KERNEL<20>), __VA_ARGS__);                \
    else if (R == 16) hipLaunchKernelGGL((KERNEL<16>), __VA_ARGS__);           \
    else if (R == 8) hipLaunchKernelGGL((KERNEL<8>), __VA_ARGS__);             \
    else if (R == 32) hipLaunchKernelGGL((KERNEL<32>), __VA_ARGS__);           \
    else return (int)hipErrorInvalidValue;                                     \
  } while (0)

extern "C" int geossl_painn_interaction_fwd(const float* q, const float* mu, const float* xc, const int64_t* idx_j,
                                            const int64_t* inc_ptr, const int32_t* inc_idx, const float* phi,
                                            const float* fcut, const float* dir, const float* Wf, const float* bf,
                                            int64_t N, int F, int R, float* q_out, float* mu_out, hipStream_t stream) {
  if (N <= 0) return 0;
  if (F > 128) return (int)hipErrorInvalidValue;
  GEOSSL_PAINN_DISPATCH_R(k_painn_interaction_fwd, dim3((unsigned)N), dim3(F > 64 ? 128 : 64), 0, stream, q, mu, xc,
                          idx_j, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, (int)N, F, q_out, mu_out, nullptr, nullptr);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

// the same over a LIST of atoms (the atoms of molecules a molecule-staged launch skipped): nlist entries, of which
// min(nlist, *dyn_nlist) are real when dyn_nlist is given
extern "C" int geossl_painn_interaction_fwd_atoms(const float* q, const float* mu, const float* xc, const int64_t* idx_j,
                                                  const int64_t* inc_ptr, const int32_t* inc_idx, const float* phi,
                                                  const float* fcut, const float* dir, const float* Wf, const float* bf,
                                                  const int32_t* atom_list, int64_t nlist, const int32_t* dyn_nlist, int F,
                                                  int R, float* q_out, float* mu_out, hipStream_t stream) {
  if (nlist <= 0) return 0;
  if (F > 128 || atom_list == nullptr) return (int)hipErrorInvalidValue;
  const unsigned nb = (unsigned)(nlist < 2 * GEOSSL_PAINN_LIST_BLOCKS ? nlist : 2 * GEOSSL_PAINN_LIST_BLOCKS);
  GEOSSL_PAINN_DISPATCH_R(k_painn_interaction_fwd, dim3(nb), dim3(F > 64 ? 128 : 64), 0, stream, q, mu, xc, idx_j, inc_ptr,
                          inc_idx, phi, fcut, dir, Wf, bf, (int)nlist, F, q_out, mu_out, atom_list, dyn_nlist);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

#define GEOSSL_PAINN_BWD_BLOCKS 1536  // six per CU (the register budget of the kernel); one filter-gradient partial per block
extern "C" int64_t geossl_painn_interaction_bwd_workspace_floats(int64_t N, int F, int R) {
  const int64_t nb = N < GEOSSL_PAINN_BWD_BLOCKS ? N : GEOSSL_PAINN_BWD_BLOCKS;
  return nb * (3 * (int64_t)F * R + 3 * F);
}

static int painn_interaction_bwd_launch(const float* dq_out, const float* dmu_out, const float* mu, const float* xc,
                                        const int64_t* idx_i, const int64_t* inc_ptr, const int32_t* inc_idx,
                                        const float* phi, const float* fcut, const float* dir, const float* Wf,
                                        const float* bf, int64_t N, int F, int R, float* dxc, float* dmu_in,
                                        float* dWf, float* dbf, float* workspace, int accumulate,
                                        const int32_t* atom_list, const int32_t* dyn_nlist, hipStream_t stream) {
  if (N <= 0) return 0;
  if (F > 128) return (int)hipErrorInvalidValue;
  // (a list of atoms - the few oversized molecules of a batch - on few blocks: every block loads its filter rows and
  // leaves a 31 KB partial whether it finds an atom or not; 1536 such blocks cost more than the molecules they cover)
  const int cap_blocks = atom_list != nullptr ? GEOSSL_PAINN_LIST_BLOCKS : GEOSSL_PAINN_BWD_BLOCKS;
  const int nb = (int)(N < cap_blocks ? N : cap_blocks);
  float* pw = workspace;
  float* pb = workspace + (size_t)nb * 3 * F * R;
  GEOSSL_PAINN_DISPATCH_R(k_painn_interaction_bwd, dim3(nb), dim3(F > 64 ? 128 : 64), 0, stream, dq_out, dmu_out, mu, xc,
                          idx_i, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, (int)N, F, dxc, dmu_in, pw, pb, atom_list,
                          dyn_nlist);
  GEOSSL_CHECK_LAUNCH();
  // fixed-order two-stage sums of the per-block partials (64 outputs x 4 slices of the block list per reduction block)
  ReduceMulti rm;  // both fixed-order sums over the block partials in one launch (k_reduce_partials' arithmetic)
  float* ow[1] = {dWf};
  float* ob[1] = {dbf};
  rm.add(pw, 3 * F * R, 3 * F * R, 3 * F * R, 1, ow, 1);
  rm.add(pb, 3 * F, 3 * F, 3 * F, 1, ob, 1);
  hipLaunchKernelGGL(geossl::k_reduce_multi, dim3(rm.blocks(), 1), dim3(256), 0, stream, rm, nb, accumulate);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
extern "C" int geossl_painn_interaction_bwd(const float* dq_out, const float* dmu_out, const float* mu, const float* xc,
                                            const int64_t* idx_i, const int64_t* inc_ptr, const int32_t* inc_idx,
                                            const float* phi, const float* fcut, const float* dir, const float* Wf,
                                            const float* bf, int64_t N, int F, int R, float* dxc, float* dmu_in,
                                            float* dWf, float* dbf, float* workspace, int accumulate,
                                            hipStream_t stream) {
  return painn_interaction_bwd_launch(dq_out, dmu_out, mu, xc, idx_i, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, N, F, R, dxc,
                                      dmu_in, dWf, dbf, workspace, accumulate, nullptr, nullptr, stream);
}
// the same over a LIST of atoms (see geossl_painn_interaction_fwd_atoms); workspace: geossl_painn_interaction_bwd_workspace_floats(nlist, F, R)
extern "C" int geossl_painn_interaction_bwd_atoms(const float* dq_out, const float* dmu_out, const float* mu,
                                                  const float* xc, const int64_t* idx_i, const int64_t* inc_ptr,
                                                  const int32_t* inc_idx, const float* phi, const float* fcut,
                                                  const float* dir, const float* Wf, const float* bf,
                                                  const int32_t* atom_list, int64_t nlist, const int32_t* dyn_nlist, int F,
                                                  int R, float* dxc, float* dmu_in, float* dWf, float* dbf,
                                                  float* workspace, int accumulate, hipStream_t stream) {
  if (atom_list == nullptr) return (int)hipErrorInvalidValue;
  return painn_interaction_bwd_launch(dq_out, dmu_out, mu, xc, idx_i, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, nlist, F, R,
                                      dxc, dmu_in, dWf, dbf, workspace, accumulate, atom_list, dyn_nlist, stream);
}

// ---- the same two entry points with the molecule layout (mol_ptr [B+1] int32, max_n): one block per molecule.
// F must be a multiple of 64 and the molecule's rows must fit the LDS; otherwise the per-atom kernels are used.
static inline bool painn_mol_ok(int F, int max_n, size_t floats_per_atom) {
  static const bool per_atom = getenv("GEOSSL_PAINN_PER_ATOM") != nullptr;  // the per-atom kernels, for A/B runs
  return !per_atom && (F == 64 || F == 128) && max_n >= 1 && (size_t)max_n * floats_per_atom * sizeof(float) <= 150 * 1024;
}
#define GEOSSL_PAINN_BWD_MOL_BLOCKS 256  // one 512-thread block per CU (its register budget); one filter-gradient partial per block
extern "C" int geossl_painn_interaction_fwd_mol(const float* q, const float* mu, const float* xc, const int64_t* idx_j,
                                                const int64_t* inc_ptr, const int32_t* inc_idx, const float* phi,
                                                const float* fcut, const float* dir, const float* Wf, const float* bf,
                                                const int32_t* mol_ptr, int64_t B, int max_n, int64_t N, int F, int R,
                                                float* q_out, float* mu_out, hipStream_t stream) {
  if (N <= 0 || B <= 0) return 0;
  if (!painn_mol_ok(F, max_n, (size_t)6 * F))
    return geossl_painn_interaction_fwd(q, mu, xc, idx_j, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, N, F, R, q_out, mu_out,
                                        stream);
  const size_t estage = (size_t)(4 * F / 64) * ECHUNK * (((R + 5 + 3) / 4) * 4);  // per-wave edge stages (floats)
  const size_t lds = ((size_t)max_n * 6 * F + estage) * sizeof(float);
  if (lds > 160 * 1024)  // (rows + edge stages above the LDS: the per-atom form)
    return geossl_painn_interaction_fwd(q, mu, xc, idx_j, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, N, F, R, q_out, mu_out,
                                        stream);
  // persistent blocks (the filter rows of a thread are fetched once per block): as many as fit the chip
  const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
  const int nb = (int)(B < 256 * per_cu ? B : 256 * per_cu);
#define LAUNCH_FWD_MOL(RV)                                                                                         \
  do {                                                                                                             \
    allow_big_lds(&k_painn_interaction_fwd_mol<RV>);                                                               \
    hipLaunchKernelGGL((k_painn_interaction_fwd_mol<RV>), dim3((unsigned)nb), dim3(4 * F), lds, stream, q, mu, xc, \
                       idx_j, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, mol_ptr, (int)B, max_n, F, q_out, mu_out); \
  } while (0)
  if (R == 20) LAUNCH_FWD_MOL(20); else if (R == 16) LAUNCH_FWD_MOL(16); else if (R == 8) LAUNCH_FWD_MOL(8);
  else if (R == 32) LAUNCH_FWD_MOL(32); else return (int)hipErrorInvalidValue;
#undef LAUNCH_FWD_MOL
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
extern "C" int64_t geossl_painn_interaction_bwd_mol_workspace_floats(int64_t N, int64_t B, int F, int R) {
  const int64_t a = geossl_painn_interaction_bwd_workspace_floats(N, F, R);  // (the per-atom form may be chosen)
  const int64_t nb = B < GEOSSL_PAINN_BWD_MOL_BLOCKS ? B : GEOSSL_PAINN_BWD_MOL_BLOCKS;
  const int64_t b = nb * (3 * (int64_t)F * R + 3 * F);
  return a > b ? a : b;
}
// Largest molecule (atoms) whose rows fit the LDS of a molecule-staged interaction launch: kind 0 = matrix-pipe forward
// (painn_mma.hip), 1 = vector forward, 2 = backward; 0 when the shape has no such form.  A launch whose layout holds
// larger molecules SKIPS them (geossl_painn_interaction_fwd_mma_dyn, geossl_painn_interaction_bwd_mol_skip): the caller
// covers their atoms with geossl_painn_interaction_fwd_atoms / _bwd_atoms - one oversized molecule no longer sends the
// whole batch to the per-atom kernels.
extern "C" int geossl_painn_stage_cap(int kind, int F, int R) {
  if (F != 64 && F != 128) return 0;
  if (kind != 0 && getenv("GEOSSL_PAINN_PER_ATOM") != nullptr) return 0;  // (A/B runs: the vector kernels per atom only)
  if (kind == 0) return (F == 128 && (R == 8 || R == 16 || R == 20)) ? 44 : 0;   // painn_mma_lds(44) = 157.6 KB
  // vector kernels: the molecule's rows (6 F floats per atom forward, 4 F backward) + the per-wave edge stages in 160 KB,
  // and never more than painn_mol_ok admits
  const size_t estage = (size_t)(4 * F / 64) * ECHUNK * (((R + 5 + 3) / 4) * 4) * sizeof(float);
  const size_t per_atom = (size_t)(kind == 1 ? 6 : 4) * F * sizeof(float);
  if (kind == 1 || (kind == 2 && R != 32)) {
    const size_t a = (160 * 1024 - estage) / per_atom, b = (150 * 1024) / per_atom;
    return (int)(a < b ? a : b);
  }
  return 0;
}

static int painn_interaction_bwd_mol_launch(const float* dq_out, const float* dmu_out, const float* mu,
                                            const float* xc, const int64_t* idx_i, const int64_t* inc_ptr,
                                            const int32_t* inc_idx, const float* phi, const float* fcut,
                                            const float* dir, const float* Wf, const float* bf,
                                            const int32_t* mol_ptr, int64_t B, int max_n, int64_t N, int F, int R,
                                            float* dxc, float* dmu_in, float* dWf, float* dbf, float* workspace,
                                            int accumulate, bool skip_big, hipStream_t stream) {
  if (N <= 0 || B <= 0) return 0;
  // mu == NULL: the input mu is identically zero (the first interaction) and dmu_in is not wanted (NULL as well) - only
  // the molecule-staged kernel knows that form
  const bool mz = mu == nullptr;
  if (mz != (dmu_in == nullptr)) return (int)hipErrorInvalidValue;
  if (skip_big) {  // molecules above the LDS rows are left to the caller (geossl_painn_interaction_bwd_atoms)
    const int cap = geossl_painn_stage_cap(2, F, R);
    if (cap <= 0) return (int)hipErrorInvalidValue;
    max_n = max_n < cap ? max_n : cap;
  }
  size_t stage = (size_t)max_n * 4 * F, red = (size_t)3 * F * (R + 1);
  // (n_rbf = 32: the molecule-staged backward would hold 32 filter rows per thread and spill; the per-atom form runs)
  if (!painn_mol_ok(F, max_n, (size_t)4 * F) || red * sizeof(float) > 150 * 1024 || R == 32)
    return (skip_big || mz) ? (int)hipErrorInvalidValue :
        geossl_painn_interaction_bwd(dq_out, dmu_out, mu, xc, idx_i, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, N, F, R,
                                     dxc, dmu_in, dWf, dbf, workspace, accumulate, stream);
  const int nb = (int)(B < GEOSSL_PAINN_BWD_MOL_BLOCKS ? B : GEOSSL_PAINN_BWD_MOL_BLOCKS);
  // Batches whose largest molecule has more than twice the average number of atoms (Molecule3D with hydrogens: 72
  // against 26; a capacity bucket's bounds stand in for both): equal shares of the edges per block - set C, 2 x 128
  // molecules: 1.94 -> 1.68 ms per step.  Otherwise a block per molecule in turn, as up to round 4: with molecules of
  // similar size there is little to balance and a molecule cut at a block boundary is staged twice (set B, 2 x 128:
  // 1.15 against 1.18 ms; set A: 1.12 against 1.16).  GEOSSL_PAINN_BALANCE=0 / =1 force either form.
  const char* balance_env = getenv("GEOSSL_PAINN_BALANCE");   // (read per call: tests switch it inside one process)
  const int balance = balance_env ? (balance_env[0] == '0' ? 0 : 1) : (B * (int64_t)max_n > 2 * N ? 1 : 0);
  if (N >= ((int64_t)1 << 31)) return (int)hipErrorInvalidValue;
  const size_t estage = (size_t)(4 * F / 64) * ECHUNK * (((R + 5 + 3) / 4) * 4);  // per-wave edge stages (floats)
  const size_t lds = ((stage + estage) > red ? (stage + estage) : red) * sizeof(float);
  if (lds > 160 * 1024)  // (rows + edge stages above the LDS: the per-atom form)
    return (skip_big || mz) ? (int)hipErrorInvalidValue :
        geossl_painn_interaction_bwd(dq_out, dmu_out, mu, xc, idx_i, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, N, F, R,
                                     dxc, dmu_in, dWf, dbf, workspace, accumulate, stream);
  float* pw = workspace;
  float* pb = workspace + (size_t)nb * 3 * F * R;
#define LAUNCH_BWD_MOL(RV)                                                                                          \
  do {                                                                                                              \
    if (mz) {                                                                                                       \
      allow_big_lds(&k_painn_interaction_bwd_mol<RV, true>);                                                        \
      hipLaunchKernelGGL((k_painn_interaction_bwd_mol<RV, true>), dim3(nb), dim3(4 * F), lds, stream, dq_out,       \
                         dmu_out, mu, xc, idx_i, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, mol_ptr, (int)B, max_n,  \
                         F, dxc, dmu_in, pw, pb, (int)N, balance);                                                  \
    } else {                                                                                                        \
      allow_big_lds(&k_painn_interaction_bwd_mol<RV>);                                                              \
      hipLaunchKernelGGL((k_painn_interaction_bwd_mol<RV>), dim3(nb), dim3(4 * F), lds, stream, dq_out, dmu_out,    \
                         mu, xc, idx_i, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, mol_ptr, (int)B, max_n, F, dxc,   \
                         dmu_in, pw, pb, (int)N, balance);                                                          \
    }                                                                                                               \
  } while (0)
  if (R == 20) LAUNCH_BWD_MOL(20); else if (R == 16) LAUNCH_BWD_MOL(16); else if (R == 8) LAUNCH_BWD_MOL(8);
  else return (int)hipErrorInvalidValue;
#undef LAUNCH_BWD_MOL
  GEOSSL_CHECK_LAUNCH();
  ReduceMulti rm;  // both fixed-order sums over the block partials in one launch (k_reduce_partials' arithmetic)
  float* ow[1] = {dWf};
  float* ob[1] = {dbf};
  rm.add(pw, 3 * F * R, 3 * F * R, 3 * F * R, 1, ow, 1);
  rm.add(pb, 3 * F, 3 * F, 3 * F, 1, ob, 1);
  hipLaunchKernelGGL(geossl::k_reduce_multi, dim3(rm.blocks(), 1), dim3(256), 0, stream, rm, nb, accumulate);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
extern "C" int geossl_painn_interaction_bwd_mol(const float* dq_out, const float* dmu_out, const float* mu,
                                                const float* xc, const int64_t* idx_i, const int64_t* inc_ptr,
                                                const int32_t* inc_idx, const float* phi, const float* fcut,
                                                const float* dir, const float* Wf, const float* bf,
                                                const int32_t* mol_ptr, int64_t B, int max_n, int64_t N, int F, int R,
                                                float* dxc, float* dmu_in, float* dWf, float* dbf, float* workspace,
                                                int accumulate, hipStream_t stream) {
  return painn_interaction_bwd_mol_launch(dq_out, dmu_out, mu, xc, idx_i, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, mol_ptr,
                                          B, max_n, N, F, R, dxc, dmu_in, dWf, dbf, workspace, accumulate, false, stream);
}
// molecules of more than geossl_painn_stage_cap(2, F, R) atoms are SKIPPED (their rows of dxc / dmu_in are not written,
// their edges add nothing to dWf / dbf): the caller covers them with geossl_painn_interaction_bwd_atoms(accumulate = 1)
extern "C" int geossl_painn_interaction_bwd_mol_skip(const float* dq_out, const float* dmu_out, const float* mu,
                                                     const float* xc, const int64_t* idx_i, const int64_t* inc_ptr,
                                                     const int32_t* inc_idx, const float* phi, const float* fcut,
                                                     const float* dir, const float* Wf, const float* bf,
                                                     const int32_t* mol_ptr, int64_t B, int max_n, int64_t N, int F, int R,
                                                     float* dxc, float* dmu_in, float* dWf, float* dbf, float* workspace,
                                                     int accumulate, hipStream_t stream) {
  return painn_interaction_bwd_mol_launch(dq_out, dmu_out, mu, xc, idx_i, inc_ptr, inc_idx, phi, fcut, dir, Wf, bf, mol_ptr,
                                          B, max_n, N, F, R, dxc, dmu_in, dWf, dbf, workspace, accumulate, true, stream);
}

extern "C" int geossl_painn_mix_pre_fwd_dyn(const float* q, const float* mm, int64_t N, int F, float eps, float* ctx,
                                            float* dot, const int32_t* dyn_N, hipStream_t stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_painn_mix_pre_fwd, dim3(grid1d(N * F, 256)), dim3(256), 0, stream, q, mm, N, F, eps, ctx, dot,
                     dyn_N);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
extern "C" int geossl_painn_mix_pre_fwd(const float* q, const float* mm, int64_t N, int F, float eps, float* ctx,
                                        float* dot, hipStream_t stream) {
  return geossl_painn_mix_pre_fwd_dyn(q, mm, N, F, eps, ctx, dot, nullptr, stream);
}
extern "C" int geossl_painn_mix_post_fwd_dyn(const float* q, const float* mu, const float* mm, const float* xx,
                                             const float* dot, int64_t N, int F, float* q_out, float* mu_out,
                                             const int32_t* dyn_N, hipStream_t stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_painn_mix_post_fwd, dim3(grid1d(N * F, 256)), dim3(256), 0, stream, q, mu, mm, xx, dot, N, F, q_out,
                     mu_out, dyn_N);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
extern "C" int geossl_painn_mix_post_fwd(const float* q, const float* mu, const float* mm, const float* xx,
                                         const float* dot, int64_t N, int F, float* q_out, float* mu_out,
                                         hipStream_t stream) {
  return geossl_painn_mix_post_fwd_dyn(q, mu, mm, xx, dot, N, F, q_out, mu_out, nullptr, stream);
}
extern "C" int geossl_painn_mix_post_bwd_dyn(const float* dq_new, const float* dmu_new, const float* mm, const float* xx,
                                             const float* dot, int64_t N, int F, float* dxx, float* dmm,
                                             const int32_t* dyn_N, hipStream_t stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_painn_mix_post_bwd, dim3(grid1d(N * F, 256)), dim3(256), 0, stream, dq_new, dmu_new, mm, xx, dot, N,
                     F, dxx, dmm, dyn_N);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
extern "C" int geossl_painn_mix_post_bwd(const float* dq_new, const float* dmu_new, const float* mm, const float* xx,
                                         const float* dot, int64_t N, int F, float* dxx, float* dmm,
                                         hipStream_t stream) {
  return geossl_painn_mix_post_bwd_dyn(dq_new, dmu_new, mm, xx, dot, N, F, dxx, dmm, nullptr, stream);
}
extern "C" int geossl_painn_mix_pre_bwd_dyn(const float* dq_new, const float* dctx, const float* ctx, const float* mm,
                                            int64_t N, int F, float* dq_in, float* dmm, const int32_t* dyn_N,
                                            hipStream_t stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_painn_mix_pre_bwd, dim3(grid1d(N * F, 256)), dim3(256), 0, stream, dq_new, dctx, ctx, mm, N, F,
                     dq_in, dmm, dyn_N);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
extern "C" int geossl_painn_mix_pre_bwd(const float* dq_new, const float* dctx, const float* ctx, const float* mm,
                                        int64_t N, int F, float* dq_in, float* dmm, hipStream_t stream) {
  return geossl_painn_mix_pre_bwd_dyn(dq_new, dctx, ctx, mm, N, F, dq_in, dmm, nullptr, stream);
}
extern "C" int geossl_add(const float* a, const float* b, int64_t n, float* out, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_add, dim3(grid1d(n, 256)), dim3(256), 0, stream, a, b, n, out);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
