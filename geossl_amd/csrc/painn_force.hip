// PaiNN position gradient, first order (examples/finetune_md17.py:46 takes
// grad(energy, positions) through Geom3D/models/painn.py:232-241,54-64).  Positions enter PaiNN through the edge geometry
// only: dir = r_ij / d (painn.py:237), the radial basis phi(d) (painn_utils.py:99-103) and the cosine cutoff fcut(d)
// (:152-154), the last two through the filter W_ij = (phi W^T + b) * fcut (painn.py:239-241).  Three kernels:
//   edge_grads      per interaction block: dL/dphi, dL/dfcut, dL/ddir of every edge (a sum over the F features of the
//                   edge: one wave per edge, fixed-order butterfly sums), accumulated over the blocks in launch order;
//   edge_geom_bwd   the derivative of (dir, phi, fcut) with respect to r_ij applied to those: dL/dr_ij per edge;
//   position_grad   dL/dpos[a] = sum over edges with idx_i = a of dL/dr - sum over edges with idx_j = a of dL/dr,
//                   each in ascending edge order through the batch's incidence lists (no atomics).
#include "common.h"
#include "geossl_hip.h"

using namespace geossl;

namespace {

constexpr int EG_WAVES = 4;

// Wf/bf: the filter_net rows of this block ([3F][R], [3F]); dq_out / dmu_out: the gradient at the block's output
// (target rows idx_i), mu / xc: the block's input vectors and Dense(q) of the source rows idx_j (painn.py:53-60).
template <int R>
__global__ __launch_bounds__(64 * EG_WAVES) void k_painn_edge_grads(
    const float* __restrict__ dq_out, const float* __restrict__ dmu_out, const float* __restrict__ mu,
    const float* __restrict__ xc, const int64_t* __restrict__ idx_i, const int64_t* __restrict__ idx_j,
    const float* __restrict__ phi, const float* __restrict__ fcut, const float* __restrict__ dir,
    const float* __restrict__ Wf, const float* __restrict__ bf, int64_t E, int F, float* __restrict__ dphi,
    float* __restrict__ dfcut, float* __restrict__ ddir, int accumulate) {
  extern __shared__ float smem[];
  constexpr int RP = R + 1;  // row pitch: lanes read consecutive rows, an odd pitch spreads them over the banks
  float* w = smem;            // [3F][RP]
  float* b = smem + 3 * F * RP;
  for (int t = threadIdx.x; t < 3 * F * R; t += blockDim.x) w[(t / R) * RP + t % R] = Wf[t];
  for (int t = threadIdx.x; t < 3 * F; t += blockDim.x) b[t] = bf[t];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t e = (int64_t)blockIdx.x * EG_WAVES + wave; e < E; e += (int64_t)gridDim.x * EG_WAVES) {
    const int64_t i = idx_i[e], j = idx_j[e];
    const float fc = fcut[e];
    const float dx = dir[3 * e], dy = dir[3 * e + 1], dz = dir[3 * e + 2];
    float ph[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ph[r] = phi[e * R + r];
    float aphi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) aphi[r] = 0.0f;
    float afc = 0.0f, adx = 0.0f, ady = 0.0f, adz = 0.0f;
    for (int f = lane; f < F; f += 64) {
      const float x0 = xc[j * 3 * F + f], x1 = xc[j * 3 * F + F + f], x2 = xc[j * 3 * F + 2 * F + f];
      const float gq = dq_out[i * F + f];
      const float gx = dmu_out[(i * 3) * F + f], gy = dmu_out[(i * 3 + 1) * F + f], gz = dmu_out[(i * 3 + 2) * F + f];
      const float mx = mu[(j * 3) * F + f], my = mu[(j * 3 + 1) * F + f], mz = mu[(j * 3 + 2) * F + f];
      // dL/dW of the three channels (painn.py:56-60): dq = W0 x0, dmu = W1 x1 dir + W2 x2 mu_j
      const float a0 = gq * x0, a1 = (gx * dx + gy * dy + gz * dz) * x1, a2 = (gx * mx + gy * my + gz * mz) * x2;
      const float* w0 = w + f * RP;
      const float* w1 = w + (F + f) * RP;
      const float* w2 = w + (2 * F + f) * RP;
      float r0 = b[f], r1 = b[F + f], r2 = b[2 * F + f];  // the filter before the cutoff
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float v0 = w0[r], v1 = w1[r], v2 = w2[r];
        r0 = fmaf(ph[r], v0, r0);
        r1 = fmaf(ph[r], v1, r1);
        r2 = fmaf(ph[r], v2, r2);
        aphi[r] += a0 * v0 + a1 * v1 + a2 * v2;
      }
      afc += a0 * r0 + a1 * r1 + a2 * r2;
      const float t = r1 * fc * x1;  // d dmu / d dir = W1 x1
      adx = fmaf(gx, t, adx);
      ady = fmaf(gy, t, ady);
      adz = fmaf(gz, t, adz);
    }
    float mine = 0.0f;  // lane r keeps entry r of dL/dphi, lanes R .. R+3: dL/dfcut and dL/ddir
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float v = wave_sum(aphi[r]) * fc;
      if (lane == r) mine = v;
    }
    {
      const float v = wave_sum(afc);
      if (lane == R) mine = v;
    }
    {
      const float vx = wave_sum(adx), vy = wave_sum(ady), vz = wave_sum(adz);
      if (lane == R + 1) mine = vx;
      if (lane == R + 2) mine = vy;
      if (lane == R + 3) mine = vz;
    }
    float* dst = lane < R ? dphi + e * R + lane : (lane == R ? dfcut + e : ddir + 3 * e + (lane - R - 1));
    if (lane < R + 4) *dst = accumulate ? *dst + mine : mine;
  }
}

// d(dir, phi, fcut)/d r_ij (painn.py:232-239, painn_utils.py:99-103,152-154) applied to the edge gradients
__global__ void k_painn_edge_geom_bwd(const float* __restrict__ pos, const int64_t* __restrict__ idx_i,
                                      const int64_t* __restrict__ idx_j, int64_t E, float cutoff,
                                      const float* __restrict__ offsets, const float* __restrict__ widths, int R,
                                      const float* __restrict__ dphi, const float* __restrict__ dfcut,
                                      const float* __restrict__ ddir, float* __restrict__ dr) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx_i[e], j = idx_j[e];
    const float rx = pos[3 * i] - pos[3 * j], ry = pos[3 * i + 1] - pos[3 * j + 1], rz = pos[3 * i + 2] - pos[3 * j + 2];
    const float d = sqrtf(norm2_rn(rx, ry, rz));
    const float ux = rx / d, uy = ry / d, uz = rz / d;
    float dd = 0.0f;  // dL/dd through the radial basis and the cutoff
    for (int r = 0; r < R; ++r) {
      const float wd = widths[r];
      const float coeff = -0.5f / (wd * wd);
      const float diff = d - offsets[r];
      dd = fmaf(dphi[e * R + r], expf(coeff * (diff * diff)) * (2.0f * coeff * diff), dd);
    }
    if (d < cutoff) {
      const float k = GEOSSL_PI_F / cutoff;
      dd = fmaf(dfcut[e], -0.5f * k * sinf(d * k), dd);
    }
    // dir = r / d: d dir / d r = (I - dir dir^T) / d
    const float gx = ddir[3 * e], gy = ddir[3 * e + 1], gz = ddir[3 * e + 2];
    const float gu = gx * ux + gy * uy + gz * uz;
    dr[3 * e] = fmaf(dd, ux, (gx - gu * ux) / d);
    dr[3 * e + 1] = fmaf(dd, uy, (gy - gu * uy) / d);
    dr[3 * e + 2] = fmaf(dd, uz, (gz - gu * uz) / d);
  }
}

// r_ij = pos[idx_i] - pos[idx_j] (painn.py:232): + for the edges that list the atom in row 0, - in row 1
__global__ void k_painn_position_grad(const float* __restrict__ dr, const int64_t* __restrict__ ptr_i,
                                      const int32_t* __restrict__ inc_i, const int64_t* __restrict__ ptr_j,
                                      const int32_t* __restrict__ inc_j, int64_t N, float* __restrict__ dpos) {
  for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < N; a += (int64_t)gridDim.x * blockDim.x) {
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    for (int64_t k = ptr_i[a]; k < ptr_i[a + 1]; ++k) {
      const int64_t e = inc_i[k];
      sx += dr[3 * e]; sy += dr[3 * e + 1]; sz += dr[3 * e + 2];
    }
    for (int64_t k = ptr_j[a]; k < ptr_j[a + 1]; ++k) {
      const int64_t e = inc_j[k];
      sx -= dr[3 * e]; sy -= dr[3 * e + 1]; sz -= dr[3 * e + 2];
    }
    dpos[3 * a] = sx; dpos[3 * a + 1] = sy; dpos[3 * a + 2] = sz;
  }
}

inline int grid_for(int64_t n, int per_block, int cap) {
  const int64_t g = (n + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int geossl_painn_edge_grads(const float* dq_out, const float* dmu_out, const float* mu, const float* xc,
                                       const int64_t* idx_i, const int64_t* idx_j, const float* phi, const float* fcut,
                                       const float* dir, const float* Wf, const float* bf, int64_t E, int F, int R,
                                       float* dphi, float* dfcut, float* ddir, int accumulate, hipStream_t stream) {
  if (E <= 0) return 0;
  if (F > 128 || F < 1) return (int)hipErrorInvalidValue;
  const dim3 grid(grid_for(E, EG_WAVES * 8, 4096)), block(64 * EG_WAVES);
  const size_t lds = (size_t)(3 * F * (R + 1) + 3 * F) * sizeof(float);
#define GEOSSL_EG(RR)                                                                                              \
  hipLaunchKernelGGL((k_painn_edge_grads<RR>), grid, block, lds, stream, dq_out, dmu_out, mu, xc, idx_i, idx_j, phi, \
                     fcut, dir, Wf, bf, E, F, dphi, dfcut, ddir, accumulate)
  if (R == 20) GEOSSL_EG(20);
  else if (R == 16) GEOSSL_EG(16);
  else if (R == 8) GEOSSL_EG(8);
  else if (R == 32) GEOSSL_EG(32);
  else return (int)hipErrorInvalidValue;
#undef GEOSSL_EG
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_painn_edge_geom_bwd(const float* pos, const int64_t* idx_i, const int64_t* idx_j, int64_t E,
                                          float cutoff, const float* offsets, const float* widths, int R,
                                          const float* dphi, const float* dfcut, const float* ddir, float* dr,
                                          hipStream_t stream) {
  if (E <= 0) return 0;
  hipLaunchKernelGGL(k_painn_edge_geom_bwd, dim3(grid_for(E, 256, 4096)), dim3(256), 0, stream, pos, idx_i, idx_j, E,
                     cutoff, offsets, widths, R, dphi, dfcut, ddir, dr);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_painn_position_grad(const float* dr, const int64_t* inc_i_ptr, const int32_t* inc_i_idx,
                                          const int64_t* inc_j_ptr, const int32_t* inc_j_idx, int64_t N, float* dpos,
                                          hipStream_t stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_painn_position_grad, dim3(grid_for(N, 256, 4096)), dim3(256), 0, stream, dr, inc_i_ptr, inc_i_idx,
                     inc_j_ptr, inc_j_idx, N, dpos);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
