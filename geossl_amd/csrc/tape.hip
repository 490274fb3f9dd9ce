// Kernels of the second-order route (geossl_amd/tape.py): training on forces (finetune_md17.py:46-54) differentiates the
// position gradient of the backbone again.  That route is a tape of small primitives, each of which has a derivative
// made of the same primitives; the dense ones (row / column GEMMs, neighbour aggregation, pair product) are the kernels
// of the first-order path, the rest is here: element-wise maps with their first and second derivatives as maps of their
// own, broadcast arithmetic over the row / column / xyz-triple layouts the two backbones use, fixed-order reductions
// over the same layouts, row gather and its adjoint over a sorted incidence list, block copies.
//
// Everything is fp32, HBM-bound and bit-reproducible (no atomics: a reduction has one owner per output element and a
// fixed order of summation).  Layout vocabulary: a tensor is a row-major [R][D] matrix; PaiNN's vector features
// [n][3][F] are the matrix [3 n][F] (row 3 a + c), so "a third row" is the [n][F] matrix whose row r / 3 belongs to row r.
#include "common.h"

namespace geossl {
namespace {

inline int tape_grid(int64_t n, int block, int cap = 8192) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---------------------------------------------------------------------------------------------------- element-wise maps
// y = f(alpha x + beta).  A map's derivative is the next map of its family (tape.py: _DERIV), so that two
// differentiations of the stack stay inside this table.
enum Unary {
  U_AFFINE = 0,   // t
  U_EXP = 1,
  U_COS = 2,
  U_SIN = 3,
  U_SSP = 4,      // softplus(t) - log 2 (ShiftedSoftplus, schnet.py:213-216; F.softplus: t above 20 passes through)
  U_SIGMOID = 5,  // ssp'
  U_DSIGMOID = 6, // s (1 - s)
  U_D2SIGMOID = 7,// s (1 - s)(1 - 2 s)
  U_RECIP = 8,
  U_SQRT = 9,
  U_SILU = 10,    // t s(t)
  U_DSILU = 11,   // s (1 + t (1 - s))
  U_D2SILU = 12,  // s (1 - s)(2 + t (1 - 2 s))
  U_GAUSS = 13,   // exp(alpha x^2), beta unused (Gaussian smearing of a centred distance)
  U_DGAUSS = 14,  // 2 alpha x exp(alpha x^2)
  U_D2GAUSS = 15, // (2 alpha + 4 alpha^2 x^2) exp(alpha x^2)
  U_LT = 16,      // x < alpha ? 1 : 0
  U_ABS = 17,     // |t|   (L1 loss on energies / forces, finetune_md17.py:51-54 with --loss l1)
  U_SIGN = 18,    // sign(t), 0 at 0
  U_DRECIP = 19,  // -1 / t^2
  U_D2RECIP = 20, // 2 / t^3
  U_RSQRT = 21,   // t^-1/2   (sqrt' = rsqrt / 2)
  U_RSQRT3 = 22,  // t^-3/2   (rsqrt' = -rsqrt3 / 2)
  U_COUNT = 23
};

__device__ __forceinline__ float sigmoidf(float t) { return 1.0f / (1.0f + expf(-t)); }

__device__ __forceinline__ float unary_apply(int kind, float x, float alpha, float beta) {
  const float t = alpha * x + beta;
  switch (kind) {
    case U_AFFINE: return t;
    case U_EXP: return expf(t);
    case U_COS: return cosf(t);
    case U_SIN: return sinf(t);
    case U_SSP: return (t > 20.0f ? t : log1pf(expf(t))) - 0.69314718055994531f;
    case U_SIGMOID: return sigmoidf(t);
    case U_DSIGMOID: { const float s = sigmoidf(t); return s * (1.0f - s); }
    case U_D2SIGMOID: { const float s = sigmoidf(t); return s * (1.0f - s) * (1.0f - 2.0f * s); }
    case U_RECIP: return 1.0f / t;
    case U_SQRT: return sqrtf(t);
    case U_SILU: return t * sigmoidf(t);
    case U_DSILU: { const float s = sigmoidf(t); return s * (1.0f + t * (1.0f - s)); }
    case U_D2SILU: { const float s = sigmoidf(t); return s * (1.0f - s) * (2.0f + t * (1.0f - 2.0f * s)); }
    case U_GAUSS: return expf(alpha * x * x);
    case U_DGAUSS: return 2.0f * alpha * x * expf(alpha * x * x);
    case U_D2GAUSS: { const float ax = alpha * x; return (2.0f * alpha + 4.0f * ax * ax) * expf(ax * x); }
    case U_LT: return x < alpha ? 1.0f : 0.0f;
    case U_ABS: return fabsf(t);
    case U_SIGN: return t > 0.0f ? 1.0f : (t < 0.0f ? -1.0f : 0.0f);
    case U_DRECIP: return -1.0f / (t * t);
    case U_D2RECIP: return 2.0f / (t * t * t);
    case U_RSQRT: return 1.0f / sqrtf(t);
    case U_RSQRT3: return 1.0f / (t * sqrtf(t));
  }
  return 0.0f;
}

__global__ void k_tape_unary(int kind, const float* __restrict__ x, int64_t n, float alpha, float beta,
                             float* __restrict__ y, int vec) {
  const int64_t n4 = vec ? n >> 2 : 0, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += step) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 o;
    o.x = unary_apply(kind, v.x, alpha, beta);
    o.y = unary_apply(kind, v.y, alpha, beta);
    o.z = unary_apply(kind, v.z, alpha, beta);
    o.w = unary_apply(kind, v.w, alpha, beta);
    reinterpret_cast<f32x4*>(y)[i] = o;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
    y[i] = unary_apply(kind, x[i], alpha, beta);
}

// ------------------------------------------------------------------------------------------------ broadcast arithmetic
// y[r][c] = scale * (A(r, c) op B(r, c)); each operand is addressed by its mode:
enum Mode {
  M_FULL = 0,   // [R][D]
  M_ROW = 1,    // [R]: one scalar per row
  M_COL = 2,    // [D]: one scalar per column
  M_THIRD = 3   // [R / 3][D]: the row r / 3 (a per-atom / per-edge row against the three xyz rows)
};
enum Binary { B_ADD = 0, B_SUB = 1, B_MUL = 2, B_FIRST = 3 /* scale * A: a broadcast written out */ };

__device__ __forceinline__ float operand(const float* __restrict__ p, int mode, int64_t r, int c, int D) {
  switch (mode) {
    case M_FULL: return p[r * D + c];
    case M_ROW: return p[r];
    case M_COL: return p[c];
    default: return p[(r / 3) * D + c];
  }
}

__global__ void k_tape_binary(int op, const float* __restrict__ A, int am, const float* __restrict__ Bv, int bm,
                              int64_t R, int D, float scale, float* __restrict__ y) {
  const int64_t n = R * D, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const int64_t r = i / D;
    const int c = (int)(i - r * D);
    const float a = operand(A, am, r, c, D);
    float v;
    if (op == B_FIRST) v = a;
    else {
      const float b = operand(Bv, bm, r, c, D);
      v = op == B_ADD ? a + b : (op == B_SUB ? a - b : a * b);
    }
    y[i] = scale * v;
  }
}

// the same with four columns per thread (D a multiple of 4, 16-byte aligned operands, fewer than 2^31 elements): one
// 32-bit division per four outputs, 16-byte accesses for every operand that has columns
__device__ __forceinline__ f32x4 operand4(const float* __restrict__ p, int mode, int r, int c, int D) {
  switch (mode) {
    case M_FULL: return *reinterpret_cast<const f32x4*>(p + (int64_t)r * D + c);
    case M_ROW: { const float v = p[r]; return f32x4{v, v, v, v}; }
    case M_COL: return *reinterpret_cast<const f32x4*>(p + c);
    default: return *reinterpret_cast<const f32x4*>(p + (int64_t)(r / 3) * D + c);
  }
}
__global__ void k_tape_binary4(int op, const float* __restrict__ A, int am, const float* __restrict__ Bv, int bm, int R,
                               int D, float scale, float* __restrict__ y) {
  const int Dq = D >> 2;
  const uint32_t n = (uint32_t)R * (uint32_t)Dq, step = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const int r = (int)(i / (uint32_t)Dq), c = (int)(i - (uint32_t)r * (uint32_t)Dq) << 2;
    const f32x4 a = operand4(A, am, r, c, D);
    f32x4 v = a;
    if (op != B_FIRST) {
      const f32x4 b = operand4(Bv, bm, r, c, D);
      v = op == B_ADD ? a + b : (op == B_SUB ? a - b : a * b);
    }
    *reinterpret_cast<f32x4*>(y + (int64_t)r * D + c) = v * scale;
  }
}

// both operands full: 16-byte accesses
__global__ void k_tape_binary_full(int op, const float* __restrict__ A, const float* __restrict__ Bv, int64_t n,
                                   float scale, float* __restrict__ y) {
  const int64_t n4 = n >> 2, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += step) {
    const f32x4 a = reinterpret_cast<const f32x4*>(A)[i], b = reinterpret_cast<const f32x4*>(Bv)[i];
    const f32x4 v = op == B_ADD ? a + b : (op == B_SUB ? a - b : a * b);
    reinterpret_cast<f32x4*>(y)[i] = v * scale;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const float a = A[i], b = Bv[i];
    y[i] = scale * (op == B_ADD ? a + b : (op == B_SUB ? a - b : a * b));
  }
}

// ------------------------------------------------------------------------------------------------------------ reductions
// row sums: one wave per row, a lane's columns in ascending order, then a butterfly (fixed order)
__global__ __launch_bounds__(256) void k_tape_rowsum(const float* __restrict__ x, int64_t R, int D,
                                                     float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < R; r += nw) {
    float s = 0.0f;
    for (int c = lane; c < D; c += 64) s += x[r * D + c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[r] = s;
  }
}

// narrow rows (xyz rows, D <= 16): one thread per row
__global__ void k_tape_rowsum_narrow(const float* __restrict__ x, int64_t R, int D, float* __restrict__ y) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int c = 0; c < D; ++c) s += x[r * D + c];
    y[r] = s;
  }
}

// column sums, stage 1: block (x, y) owns the rows [x * chunk, (x + 1) * chunk) of the 64 columns [64 y, 64 y + 64): its
// four waves take every fourth row (two running sums each), then add up in wave order; stage 2 adds the blocks' partial
// sums in block order
__global__ __launch_bounds__(256) void k_tape_colsum_partial(const float* __restrict__ x, int64_t R, int D, int64_t chunk,
                                                             float* __restrict__ partial) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, c = blockIdx.y * 64 + tx;
  const int64_t r0 = (int64_t)blockIdx.x * chunk, r1 = r0 + chunk < R ? r0 + chunk : R;
  float s0 = 0.0f, s1 = 0.0f;
  if (c < D) {
    int64_t r = r0 + ty;
    for (; r + 4 < r1; r += 8) {
      s0 += x[r * D + c];
      s1 += x[(r + 4) * D + c];
    }
    if (r < r1) s0 += x[r * D + c];
  }
  part[ty][tx] = s0 + s1;
  __syncthreads();
  if (ty == 0 && c < D) partial[(int64_t)blockIdx.x * D + c] = ((part[0][tx] + part[1][tx]) + part[2][tx]) + part[3][tx];
}
// stage 2: block (16 x 64): thread (ty, tx) adds the partial sums b = ty, ty + 16, ... of column c in ascending b, then the
// sixteen partial results are added in ty order - a fixed order whatever the grid (round 6: the one-thread-per-column
// form walked up to 1024 dependent loads: 20-40 us for a 5 us reduction)
__global__ __launch_bounds__(1024) void k_tape_colsum_final(const float* __restrict__ partial, int nb, int D,
                                                             float* __restrict__ y) {
  __shared__ float part[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, c = blockIdx.x * 64 + tx;
  float s = 0.0f;
  if (c < D)
    for (int b = ty; b < nb; b += 16) s += partial[(int64_t)b * D + c];
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < D) {
    float t = part[0][tx];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += part[k][tx];
    y[c] = t;
  }
}

// sums over the xyz triple: y[e][c] = x[3e][c] + x[3e+1][c] + x[3e+2][c]
__global__ void k_tape_sum3(const float* __restrict__ x, int64_t R3, int D, float* __restrict__ y) {
  const int64_t n = R3 * D, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const int64_t e = i / D;
    const int c = (int)(i - e * D);
    const float* p = x + 3 * e * D + c;
    y[i] = (p[0] + p[D]) + p[2 * D];
  }
}

// -------------------------------------------------------------------------------------------------- gather and its adjoint
template <typename I>
__global__ void k_tape_gather(const float* __restrict__ src, const I* __restrict__ idx, int64_t E, int D,
                              float* __restrict__ out) {
  const int64_t n = E * D, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const int64_t e = i / D;
    const int c = (int)(i - e * D);
    out[i] = src[(int64_t)idx[e] * D + c];
  }
}
// out[n] = sum of the rows src[perm[k]], k in [ptr[n], ptr[n+1]) - the incidence list of target row n, ascending
template <typename P>
__global__ void k_tape_scatter(const float* __restrict__ src, const P* __restrict__ ptr,
                               const int32_t* __restrict__ perm, int64_t N, int D, float* __restrict__ out) {
  const int64_t n = N * D, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const int64_t t = i / D;
    const int c = (int)(i - t * D);
    float s = 0.0f;
    for (int64_t k = ptr[t], k1 = ptr[t + 1]; k < k1; ++k) s += src[(int64_t)perm[k] * D + c];
    out[i] = s;
  }
}
// The same sum for FEW target rows with LONG lists (a readout's adjoint, per-column statistics: N D threads would not fill
// a CU and each would walk thousands of rows): one block per (target row, 64-column slab), sixteen sub-lists k = ty,
// ty + 16, ... in ascending k, added in ty order - fixed, but not the order of the form above (which one a call takes
// depends on N and D only).
template <typename P>
__global__ __launch_bounds__(1024) void k_tape_scatter_long(const float* __restrict__ src, const P* __restrict__ ptr,
                                                             const int32_t* __restrict__ perm, int D,
                                                             float* __restrict__ out) {
  __shared__ float part[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, c = blockIdx.y * 64 + tx;
  const int64_t t = blockIdx.x;
  float s = 0.0f;
  if (c < D)
    for (int64_t k = (int64_t)ptr[t] + ty, k1 = ptr[t + 1]; k < k1; k += 16) s += src[(int64_t)perm[k] * D + c];
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < D) {
    float v = part[0][tx];
#pragma unroll
    for (int k = 1; k < 16; ++k) v += part[k][tx];
    out[t * D + c] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------ block copies
__global__ void k_tape_copy2d(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd, int64_t R,
                              int C) {
  const int64_t n = R * C, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    dst[r * ldd + c] = src[r * lds + c];
  }
}
__global__ void k_tape_fill(float* __restrict__ dst, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = v;
}

}  // namespace
}  // namespace geossl

using namespace geossl;

extern "C" int geossl_tape_unary(int kind, const float* x, int64_t n, float alpha, float beta, float* y,
                                 hipStream_t stream) {
  if (n <= 0) return 0;
  if (kind < 0 || kind >= U_COUNT) return (int)hipErrorInvalidValue;
  const int vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;  // 16-byte accesses
  hipLaunchKernelGGL(k_tape_unary, dim3(tape_grid(vec ? (n + 3) / 4 : n, 256)), dim3(256), 0, stream, kind, x, n, alpha,
                     beta, y, vec);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_tape_binary(int op, const float* a, int amode, const float* b, int bmode, int64_t R, int D,
                                  float scale, float* y, hipStream_t stream) {
  if (R <= 0 || D <= 0) return 0;
  if (op < B_ADD || op > B_FIRST || amode < M_FULL || amode > M_THIRD) return (int)hipErrorInvalidValue;
  if (op != B_FIRST && (b == nullptr || bmode < M_FULL || bmode > M_THIRD)) return (int)hipErrorInvalidValue;
  if ((amode == M_THIRD || (op != B_FIRST && bmode == M_THIRD)) && R % 3) return (int)hipErrorInvalidValue;
  const int64_t n = R * D;
  const bool aligned = !((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) & 15);
  if (op != B_FIRST && amode == M_FULL && bmode == M_FULL && aligned)
    hipLaunchKernelGGL(k_tape_binary_full, dim3(tape_grid((n + 3) / 4, 256)), dim3(256), 0, stream, op, a, b, n, scale, y);
  else if (aligned && (D & 3) == 0 && n < ((int64_t)1 << 31))
    hipLaunchKernelGGL(k_tape_binary4, dim3(tape_grid(n / 4, 256)), dim3(256), 0, stream, op, a, amode, b, bmode, (int)R, D,
                       scale, y);
  else
    hipLaunchKernelGGL(k_tape_binary, dim3(tape_grid(n, 256)), dim3(256), 0, stream, op, a, amode, b, bmode, R, D, scale, y);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

// kind: 1 row sums [R][D] -> [R], 2 column sums -> [D], 3 xyz-triple sums [R][D] -> [R / 3][D] (the Mode values).
// Column sums need `workspace` of geossl_tape_colsum_workspace_floats(R, D) floats.
extern "C" int64_t geossl_tape_colsum_workspace_floats(int64_t R, int D) {
  const int64_t nb = R <= 0 ? 1 : (R + 63) / 64 > 2048 ? 2048 : (R + 63) / 64;   // 64-row chunks: enough blocks to fill the chip
  return nb * (int64_t)D;
}
extern "C" int geossl_tape_reduce(int kind, const float* x, int64_t R, int D, float* y, float* workspace,
                                  hipStream_t stream) {
  if (D <= 0) return 0;
  if (kind == M_ROW) {
    if (R <= 0) return 0;
    if (D <= 16) hipLaunchKernelGGL(k_tape_rowsum_narrow, dim3(tape_grid(R, 256)), dim3(256), 0, stream, x, R, D, y);
    else hipLaunchKernelGGL(k_tape_rowsum, dim3(tape_grid(R, 4)), dim3(256), 0, stream, x, R, D, y);
  } else if (kind == M_COL) {
    if (R <= 0) {
      hipLaunchKernelGGL(k_tape_fill, dim3(tape_grid(D, 256)), dim3(256), 0, stream, y, (int64_t)D, 0.0f);
    } else {
      if (workspace == nullptr) return (int)hipErrorInvalidValue;
      const int64_t nb = geossl_tape_colsum_workspace_floats(R, D) / D, chunk = (R + nb - 1) / nb;
      hipLaunchKernelGGL(k_tape_colsum_partial, dim3((int)nb, (D + 63) / 64), dim3(256), 0, stream, x, R, D, chunk,
                         workspace);
      GEOSSL_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_tape_colsum_final, dim3((D + 63) / 64), dim3(1024), 0, stream, workspace, (int)nb, D, y);
    }
  } else if (kind == M_THIRD) {
    if (R % 3) return (int)hipErrorInvalidValue;
    if (R <= 0) return 0;
    hipLaunchKernelGGL(k_tape_sum3, dim3(tape_grid(R / 3 * D, 256)), dim3(256), 0, stream, x, R / 3, D, y);
  } else {
    return (int)hipErrorInvalidValue;
  }
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_tape_gather_rows(const float* src, const void* idx, int idx64, int64_t E, int D, float* out,
                                       hipStream_t stream) {
  if (E <= 0 || D <= 0) return 0;
  if (idx64)
    hipLaunchKernelGGL(k_tape_gather<int64_t>, dim3(tape_grid(E * D, 256)), dim3(256), 0, stream, src,
                       static_cast<const int64_t*>(idx), E, D, out);
  else
    hipLaunchKernelGGL(k_tape_gather<int32_t>, dim3(tape_grid(E * D, 256)), dim3(256), 0, stream, src,
                       static_cast<const int32_t*>(idx), E, D, out);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_tape_scatter_rows(const float* src, const void* ptr, int ptr64, const int32_t* perm, int64_t N,
                                        int D, float* out, hipStream_t stream) {
  if (N <= 0 || D <= 0) return 0;
  if (N * (int64_t)((D + 63) / 64) <= 4096 && N <= 65535) {   // few target rows: a block per row and column slab
    const dim3 grid((unsigned)N, (unsigned)((D + 63) / 64));
    if (ptr64)
      hipLaunchKernelGGL(k_tape_scatter_long<int64_t>, grid, dim3(1024), 0, stream, src, static_cast<const int64_t*>(ptr),
                         perm, D, out);
    else
      hipLaunchKernelGGL(k_tape_scatter_long<int32_t>, grid, dim3(1024), 0, stream, src, static_cast<const int32_t*>(ptr),
                         perm, D, out);
    GEOSSL_CHECK_LAUNCH();
    return 0;
  }
  if (ptr64)
    hipLaunchKernelGGL(k_tape_scatter<int64_t>, dim3(tape_grid(N * D, 256)), dim3(256), 0, stream, src,
                       static_cast<const int64_t*>(ptr), perm, N, D, out);
  else
    hipLaunchKernelGGL(k_tape_scatter<int32_t>, dim3(tape_grid(N * D, 256)), dim3(256), 0, stream, src,
                       static_cast<const int32_t*>(ptr), perm, N, D, out);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_tape_copy2d(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t R, int C,
                                  hipStream_t stream) {
  if (R <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(k_tape_copy2d, dim3(tape_grid(R * C, 256)), dim3(256), 0, stream, src, ld_src, dst, ld_dst, R, C);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_tape_fill(float* dst, int64_t n, float value, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_tape_fill, dim3(tape_grid(n, 256)), dim3(256), 0, stream, dst, n, value);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
