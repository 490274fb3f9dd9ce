// Batch layout, radius graph and super-edge bookkeeping (integer / index side of the path).
//
// One wave per molecule: positions of the molecule are staged in LDS, every target atom scans its sources in
// ascending order 64 at a time (ballot + popcount gives the running hit count for the neighbour cap), and
// the adjacency is kept as an n x ceil(n/64) bit matrix in LDS.  All outputs are written in the canonical
// order (target-major, sources ascending) without atomics, so edge lists are bit-reproducible.
#include "common.h"
#include "geossl_hip.h"

using namespace geossl;

namespace {

__device__ __forceinline__ float dist2_nofma(const float* pi, const float* pj) {
  // fl32(fl32(fl32(dx*dx)+fl32(dy*dy))+fl32(dz*dz)), d = x_j - x_i, nothing contracted (common.h)
  return norm2_rn(pj[0] - pi[0], pj[1] - pi[1], pj[2] - pi[2]);
}

// ---------------------------------------------------------------------------------------------- layout
__global__ __launch_bounds__(1024) void k_layout_build(const int64_t* __restrict__ batch, int N, int B,
                                                       int32_t* __restrict__ mol_ptr, int32_t* __restrict__ pair_ptr,
                                                       int64_t* __restrict__ stats) {
  __shared__ int s_scan[1024];
  __shared__ int s_carry, s_bad, s_maxn;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_carry = 0;
    s_bad = 0;
    s_maxn = 0;
  }
  __syncthreads();
  int bad = 0;
  for (int a = tid; a < N; a += 1024) {
    const int64_t g = batch[a];
    if (g < 0 || g >= B || (a > 0 && batch[a - 1] > g)) bad = 1;
  }
  if (bad) atomicOr(&s_bad, 1);
  for (int g = tid; g <= B; g += 1024) {  // mol_ptr[g] = first atom with batch >= g
    int lo = 0, hi = N;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (batch[mid] < g) lo = mid + 1; else hi = mid;
    }
    mol_ptr[g] = lo;
  }
  __syncthreads();
  __threadfence_block();
  int maxn = 0;
  for (int base = 0; base < B; base += 1024) {
    const int g = base + tid;
    int n = 0;
    if (g < B) n = mol_ptr[g + 1] - mol_ptr[g];
    maxn = max(maxn, n);
    const int v = n * (n - 1) / 2;
    s_scan[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // inclusive Hillis-Steele scan
      const int add = tid >= o ? s_scan[tid - o] : 0;
      __syncthreads();
      s_scan[tid] += add;
      __syncthreads();
    }
    const int carry = s_carry;
    if (g < B) pair_ptr[g] = carry + s_scan[tid] - v;
    __syncthreads();
    if (tid == 1023) s_carry = carry + s_scan[1023];
    __syncthreads();
  }
  atomicMax(&s_maxn, maxn);
  __syncthreads();
  if (tid == 0) {
    pair_ptr[B] = s_carry;
    stats[0] = s_maxn;
    stats[1] = s_carry;
    stats[2] = s_bad;
    stats[3] = 0;
  }
}

__global__ void k_pair_index_fill(const int32_t* __restrict__ mol_ptr, const int32_t* __restrict__ pair_ptr, int B,
                                  int32_t* __restrict__ pair_i, int32_t* __restrict__ pair_j) {
  const int m = blockIdx.x;
  if (m >= B) return;
  const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0, base = pair_ptr[m];
  for (int a = 0; a + 1 < n; ++a) {
    const int row = base + a * n - a * (a + 1) / 2 - a - 1;  // slot(a,b) = row + b
    for (int b = a + 1 + threadIdx.x; b < n; b += blockDim.x) {
      pair_i[row + b] = a0 + a;
      pair_j[row + b] = a0 + b;
    }
  }
}

// Device-side AtomTupleExtractor (dataloaders_AtomTuple.py:15-37, ratio = 1) + the node offset added by
// BatchAtomTuple.from_data_list (:64-65).  option 0 = "combination": the n(n-1)/2 pairs i<j in lexicographic order
// (itertools.combinations); option 1 = "permutation": the n(n-1) ordered pairs i != j in itertools.permutations
// order ((0,1),(0,2),...,(1,0),(1,2),...).  tuple_ptr[m] = first output column of molecule m.
__global__ void k_atom_tuples(const int32_t* __restrict__ mol_ptr, const int64_t* __restrict__ tuple_ptr, int B,
                              int option, int64_t* __restrict__ out0, int64_t* __restrict__ out1) {
  const int m = blockIdx.x;
  if (m >= B) return;
  const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0;
  const int64_t base = tuple_ptr[m];
  if (option == 0) {
    for (int a = 0; a + 1 < n; ++a) {
      const int64_t row = base + a * n - a * (a + 1) / 2 - a - 1;  // column of (a, b) = row + b
      for (int b = a + 1 + threadIdx.x; b < n; b += blockDim.x) {
        out0[row + b] = a0 + a;
        out1[row + b] = a0 + b;
      }
    }
  } else {
    for (int a = 0; a < n; ++a) {
      const int64_t row = base + (int64_t)a * (n - 1);
      for (int b = threadIdx.x; b < n; b += blockDim.x) {
        if (b == a) continue;
        const int64_t c = row + (b < a ? b : b - 1);
        out0[c] = a0 + a;
        out1[c] = a0 + b;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------- radius graph
// MODE 0: in-degree; MODE 1: edge list fill; MODE 2: pair-slot geometry.  One 64-lane block per molecule.
template <int MODE>
__global__ __launch_bounds__(64) void k_radius(const float* __restrict__ pos, const int32_t* __restrict__ mol_ptr,
                                               const int32_t* __restrict__ pair_ptr, int B, int max_n, float r2, int cap,
                                               int32_t* __restrict__ deg, const int64_t* __restrict__ edge_ptr,
                                               int64_t* __restrict__ edge_src, int64_t* __restrict__ edge_dst,
                                               float* __restrict__ edge_weight, float* __restrict__ pair_d,
                                               uint8_t* __restrict__ pair_flag, float cutoff,
                                               float* __restrict__ pair_c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int m = blockIdx.x;
  if (m >= B) return;
  const int lane = threadIdx.x;
  const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0;
  const int words = (max_n + 63) / 64;
  unsigned long long* adj = reinterpret_cast<unsigned long long*>(smem_raw);  // [max_n][words]
  float* sp = reinterpret_cast<float*>(adj + (size_t)max_n * words);         // [max_n][3]
  for (int i = lane; i < 3 * n; i += 64) sp[i] = pos[(size_t)a0 * 3 + i];
  __syncthreads();
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int i = 0; i < n; ++i) {
    int found = 0, d_in = 0;
    int64_t ebase = 0;
    if (MODE == 1) ebase = edge_ptr[a0 + i];
    for (int c = 0; c * 64 < n; ++c) {
      const int jn = c * 64 + lane;
      float d2 = 0.0f;
      bool hit = false;
      if (jn < n) {
        d2 = dist2_nofma(sp + 3 * i, sp + 3 * jn);
        hit = d2 < r2;
      }
      const unsigned long long mask = __ballot(hit);
      const int rank = found + __popcll(mask & lt);
      bool keep = hit && rank < cap;
      found += __popcll(mask);
      if (jn == i) keep = false;  // self edge dropped after the cap was applied
      const unsigned long long kept = __ballot(keep);
      if (MODE == 2 && lane == 0) adj[(size_t)i * words + c] = kept;
      if (MODE == 1 && keep) {
        const int64_t e = ebase + d_in + __popcll(kept & lt);
        edge_src[e] = a0 + jn;
        edge_dst[e] = a0 + i;
        edge_weight[e] = sqrtf(d2);
      }
      d_in += __popcll(kept);
    }
    if (MODE == 0 && lane == 0) deg[a0 + i] = d_in;
  }
  if (MODE == 2) {
    __syncthreads();
    const int base = pair_ptr[m];
    for (int a = 0; a + 1 < n; ++a) {
      const int row = base + a * n - a * (a + 1) / 2 - a - 1;
      for (int b = a + 1 + lane; b < n; b += 64) {
        const float d2 = dist2_nofma(sp + 3 * a, sp + 3 * b);
        const unsigned f0 = (adj[(size_t)a * words + (b >> 6)] >> (b & 63)) & 1ull;  // edge b -> a (target a)
        const unsigned f1 = (adj[(size_t)b * words + (a >> 6)] >> (a & 63)) & 1ull;  // edge a -> b (target b)
        const float d = sqrtf(d2);
        pair_d[row + b] = d;
        // CFConv envelope, schnet.py:186: 0.5 * (cos(d * PI / cutoff) + 1.0), fp32 op by op
        pair_c[row + b] = 0.5f * (cosf(mul_rn(d, GEOSSL_PI_F) / cutoff) + 1.0f);
        pair_flag[row + b] = (uint8_t)(f0 | (f1 << 1));
      }
    }
  }
}

// MODE 2 for molecules of at most `cap` atoms (the neighbour cap of torch_cluster.radius_graph, self hit included, can
// then never cut a list: the flags are the plain threshold test, symmetric bit for bit because the squared distance is):
// no adjacency pass, the pair slots of a molecule dealt flat to the 64 lanes of its wave - 3 rounds for 18 atoms where
// the general kernel walks 18 + 17 dependent rows.  Same arithmetic, same outputs.
__global__ __launch_bounds__(64) void k_pair_geometry_flat(const float* __restrict__ pos, const int32_t* __restrict__ mol_ptr,
                                                           const int32_t* __restrict__ pair_ptr, int B, float r2,
                                                           float cutoff, float* __restrict__ pair_d,
                                                           float* __restrict__ pair_c, uint8_t* __restrict__ pair_flag) {
  __shared__ float sp[3 * 64];
  const int m = blockIdx.x, lane = threadIdx.x;
  if (m >= B) return;
  const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0;
  const int base = pair_ptr[m], np = n * (n - 1) / 2;
  for (int i = lane; i < 3 * n; i += 64) sp[i] = pos[(size_t)a0 * 3 + i];
  __syncthreads();
  const float tn = (float)(2 * n - 1);
  for (int q = lane; q < np; q += 64) {
    // slot q = a n - a (a + 1) / 2 + (b - a - 1), a < b: the row from the closed form, corrected by one either way
    int a = (int)((tn - sqrtf(fmaxf(tn * tn - 8.0f * (float)q, 0.0f))) * 0.5f);
    a = min(max(a, 0), n - 2);
    if (a * n - a * (a + 1) / 2 > q) --a;
    else if ((a + 1) * n - (a + 1) * (a + 2) / 2 <= q) ++a;
    const int b = q - (a * n - a * (a + 1) / 2) + a + 1;
    const float d2 = dist2_nofma(sp + 3 * a, sp + 3 * b);
    const float d = sqrtf(d2);
    pair_d[base + q] = d;
    pair_c[base + q] = 0.5f * (cosf(mul_rn(d, GEOSSL_PI_F) / cutoff) + 1.0f);  // schnet.py:186, fp32 op by op
    pair_flag[base + q] = d2 < r2 ? (uint8_t)3 : (uint8_t)0;
  }
}

inline size_t radius_lds(int max_n) {
  const size_t words = (max_n + 63) / 64;
  return (size_t)max_n * words * 8 + (size_t)max_n * 12;
}

// ------------------------------------------------------------------------------- super-edge bookkeeping
__global__ void k_super_edge_ptr(const int64_t* __restrict__ batch, const int64_t* __restrict__ sei0,
                                 const int64_t* __restrict__ sei1, int S, int B, int32_t* __restrict__ se_ptr,
                                 int64_t* __restrict__ stats) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  if (tid == 0) stats[0] = S > 0 ? batch[sei0[S - 1]] + 1 : 0;  // max(edge2graph)+1 under the ordering assumption
  for (int s = tid; s < S; s += nth) {
    const int64_t g = batch[sei0[s]];
    if (batch[sei1[s]] != g || (s > 0 && batch[sei0[s - 1]] > g)) stats[1] = 1;
  }
  for (int g = tid; g <= B; g += nth) {
    int lo = 0, hi = S;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (batch[sei0[mid]] < g) lo = mid + 1; else hi = mid;
    }
    se_ptr[g] = lo;
  }
}

// one wave per atom: entries ordered by super-edge id, the u-side entry before the v-side entry
template <int FILL>
__global__ __launch_bounds__(256) void k_incidence(const int64_t* __restrict__ batch, const int64_t* __restrict__ sei0,
                                                   const int64_t* __restrict__ sei1, const int32_t* __restrict__ se_ptr,
                                                   int N, int sides, int32_t* __restrict__ inc_cnt,
                                                   const int64_t* __restrict__ inc_ptr, int32_t* __restrict__ inc_idx) {
  const int a = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (a >= N) return;
  const int g = (int)batch[a];
  const int s0 = se_ptr[g], s1 = se_ptr[g + 1];
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int count = 0;
  const int64_t base = FILL ? inc_ptr[a] : 0;
  for (int s = s0 + lane; s - lane < s1; s += 64) {
    bool mu = false, mv = false;
    if (s < s1) {
      mu = (sides & 1) && sei0[s] == a;
      mv = (sides & 2) && sei1[s] == a;
    }
    const unsigned long long bu = __ballot(mu), bv = __ballot(mv);
    if (FILL) {
      const int before = count + __popcll(bu & lt) + __popcll(bv & lt);
      if (mu) inc_idx[base + before] = s;
      if (mv) inc_idx[base + before + (mu ? 1 : 0)] = s;
    }
    count += __popcll(bu) + __popcll(bv);
  }
  if (!FILL && lane == 0) inc_cnt[a] = count;
}

__global__ void k_pair_distance(const float* __restrict__ pos, const int64_t* __restrict__ sei0,
                                const int64_t* __restrict__ sei1, int S, float* __restrict__ out) {
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += gridDim.x * blockDim.x) {
    const int64_t u = sei0[s], v = sei1[s];
    out[s] = sqrtf(norm2_rn(pos[3 * u] - pos[3 * v], pos[3 * u + 1] - pos[3 * v + 1], pos[3 * u + 2] - pos[3 * v + 2]));
  }
}

// The two views of a DDM step in one launch (pretrain_GeoSSL.py:68-74,199-205): pos2 = [pos ; pos + noise] (the fused
// batch of both views) and the super-edge lengths in either view.  Same arithmetic as k_axpy (alpha = 1) followed by
// k_pair_distance on each half: the perturbed coordinates are rounded to fp32 before the differences are taken.
__global__ void k_ddm_views(const float* __restrict__ pos, const float* __restrict__ noise,
                            const int64_t* __restrict__ sei0, const int64_t* __restrict__ sei1, int64_t n3, int S,
                            float* __restrict__ pos2, float* __restrict__ d01, float* __restrict__ d02,
                            const int64_t* __restrict__ z, int64_t zs, int64_t* __restrict__ z2,
                            const int32_t* __restrict__ dyn_N, const int32_t* __restrict__ dyn_S) {
  // (capacity launch: the second view starts right behind the REAL atoms of the first)
  n3 = 3 * (int64_t)dyn_count((int)(n3 / 3), dyn_N);
  S = dyn_count(S, dyn_S);
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
  if (z2 != nullptr)  // atom types of the two-view batch: the column z[:, 0] twice
    for (int64_t i = t; i < n3 / 3; i += nt) {
      const int64_t v = z[i * zs];
      z2[i] = v;
      z2[n3 / 3 + i] = v;
    }
  for (int64_t i = t; i < n3; i += nt) {
    const float a = pos[i];
    pos2[i] = a;
    pos2[n3 + i] = add_rn(a, noise[i]);
  }
  for (int64_t s = t; s < S; s += nt) {
    const int64_t u = sei0[s], v = sei1[s];
    const float ux = pos[3 * u], uy = pos[3 * u + 1], uz = pos[3 * u + 2];
    const float vx = pos[3 * v], vy = pos[3 * v + 1], vz = pos[3 * v + 2];
    d01[s] = sqrtf(norm2_rn(ux - vx, uy - vy, uz - vz));
    const float px = add_rn(ux, noise[3 * u]), py = add_rn(uy, noise[3 * u + 1]), pz = add_rn(uz, noise[3 * u + 2]);
    const float qx = add_rn(vx, noise[3 * v]), qy = add_rn(vy, noise[3 * v + 1]), qz = add_rn(vz, noise[3 * v + 2]);
    d02[s] = sqrtf(norm2_rn(px - qx, py - qy, pz - qz));
  }
}

// two device-to-device copies in one launch (the per-step refresh of a replayed graph's inputs: atom types and
// positions); sizes in bytes, multiples of 4, buffers 4-byte aligned
__global__ void k_copy2(uint32_t* __restrict__ d0, const uint32_t* __restrict__ s0, int64_t n0,
                        uint32_t* __restrict__ d1, const uint32_t* __restrict__ s1, int64_t n1) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = t; i < n0; i += nt) d0[i] = s0[i];
  for (int64_t i = t; i < n1; i += nt) d1[i] = s1[i];
}

// ---------------------------------------------------------------------------------------------------------------
// The five random draws of a DDM step in ONE launch (perturb pretrain_GeoSSL.py:72: N(mu, sigma) per coordinate; per
// head a noise level per molecule NCSN.py:190 and N(0, 1) per super-edge :194) for a caller that owns its random
// stream (DDMTrainer with device noise): counter-based Philox4x32-10 keyed by a 64-bit seed read from the device (the
// caller draws it from torch's generator, so torch.cuda.manual_seed governs the stream), counter = (group of four
// outputs, segment); normals by Box-Muller.  Not the values torch's own calls would give - the reference's loop keeps
// those (do_DDM).
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ void normal4(uint4 r, float (&z)[4]) {
  const float u0 = ((float)(r.x >> 8) + 1.0f) * (1.0f / 16777216.0f), u1 = (float)(r.y >> 8) * (1.0f / 16777216.0f);
  const float u2 = ((float)(r.z >> 8) + 1.0f) * (1.0f / 16777216.0f), u3 = (float)(r.w >> 8) * (1.0f / 16777216.0f);
  const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
  float sa, ca, sb, cb;
  sincosf(6.283185307179586f * u1, &sa, &ca);
  sincosf(6.283185307179586f * u3, &sb, &cb);
  z[0] = ra * ca; z[1] = ra * sa; z[2] = rb * cb; z[3] = rb * sb;
}
__global__ void k_ddm_noise(const int64_t* __restrict__ seed, uint64_t seed_value, float mu, float sigma, int64_t n_pos,
                            int64_t S, int64_t B, int K1, int K2, float* __restrict__ pos_noise,
                            int64_t* __restrict__ nl1, float* __restrict__ dn1, int64_t* __restrict__ nl2,
                            float* __restrict__ dn2) {
  const uint64_t sd = seed != nullptr ? (uint64_t)seed[0] : seed_value;
  const uint2 key = make_uint2((uint32_t)sd, (uint32_t)(sd >> 32));
  const int64_t g_pos = (n_pos + 3) / 4, g_s = (S + 3) / 4, g_b = (B + 3) / 4;
  const int64_t total = g_pos + 2 * g_s + 2 * g_b;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int seg;
    int64_t g = t;
    if (g < g_pos) seg = 0;
    else if ((g -= g_pos) < g_s) seg = 1;
    else if ((g -= g_s) < g_s) seg = 2;
    else if ((g -= g_s) < g_b) seg = 3;
    else { g -= g_b; seg = 4; }
    const uint4 r = philox4x32_10(make_uint4((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)seg, 0u), key);
    if (seg <= 2) {
      float z[4];
      normal4(r, z);
      float* dst = seg == 0 ? pos_noise : (seg == 1 ? dn1 : dn2);
      const int64_t n = seg == 0 ? n_pos : S;
      const float m = seg == 0 ? mu : 0.0f, sg = seg == 0 ? sigma : 1.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * g + e < n) dst[4 * g + e] = add_rn(mul_rn(z[e], sg), m);
    } else {
      int64_t* dst = seg == 3 ? nl1 : nl2;
      const uint32_t K = (uint32_t)(seg == 3 ? K1 : K2);
      const uint32_t v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * g + e < B) dst[4 * g + e] = (int64_t)(v[e] % K);
    }
  }
}

__global__ void k_axpy(const float* __restrict__ a, const float* __restrict__ b, float alpha, int64_t n,
                       float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a[i] + alpha * b[i];
}

inline int grid1d(int64_t n, int block, int cap = 2048) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int geossl_layout_build(const int64_t* batch, int64_t N, int64_t B, int32_t* mol_ptr, int32_t* pair_ptr,
                                   int64_t* stats, hipStream_t stream) {
  hipLaunchKernelGGL(k_layout_build, dim3(1), dim3(1024), 0, stream, batch, (int)N, (int)B, mol_ptr, pair_ptr, stats);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_pair_index_fill(const int32_t* mol_ptr, const int32_t* pair_ptr, int64_t B, int32_t* pair_i,
                                      int32_t* pair_j, hipStream_t stream) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(k_pair_index_fill, dim3((unsigned)B), dim3(64), 0, stream, mol_ptr, pair_ptr, (int)B, pair_i,
                     pair_j);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_atom_tuples(const int32_t* mol_ptr, const int64_t* tuple_ptr, int64_t B, int option,
                                  int64_t* out0, int64_t* out1, hipStream_t stream) {
  if (B <= 0) return 0;
  if (option != 0 && option != 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_atom_tuples, dim3((unsigned)B), dim3(64), 0, stream, mol_ptr, tuple_ptr, (int)B, option, out0,
                     out1);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_radius_graph_count(const float* pos, const int32_t* mol_ptr, int64_t B, int max_n, float r2,
                                         int cap, int32_t* deg, hipStream_t stream) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL((k_radius<0>), dim3((unsigned)B), dim3(64), radius_lds(max_n), stream, pos, mol_ptr, nullptr,
                     (int)B, max_n, r2, cap, deg, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0f, nullptr);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_radius_graph_fill(const float* pos, const int32_t* mol_ptr, int64_t B, int max_n, float r2,
                                        int cap, const int64_t* edge_ptr, int64_t* edge_src, int64_t* edge_dst,
                                        float* edge_weight, hipStream_t stream) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL((k_radius<1>), dim3((unsigned)B), dim3(64), radius_lds(max_n), stream, pos, mol_ptr, nullptr,
                     (int)B, max_n, r2, cap, nullptr, edge_ptr, edge_src, edge_dst, edge_weight, nullptr, nullptr, 1.0f,
                     nullptr);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_pair_geometry(const float* pos, const int32_t* mol_ptr, const int32_t* pair_ptr, int64_t B,
                                    int max_n, float r2, int cap, float cutoff, float* pair_d, float* pair_c,
                                    uint8_t* pair_flag, hipStream_t stream) {
  if (B <= 0) return 0;
  if (max_n <= cap && max_n <= 64) {
    hipLaunchKernelGGL(k_pair_geometry_flat, dim3((unsigned)B), dim3(64), 0, stream, pos, mol_ptr, pair_ptr, (int)B, r2,
                       cutoff, pair_d, pair_c, pair_flag);
    GEOSSL_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL((k_radius<2>), dim3((unsigned)B), dim3(64), radius_lds(max_n), stream, pos, mol_ptr, pair_ptr,
                     (int)B, max_n, r2, cap, nullptr, nullptr, nullptr, nullptr, nullptr, pair_d, pair_flag, cutoff, pair_c);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_super_edge_ptr(const int64_t* batch, const int64_t* sei0, const int64_t* sei1, int64_t S,
                                     int64_t B, int32_t* se_ptr, int64_t* stats, hipStream_t stream) {
  hipLaunchKernelGGL(k_super_edge_ptr, dim3(grid1d(S > B ? S : B + 1, 256)), dim3(256), 0, stream, batch, sei0, sei1,
                     (int)S, (int)B, se_ptr, stats);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_incidence_count(const int64_t* batch, const int64_t* sei0, const int64_t* sei1,
                                      const int32_t* se_ptr, int64_t N, int sides, int32_t* inc_cnt,
                                      hipStream_t stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL((k_incidence<0>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, batch, sei0, sei1, se_ptr,
                     (int)N, sides, inc_cnt, nullptr, nullptr);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_incidence_fill(const int64_t* batch, const int64_t* sei0, const int64_t* sei1,
                                     const int32_t* se_ptr, int64_t N, int sides, const int64_t* inc_ptr,
                                     int32_t* inc_idx, hipStream_t stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL((k_incidence<1>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, batch, sei0, sei1, se_ptr,
                     (int)N, sides, nullptr, inc_ptr, inc_idx);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_pair_distance(const float* pos, const int64_t* sei0, const int64_t* sei1, int64_t S, float* out,
                                    hipStream_t stream) {
  if (S <= 0) return 0;
  hipLaunchKernelGGL(k_pair_distance, dim3(grid1d(S, 256)), dim3(256), 0, stream, pos, sei0, sei1, (int)S, out);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_ddm_views_dyn(const float* pos, const float* noise, const int64_t* sei0, const int64_t* sei1,
                                    int64_t N, int64_t S, float* pos2, float* d01, float* d02, const int64_t* z,
                                    int64_t z_stride, int64_t* z2, const int32_t* dyn_N, const int32_t* dyn_S,
                                    hipStream_t stream) {
  if (N <= 0) return 0;
  const int64_t work = 3 * N > S ? 3 * N : S;
  hipLaunchKernelGGL(k_ddm_views, dim3(grid1d(work, 256)), dim3(256), 0, stream, pos, noise, sei0, sei1, 3 * N, (int)S,
                     pos2, d01, d02, z, z_stride, z2, dyn_N, dyn_S);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_ddm_views(const float* pos, const float* noise, const int64_t* sei0, const int64_t* sei1, int64_t N,
                                int64_t S, float* pos2, float* d01, float* d02, const int64_t* z, int64_t z_stride,
                                int64_t* z2, hipStream_t stream) {
  return geossl_ddm_views_dyn(pos, noise, sei0, sei1, N, S, pos2, d01, d02, z, z_stride, z2, nullptr, nullptr, stream);
}

extern "C" int geossl_copy2(void* dst0, const void* src0, int64_t bytes0, void* dst1, const void* src1, int64_t bytes1,
                            hipStream_t stream) {
  if ((bytes0 | bytes1) & 3) return (int)hipErrorInvalidValue;
  if (((uintptr_t)dst0 | (uintptr_t)src0 | (uintptr_t)dst1 | (uintptr_t)src1) & 3) return (int)hipErrorInvalidValue;
  const int64_t n0 = bytes0 / 4, n1 = bytes1 / 4, work = n0 > n1 ? n0 : n1;
  if (work <= 0) return 0;
  hipLaunchKernelGGL(k_copy2, dim3(grid1d(work, 256)), dim3(256), 0, stream, (uint32_t*)dst0, (const uint32_t*)src0, n0,
                     (uint32_t*)dst1, (const uint32_t*)src1, n1);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

// up to GEOSSL_COPY_MAX device-to-device copies in one launch (blockIdx.y = copy): the per-step refresh of a
// capacity-bucketed graph's inputs (atom types, positions, batch vector, both rows of super_edge_index)
namespace {
__global__ void k_copy_n(GeosslCopyBatch b) {
  const uint32_t* __restrict__ s = reinterpret_cast<const uint32_t*>(b.src[blockIdx.y]);
  uint32_t* __restrict__ d = reinterpret_cast<uint32_t*>(b.dst[blockIdx.y]);
  const int64_t n = b.bytes[blockIdx.y] / 4;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
  if (s == nullptr) {  // a fill with zeros (src = NULL): a gradient buffer cleared in the launch that refreshes the inputs
    for (int64_t i = t0; i < n; i += nt) d[i] = 0u;
    return;
  }
  for (int64_t i = t0; i < n; i += nt) d[i] = s[i];
}
}  // namespace

extern "C" int geossl_copy_n(const GeosslCopyBatch* batch, int n, hipStream_t stream) {
  if (batch == nullptr || n < 0 || n > GEOSSL_COPY_MAX) return (int)hipErrorInvalidValue;
  int64_t work = 0;
  for (int i = 0; i < n; ++i) {
    if ((batch->bytes[i] & 3) || (((uintptr_t)batch->dst[i] | (uintptr_t)batch->src[i]) & 3) || batch->bytes[i] < 0 ||
        (batch->dst[i] == nullptr && batch->bytes[i] > 0))
      return (int)hipErrorInvalidValue;
    if (batch->bytes[i] / 4 > work) work = batch->bytes[i] / 4;
  }
  if (n == 0 || work <= 0) return 0;
  hipLaunchKernelGGL(k_copy_n, dim3(grid1d(work, 256, 512), n), dim3(256), 0, stream, *batch);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

static int ddm_noise_launch(const int64_t* seed, uint64_t seed_value, float mu, float sigma, int64_t n_pos, int64_t S,
                            int64_t B, int K1, int K2, float* pos_noise, int64_t* noise_level_1,
                            float* distance_noise_1, int64_t* noise_level_2, float* distance_noise_2,
                            hipStream_t stream) {
  if (K1 < 1 || K2 < 1) return (int)hipErrorInvalidValue;
  const int64_t total = (n_pos + 3) / 4 + 2 * ((S + 3) / 4) + 2 * ((B + 3) / 4);
  if (total <= 0) return 0;
  hipLaunchKernelGGL(k_ddm_noise, dim3(grid1d(total, 256)), dim3(256), 0, stream, seed, seed_value, mu, sigma, n_pos, S, B,
                     K1, K2, pos_noise, noise_level_1, distance_noise_1, noise_level_2, distance_noise_2);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

extern "C" int geossl_ddm_noise(const int64_t* seed, float mu, float sigma, int64_t n_pos, int64_t S, int64_t B, int K1,
                                int K2, float* pos_noise, int64_t* noise_level_1, float* distance_noise_1,
                                int64_t* noise_level_2, float* distance_noise_2, hipStream_t stream) {
  if (seed == nullptr) return (int)hipErrorInvalidValue;
  return ddm_noise_launch(seed, 0, mu, sigma, n_pos, S, B, K1, K2, pos_noise, noise_level_1, distance_noise_1,
                          noise_level_2, distance_noise_2, stream);
}

extern "C" int geossl_ddm_noise_seeded(uint64_t seed, float mu, float sigma, int64_t n_pos, int64_t S, int64_t B, int K1,
                                       int K2, float* pos_noise, int64_t* noise_level_1, float* distance_noise_1,
                                       int64_t* noise_level_2, float* distance_noise_2, hipStream_t stream) {
  return ddm_noise_launch(nullptr, seed, mu, sigma, n_pos, S, B, K1, K2, pos_noise, noise_level_1, distance_noise_1,
                          noise_level_2, distance_noise_2, stream);
}

extern "C" int geossl_axpy(const float* a, const float* b, float alpha, int64_t n, float* out, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_axpy, dim3(grid1d(n, 256)), dim3(256), 0, stream, a, b, alpha, n, out);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}
