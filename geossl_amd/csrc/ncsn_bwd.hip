// Backward of the denoising-distance-matching head NCSN_version_03 (NCSN.py:168-220) in ONE pass over the super-edge
// rows: the row gradients (ncsn_rows.hip: k_ncsn_bwd_rows computes the same quantities) AND the weight gradients of
// the two dense layers of output_mlp, on the 16-bit matrix pipe as products of two fp16 pieces per operand (split.h:
// three MFMAs per product; operand scales below).  Nothing per-row is re-read: the separate
// weight-gradient launches (wgrad.h with NcsnW1Ops / NcsnW2Ops) fetched a1, a2, dz1 and the gathered h_u + h_v a
// second time from HBM with 4-byte column requests (0.36 ms per step for the two heads against 0.15 ms of row pass).
//
// Per super-edge row s = (u, v), with g = d loss / d out_row (saved gscale times the upstream scalar):
//     dz2 = g w3 [a2 > 0]                       dW2 += dz2^T a1          db2 += sum dz2
//     dz1 = (dz2 o2_w) [a1 > 0]                 dW1 += dz1^T [h_u + h_v, emb]      db1 += sum dz1
//     dfeat = dz1 o1_w[:, :F]                   demb = dz1 . o1_w[:, F]
//
// A block of 2 NW waves (NW = F/32) works on one 32-row tile at a time; every wave owns a 32-wide feature block and
// keeps its weight slice and its weight-gradient accumulators in registers for all tiles of the block:
//   wave A_n (hidden units [32n, 32n+32) of layer 1):
//       dz1 for its units, evaluated with the ROWS on M (dz2 fragments from LDS x its o2_w slice): the result has the
//       unit on the lane and the rows in the registers (kperm order) - exactly the A-fragment layout of dW1's
//       contraction over rows, and a1 is requested in that layout (16 4-byte loads of a column) for the mask;
//       dW1[its units][all features] against the (h_u + h_v)^T fragments the B waves publish; the emb column and db1
//       from the same registers; the pieces of dz1 go through the matrix pipe against a selection matrix (4 MFMAs,
//       exact: an fp16 piece times 1.0) and land with the row on the lane - the B fragments of the next product - and
//       are published; dW2[all units of layer 2][its columns] against its own a1 fragments (phase 2).
//   wave B_k (features [32k, 32k+32) of layer 0's input):
//       dfeat for its features (its o1_w^T slice x the published dz1 fragments, rows on N: 16-byte row-piece stores);
//       builds, one tile ahead: the dz2 fragments (from a2, g, w3), (h_u + h_v)^T for its features (4-byte gathers of
//       atom rows from L2: a half-wave reads 128 contiguous bytes of one atom) and - B_0, B_1 - dz2^T through the
//       selection-matrix product, with db2.
// Two barriers per tile (LDS-only: s_waitcnt lgkmcnt(0) + s_barrier, global requests stay in flight across them):
// phase 1 is MFMA work of the A waves (40 per tile at F = 128) while the B waves build the next tile, phase 2 is
// MFMA work of the B waves (36: dfeat and dW2).  One read of a1 / a2 and no dz1 in HBM at all.
//
// Operand scales (fp16 has five exponent bits; every scale is a power of two, exact, and undone on the fp32 side):
//   * the upstream row gradient g spans orders of magnitude between molecules (sigma^anneal_power, NCSN.py:215-218):
//     wave A_0 keeps EG, the RUNNING exponent of the largest |g| of the block's tiles so far, and publishes it with the
//     row scalars of a tile.  dz2 = g w3 [a2 > 0] is scaled by 2^(14 - EG - e3) (e3: exponent of max |w3|);
//   * dz1 is never measured: |dz1[row][m]| <= |g| sum_k |w3[k] o2_w[k][m]| < 2^(EG + eC) with eC the exponent of the
//     largest such column sum (a weight-only constant, block-uniform), so dz1 in units of 2^(EG + eC - 14) stays
//     below 2^14 and comes out of the dz1 accumulator by ONE constant factor.  A value 2^17 below its bound still
//     has all 22 bits; below that the absolute error is 2^-39 of the bound;
//   * the forward activations (a1^T by the A waves, (h_u + h_v)^T by the B waves) are measured per wave and tile and
//     scaled by running exponents EA / EF, published next to the fragments;
//   * a weight-gradient accumulator holds its sum in units of 2^(EG + E* - 28 + const): when one of the two running
//     exponents rises (rare after the first tiles) the wave multiplies the accumulator by the power of two in between.
// The narrow gradients (output_mlp.layers.2 and the 1 -> F -> 1 distance embedding) are per-unit sums over the rows of
// quantities the tile already holds (g, demb, the perturbed distance): VALU work of the A waves in phase 2, where they
// have no MFMAs; layers.2.weight accumulates in the B waves where they turn a2 into the dz2 fragments.
// One partial per block for every weight gradient, summed in block order by k_reduce_multi (no atomics).
#include "common.h"
#include "geossl_hip.h"
#include "split.h"
#include "tn.h"

using namespace geossl;

namespace {

constexpr int TR = 32;  // rows per tile

#ifdef NB_TIMING
__device__ long long nb_dbg[2 * 32 * 8];
#define NB_MARK(slot)                                                                              \
  do {                                                                                             \
    if (blockIdx.x == 5 && lane == 0 && (wave == 0 || wave == NW) && t - t_begin < 32)             \
      nb_dbg[((wave == 0 ? 0 : 1) * 32 + (t - t_begin)) * 8 + (slot)] = clock64();                 \
  } while (0)
#else
#define NB_MARK(slot) do {} while (0)
#endif

template <int NW>
struct NbLds {
  static constexpr int F = 32 * NW, H = F / 2, KHS = (H + 15) / 16, HMB = (H + 31) / 32, KS = F / 16;
  u32x4* zb;    // [2][KHS][2][64]      dz2 as A fragments (lane = row, 8 consecutive units), double buffered
  u32x4* dz2T;  // [HMB][2][2][64]      dz2^T as A fragments (lane = unit of layer 2, rows in kperm order)
  u32x4* fT;    // [2][NW][2][2][64]    (h_u + h_v)^T as B fragments (lane = feature, rows in kperm order)
  u32x4* dz1r;  // [KS][2][64]          dz1 as B fragments (lane = row, units in kperm order)
  u32x4* abT;   // [NW][2][2][64]       a1^T as B fragments (lane = unit of layer 1, rows in kperm order)
  int4* scal;   // [3][TR]              {u, v, g, emb} of a tile's rows, ring of three tiles
  float* pdv;   // [3][TR]              perturbed distance of the rows, same ring
  float* dep;   // [TR][NW]             demb partial of every A wave
  float* wls;   // [F]                  o1_w[:, F]
  int* eg;      // [4]                  running exponent of |g| with the scalars of a tile (ring of three)
  int* ef;      // [2][NW]              running exponent of a B wave's (h_u + h_v)^T fragments, double buffered
  int* ea;      // [NW]                 running exponent of an A wave's a1^T fragments
  int* ec;      // [NW]                 prologue: exponent of the largest column sum of |w3 o2_w| per A wave
  __device__ explicit NbLds(uint8_t* smem) {
    zb = reinterpret_cast<u32x4*>(smem);
    dz2T = zb + 2 * KHS * 2 * 64;
    fT = dz2T + HMB * 2 * 2 * 64;
    dz1r = fT + 2 * NW * 2 * 2 * 64;
    abT = dz1r + KS * 2 * 64;
    scal = reinterpret_cast<int4*>(abT + NW * 2 * 2 * 64);
    pdv = reinterpret_cast<float*>(scal + 3 * TR);
    dep = pdv + 3 * TR;
    wls = dep + NW * TR;
    eg = reinterpret_cast<int*>(wls + F);
    ef = eg + 4;
    ea = ef + 2 * NW;
    ec = ea + NW;
  }
  static size_t bytes() {
    return (size_t)(2 * KHS * 2 + HMB * 4 + 2 * NW * 4 + KS * 2 + NW * 4) * 1024 + 3 * TR * sizeof(int4) +
           (size_t)(3 * TR + NW * TR + F) * sizeof(float) + (size_t)(4 + 4 * NW) * sizeof(int);
  }
};

struct NcsnFusedArgs {
  const float* h;
  const int64_t* sei0;
  const int64_t* sei1;
  int S;
  GeosslNcsnWeights w;
  GeosslNcsnSaved sv;
  const int64_t* divisor;
  float out_scale;
  const float* gout;
  float* dfeat;
  float* demb;
  float* grow;
  float* pw1;  // [nblk][F][F]
  float* pd1;  // [nblk][F]      emb column of o1_w
  float* pb1;  // [nblk][F]
  float* pw2;  // [nblk][H][F]
  float* pb2;  // [nblk][H]
  float* psm;  // [nblk][H + 3F + 2]   o3_w, in_w2, in_w1, in_b1, o3_b, in_b2 (layout of k_ncsn_small_partial, ddm.hip)
};

// base (uniform) + 32-bit byte offset: the load takes its base from scalar registers, no 64-bit address per request
__device__ __forceinline__ float ldg_off(const float* base, uint32_t byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
// row of C-layout register r for a lane of half kh, with 4*kh passed in: inside the tile loop it is an opaque copy
// (fresh per iteration), so the sixteen row indices and everything derived from them are recomputed where they are
// used - one VALU instruction each - instead of being hoisted out of the loop and held in (spilled) registers
__device__ __forceinline__ int crow4(int r, int k4) { return (r & 3) + 8 * (r >> 2) + k4; }
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// eight values that ARE fp16 numbers (a piece that went through the matrix pipe against 1.0) back into a fragment
__device__ __forceinline__ u32x4 pack8(const f32x16& t, int s) {
  u32x4 w;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f16x2 p = {(_Float16)t[8 * s + 2 * q], (_Float16)t[8 * s + 2 * q + 1]};
    w[q] = __builtin_bit_cast(uint32_t, p);
  }
  return w;
}
__device__ __forceinline__ const u32x4& piece(const Frag2& f, int pc) { return pc == 0 ? f.h : f.l; }
// exponent of a magnitude for the operand scales, floored: 2^(14 - e) and the products of two such scales stay finite
// for all-zero operands (an untrained bias, a tile past the block's range)
__device__ __forceinline__ int scale_exponent(float m) { return max(mag_exponent(m), -40); }
__device__ __forceinline__ float pow2(int e) { return __builtin_amdgcn_ldexpf(1.0f, e); }
constexpr int E_NONE = -100000;  // "no tile yet": the first rescale multiplies a zero accumulator by 2^-inf = 0

template <int NW, bool ROLE_A>
__device__ __forceinline__ void ncsn_bwd_body(const NcsnFusedArgs& a, const int S, const float* __restrict__ a_h) {
  using L_t = NbLds<NW>;
  constexpr int F = L_t::F, H = L_t::H, KHS = L_t::KHS, HMB = L_t::HMB, KS = L_t::KS, NT = 128 * NW;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  const L_t L(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = ROLE_A ? wave : wave - NW;  // this wave's 32-wide feature block
  const int ntiles = (S + TR - 1) / TR;
  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per, t_end = min(ntiles, t_begin + per);
  const float scale = a.out_scale * (a.gout != nullptr ? a.gout[0] : 1.0f) / (float)a.divisor[0];
  const uint32_t col = 32 * nb + j;  // this lane's feature

  // ---------------------------------------------------------------- per-role constant state
  Frag2 wreg[ROLE_A ? KHS : KS];  // A: o2_w[:, col] as B fragments (k = unit of layer 2, natural order), scale 2^(14-e2)
                                   // B: o1_w[:, col] as A fragments (k = unit of layer 1, kperm order), scale 2^(14-e1)
  f32x16 accw1[ROLE_A ? NW : 1];   // A: dW1[32nb + reg][32kb + lane] in units of 2^(E1[kb] + eC - 28)
  f32x16 accw2[ROLE_A ? 1 : HMB];  // B: dW2[32mb + reg][32nb + lane] in units of 2^(E2 + e3 - 28)
  int E1[ROLE_A ? NW : 1];         // A: EG + EF[kb] the accumulator is held at
  int E2 = E_NONE;                 // B: EG + EA[nb]
  int ew = 0;                      // A: e2, B: e1 (exponent of the wave's weight slice)
  float bsum = 0.0f, dsum = 0.0f;  // A: db1 / emb column of its unit;  B: db2 of its unit
  u32x4 ident[2];                  // selection matrices of the matrix-pipe transpositions (B operand, fp16 ones)
  float w3r[8];
  // A: the narrow gradients of this lane's unit of the 1 -> F -> 1 distance embedding (NCSN.py:197); wave A_0 also sums
  // g and demb over the rows (layers.2.bias, input_distance_mlp bias 2)
  float s_w2 = 0.0f, s_w1 = 0.0f, s_b1 = 0.0f, s_g = 0.0f, s_d = 0.0f;
  float iw1 = 0.0f, ib1 = 0.0f, iw2 = 0.0f;
  // e3: exponent of the largest |layers.2.weight| (every wave evaluates it for itself)
  int e3;
  {
    float m3 = 0.0f;
    for (int i = lane; i < H; i += 64) m3 = fmaxf(m3, fabsf(a.w.o3_w[i]));
    e3 = scale_exponent(wave_max(m3));
  }
  if constexpr (ROLE_A) {
    iw1 = a.w.in_w1[col];
    ib1 = a.w.in_b1[col];
    iw2 = a.w.in_w2[col];
    float raw[KHS][8];
    float wm = 0.0f, csum = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KHS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int m = 16 * ks + 8 * kh + e;
        const float v = m < H ? a.w.o2_w[(size_t)m * F + col] : 0.0f;
        raw[ks][e] = v;
        wm = fmaxf(wm, fabsf(v));
        csum = fmaf(fabsf(v), m < H ? fabsf(a.w.o3_w[m]) : 0.0f, csum);
      }
    ew = scale_exponent(wave_max(wm));
    const float sw = pow2(14 - ew);
#pragma unroll
    for (int ks = 0; ks < KHS; ++ks) wreg[ks] = split8h_scaled(raw[ks], sw);
    csum += __shfl_xor(csum, 32, 64);  // the two halves of a column
    const int ecw = max(mag_exponent(wave_max(csum)), -80);
    if (lane == 0) L.ec[nb] = ecw;
#pragma unroll
    for (int kb = 0; kb < NW; ++kb) {
      E1[kb] = E_NONE;
#pragma unroll
      for (int r = 0; r < 16; ++r) accw1[kb][r] = 0.0f;
    }
    // dz1^T (lane = unit, k = row 16s + kperm(e, kh)) -> row on the lane: B[k][n = row'] = [16s + kperm(e, kh) == row']
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k0 = 16 * s + kperm(2 * q, kh), k1 = 16 * s + kperm(2 * q + 1, kh);
        ident[s][q] = (k0 == j ? 0x3C00u : 0u) | (k1 == j ? 0x3C000000u : 0u);
      }
  } else {
    float raw[KS][8];
    float wm = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        raw[ks][e] = a.w.o1_w[(size_t)(16 * ks + kperm(e, kh)) * (F + 1) + col];
        wm = fmaxf(wm, fabsf(raw[ks][e]));
      }
    ew = scale_exponent(wave_max(wm));
    const float sw = pow2(14 - ew);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wreg[ks] = split8h_scaled(raw[ks], sw);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = 16 * nb + 8 * kh + e;  // k-step nb of the dz2 fragments is built by this wave
      w3r[e] = (nb < KHS && m < H) ? a.w.o3_w[m] : 0.0f;
    }
#pragma unroll
    for (int mb = 0; mb < HMB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) accw2[mb][r] = 0.0f;
    // dz2 (lane = row, k = unit 16q' + 8kh + e, natural order) -> unit on the lane: B[k][n] = [16q + 8kh + e == n]
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k0 = 16 * q2 + 8 * kh + 2 * q;
        ident[q2][q] = (k0 == j ? 0x3C00u : 0u) | (k0 + 1 == j ? 0x3C000000u : 0u);
      }
  }
  for (int i = tid; i < F; i += NT) L.wls[i] = a.w.o1_w[(size_t)i * (F + 1) + F];

  // ---------------------------------------------------------------- request / build steps (see the header)
  // A wave 0, lanes 0..31: the row scalars of tile tt; EG: running exponent of |g| over the tiles published so far
  int sc_u = 0, sc_v = 0;
  float sc_g = 0.0f, sc_e = 0.0f, sc_p = 0.0f;
  int EG = -60;
  auto load_scal = [&](int tt) {
    if constexpr (ROLE_A) {
      if (wave == 0) {
        const int row = min(TR * tt + j, S - 1);
        sc_u = (int)a.sei0[row];
        sc_v = (int)a.sei1[row];
        sc_g = a.sv.gscale[row];
        sc_e = a.sv.emb[row];
        sc_p = a.sv.pd[row];
      }
    }
  };
  auto put_scal = [&](int tt) {
    if constexpr (ROLE_A) {
      if (wave == 0) {
        const int row = TR * tt + lane;
        const bool valid = lane < TR && row < S && tt < t_end;
        const float gr = valid ? pin(sc_g) * scale : 0.0f;
        EG = max(EG, mag_exponent(wave_max(fabsf(gr))));
        if (lane < TR) {
          L.scal[(tt % 3) * TR + lane] = make_int4(sc_u, sc_v, __float_as_int(gr), __float_as_int(sc_e));
          L.pdv[(tt % 3) * TR + lane] = sc_p;
          if (valid) a.grow[row] = gr;
        }
        if (lane == 0) L.eg[tt % 3] = EG;
      }
    }
  };
  // A: a1 of the tile in C-layout order (register r <-> row c_row(r)), this lane's unit
  float a1raw[16];
  auto request_a1 = [&](int tt, int k4) {
    if constexpr (ROLE_A) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        a1raw[r] = ldg_off(a.sv.a1, ((uint32_t)min(TR * tt + crow4(r, k4), S - 1) * (uint32_t)F + col) * 4u);
    }
  };
  // B: a2 of this lane's row, units 16nb + 8kh .. +8 (k-step nb of the dz2 fragments)
  f32x4 a2raw[2];
  float o3acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  auto request_a2 = [&](int tt) {
    if constexpr (!ROLE_A) {
      if (nb < KHS) {
        const uint32_t off = ((uint32_t)min(TR * tt + j, S - 1) * (uint32_t)H + (16 * nb + 8 * kh)) * 4u;
        a2raw[0] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.sv.a2) + off);
        a2raw[1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.sv.a2) + 16 + off);
      }
    }
  };
  // B: h[u][col], h[v][col] for the 16 rows of this lane (C-layout order)
  float hu[16], hv[16];
  auto request_gather = [&](int tt, int k4) {
    if constexpr (!ROLE_A) {
      const int2* sc = reinterpret_cast<const int2*>(L.scal + (tt % 3) * TR);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int2 uv = sc[2 * crow4(r, k4)];  // .x, .y of the int4 entry
        hu[r] = ldg_off(a_h, ((uint32_t)uv.x * (uint32_t)F + col) * 4u);
        hv[r] = ldg_off(a_h, ((uint32_t)uv.y * (uint32_t)F + col) * 4u);
      }
    }
  };
  auto build_zb = [&](int tt) {
    if constexpr (!ROLE_A) {
      if (nb < KHS) {
        const float gr = __int_as_float(L.scal[(tt % 3) * TR + j].z);
        const float sz = pow2(14 - e3 - __builtin_amdgcn_readfirstlane(L.eg[tt % 3]));
        const float av[8] = {a2raw[0].x, a2raw[0].y, a2raw[0].z, a2raw[0].w, a2raw[1].x, a2raw[1].y, a2raw[1].z, a2raw[1].w};
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a2v = pin(av[e]);
          v[e] = a2v > 0.0f ? gr * w3r[e] : 0.0f;
          o3acc[e] = fmaf(gr, a2v, o3acc[e]);  // layers.2.weight gradient of unit 16nb + 8kh + e, this lane's rows
        }
        const Frag2 f = split8h_scaled(v, sz);
        u32x4* dst = L.zb + (size_t)(((tt & 1) * KHS + nb) * 2) * 64 + lane;
        dst[0] = f.h;
        dst[64] = f.l;
      }
    }
  };
  int EF = -40;  // B: running exponent of this wave's (h_u + h_v)^T fragments
  auto build_fT = [&](int tt) {
    if constexpr (!ROLE_A) {
      float v[16];
      float m = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] = pin(hu[r]) + pin(hv[r]);  // NCSN.py:201-203
        m = fmaxf(m, fabsf(v[r]));
      }
      EF = max(EF, mag_exponent(wave_max(m)));
      const float sf = pow2(14 - EF);
      if (lane == 0) L.ef[(tt & 1) * NW + nb] = EF;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = v[8 * s + e];
        const Frag2 f = split8h_scaled(vv, sf);
        u32x4* dst = L.fT + (size_t)((((tt & 1) * NW + nb) * 2 + s) * 2) * 64 + lane;
        dst[0] = f.h;
        dst[64] = f.l;
      }
    }
  };

  // ---------------------------------------------------------------- prologue
  load_scal(t_begin);
  put_scal(t_begin);
  load_scal(t_begin + 1);
  put_scal(t_begin + 1);
  load_scal(t_begin + 2);  // written in phase 1 of the first tile
  if (t_begin < t_end) {
    request_a1(t_begin, 4 * kh);
    request_a2(t_begin);
  }
  __syncthreads();
  int eC = -80;  // block-uniform exponent of the largest column sum of |w3 o2_w|: |dz1| < 2^(EG + eC)
#pragma unroll
  for (int i = 0; i < NW; ++i) eC = max(eC, __builtin_amdgcn_readfirstlane(L.ec[i]));
  if (t_begin < t_end) {
    request_gather(t_begin, 4 * kh);
    build_zb(t_begin);
    build_fT(t_begin);
    request_a2(t_begin + 1);
    request_gather(t_begin + 1, 4 * kh);
  }
  int EA = -40;  // A: running exponent of this wave's a1^T fragments
  const float kA = pow2(e3 + ew - eC - 14);  // A: dz1 accumulator -> dz1 in units of 2^(EG + eC - 14)
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = t & 1;
    int k4 = opaque(4 * kh);
    NB_MARK(0);
    lds_barrier();  // X(t): fragments and scalars of tile t published; everything of tile t-1 consumed
    // =============================================================== phase 1
    NB_MARK(1);
    const int EGt = __builtin_amdgcn_readfirstlane(L.eg[t % 3]);
    if constexpr (ROLE_A) {
      put_scal(t + 2);
      const float kT = pow2(EGt + eC - 14);  // one unit of this tile's scaled dz1
      // ---- dz1^T for this wave's units: rows on M
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
      {
        const u32x4* zs = L.zb + (size_t)(buf * KHS * 2) * 64 + lane;
        Frag2 a0, an;
        a0.h = zs[0]; a0.l = zs[64];
#pragma unroll
        for (int ks = 0; ks < KHS; ++ks) {
          if (ks + 1 < KHS) {
            const u32x4* s0 = zs + (size_t)((ks + 1) * 2) * 64;
            an.h = s0[0]; an.l = s0[64];
          }
          __builtin_amdgcn_sched_barrier(0);
          mma3x2(acc0, acc1, a0, wreg[ks]);
          __builtin_amdgcn_sched_barrier(0);
          if (ks + 1 < KHS) a0 = an;
        }
      }
      NB_MARK(5);
      float v[16];
      {
        const int4* sc = L.scal + (t % 3) * TR;
        float tb = 0.0f, td = 0.0f, am = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          a1raw[r] = pin(a1raw[r]);
          v[r] = a1raw[r] > 0.0f ? (acc0[r] + acc1[r]) * kA : 0.0f;  // rows past S: dz2 = 0 there
          tb += v[r];
          td = fmaf(v[r], __int_as_float(sc[crow4(r, k4)].w), td);
          am = fmaxf(am, fabsf(a1raw[r]));
        }
        bsum = fmaf(tb, kT, bsum);
        dsum = fmaf(td, kT, dsum);
        EA = max(EA, mag_exponent(wave_max(am)));
      }
      // a1^T of this wave's units as B fragments of dW2's contraction over rows (multiplied by the B wave that owns
      // these columns of dW2); the registers are then free for the next tile's request
      {
        const float sa = pow2(14 - EA);
        if (lane == 0) L.ea[nb] = EA;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float vv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] = a1raw[8 * s + e];
          const Frag2 f = split8h_scaled(vv, sa);
          u32x4* dst = L.abT + (size_t)((nb * 2 + s) * 2) * 64 + lane;
          dst[0] = f.h;
          dst[64] = f.l;
        }
      }
      Frag2 da[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = v[8 * s + e];
        da[s] = split8h(vv);
      }
      // ---- the pieces of dz1 with the row on the lane (exact), published as B fragments; demb partial
      {
        float wl[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) wl[r] = L.wls[32 * nb + crow4(r, k4)];
        float de = 0.0f;
#pragma unroll
        for (int pc = 1; pc >= 0; --pc) {  // smallest piece first
          f32x16 tp;
#pragma unroll
          for (int r = 0; r < 16; ++r) tp[r] = 0.0f;
          tp = mfma_f16(piece(da[0], pc), ident[0], tp);
          tp = mfma_f16(piece(da[1], pc), ident[1], tp);
#pragma unroll
          for (int r = 0; r < 16; ++r) de = fmaf(tp[r], wl[r], de);
#pragma unroll
          for (int half = 0; half < 2; ++half)
            L.dz1r[(size_t)((2 * nb + half) * 2 + pc) * 64 + lane] = pack8(tp, half);
        }
        de += __shfl_xor(de, 32, 64);
        if (kh == 0) L.dep[j * NW + nb] = de * kT;
      }
      NB_MARK(6);
      // ---- dW1[this wave's units][all features] += dz1^T (h_u + h_v)
#pragma unroll
      for (int kb = 0; kb < NW; ++kb) {
        const int En = EGt + __builtin_amdgcn_readfirstlane(L.ef[buf * NW + kb]);
        if (En != E1[kb]) {  // a running exponent rose: the sum so far in the new unit
          const float f = pow2(max(E1[kb] - En, -200));
#pragma unroll
          for (int r = 0; r < 16; ++r) accw1[kb][r] *= f;
          E1[kb] = En;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const u32x4* s0 = L.fT + (size_t)(((buf * NW + kb) * 2 + s) * 2) * 64 + lane;
          Frag2 bf;
          bf.h = s0[0]; bf.l = s0[64];
          mma3(accw1[kb], da[s], bf);
        }
      }
    } else {
      // ---- dz2^T for unit block nb of layer 2 through the selection matrix; db2
      if (nb < HMB) {
        float tb = 0.0f;
#pragma unroll
        for (int pc = 1; pc >= 0; --pc) {
          f32x16 tp;
#pragma unroll
          for (int r = 0; r < 16; ++r) tp[r] = 0.0f;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (2 * nb + q < KHS) {
              const u32x4* s0 = L.zb + (size_t)((buf * KHS + 2 * nb + q) * 2 + pc) * 64 + lane;
              tp = mfma_f16(s0[0], ident[q], tp);
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) tb += tp[r];
#pragma unroll
          for (int s = 0; s < 2; ++s) L.dz2T[(size_t)((nb * 2 + s) * 2 + pc) * 64 + lane] = pack8(tp, s);
        }
        bsum = fmaf(tb, pow2(EGt + e3 - 14), bsum);
      }
      NB_MARK(5);
      // ---- next tile's fragments (their requests have been in flight since phase 2 of the previous tile)
      build_zb(t + 1);
      NB_MARK(6);
      build_fT(t + 1);
    }
    k4 = opaque(4 * kh);
    NB_MARK(2);
    lds_barrier();  // Y(t): dz1 fragments, dz2^T fragments, demb partials, scalars of tile t+2 published
    NB_MARK(3);
    // =============================================================== phase 2
    if constexpr (ROLE_A) {
      // this phase has no matrix work for the A waves: their requests for the next tile go out here
      request_a1(t + 1, k4);
      load_scal(t + 3);
      // ---- narrow gradients: this lane's unit, rows 16kh .. 16kh+15
      {
        const int4* sc = L.scal + (t % 3) * TR;
        const float* pdr = L.pdv + (t % 3) * TR;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int rl = 4 * k4 + i;
          float de = 0.0f;
#pragma unroll
          for (int n2 = 0; n2 < NW; ++n2) de += L.dep[rl * NW + n2];  // 0 for rows past S
          const float pd = pdr[rl];
          const float pre = fmaf(iw1, pd, ib1);
          s_w2 = fmaf(de, fmaxf(pre, 0.0f), s_w2);
          const float dp = pre > 0.0f ? de * iw2 : 0.0f;
          s_w1 = fmaf(dp, pd, s_w1);
          s_b1 += dp;
          if (wave == 0 && j == i) {  // lanes (i, kh) of wave A_0: one per row
            s_g += __int_as_float(sc[rl].z);
            s_d += de;
          }
        }
      }
    } else {
      request_a2(t + 2);
      request_gather(t + 2, k4);
      NB_MARK(7);
      // ---- dfeat^T for this wave's features: rows on N
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
      {
        const u32x4* zs = L.dz1r + lane;
        Frag2 b0, bn;
        b0.h = zs[0]; b0.l = zs[64];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (ks + 1 < KS) {
            const u32x4* s0 = zs + (size_t)((ks + 1) * 2) * 64;
            bn.h = s0[0]; bn.l = s0[64];
          }
          __builtin_amdgcn_sched_barrier(0);
          mma3x2(acc0, acc1, wreg[ks], b0);
          __builtin_amdgcn_sched_barrier(0);
          if (ks + 1 < KS) b0 = bn;
        }
      }
      const float kD = pow2(EGt + eC + ew - 28);
      f32x16 outp;
#pragma unroll
      for (int r = 0; r < 16; ++r) outp[r] = (acc0[r] + acc1[r]) * kD;
      const int row = TR * t + j;
      if (row < S) {
        const uint32_t off = ((uint32_t)row * (uint32_t)F + 32 * nb + 4 * kh) * 4u;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.dfeat) + 32 * q + off) =
              f32x4{outp[4 * q], outp[4 * q + 1], outp[4 * q + 2], outp[4 * q + 3]};
        if (nb == 0 && kh == 0) {
          float de = 0.0f;
#pragma unroll
          for (int n2 = 0; n2 < NW; ++n2) de += L.dep[j * NW + n2];
          a.demb[row] = de;
        }
      }
      // ---- dW2[all units of layer 2][this wave's columns] += dz2^T a1
      {
        const int En = EGt + __builtin_amdgcn_readfirstlane(L.ea[nb]);
        if (En != E2) {
          const float f = pow2(max(E2 - En, -200));
#pragma unroll
          for (int mb = 0; mb < HMB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) accw2[mb][r] *= f;
          E2 = En;
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u32x4* sb = L.abT + (size_t)((nb * 2 + s) * 2) * 64 + lane;
        Frag2 bf;
        bf.h = sb[0]; bf.l = sb[64];
#pragma unroll
        for (int mb = 0; mb < HMB; ++mb) {
          const u32x4* s0 = L.dz2T + (size_t)((mb * 2 + s) * 2) * 64 + lane;
          Frag2 af;
          af.h = s0[0]; af.l = s0[64];
          mma3(accw2[mb], af, bf);
        }
      }
    }
    NB_MARK(4);
  }
  // ---------------------------------------------------------------- one partial per block
  const size_t pb = blockIdx.x;
  if constexpr (ROLE_A) {
    float* P1 = a.pw1 + pb * F * F;
#pragma unroll
    for (int kb = 0; kb < NW; ++kb) {
      const float f = pow2(max(E1[kb] + eC - 28, -200));
#pragma unroll
      for (int r = 0; r < 16; ++r) P1[(size_t)(32 * nb + c_row(r, lane)) * F + 32 * kb + j] = accw1[kb][r] * f;
    }
    const float sb = bsum + __shfl_xor(bsum, 32, 64), sd = dsum + __shfl_xor(dsum, 32, 64);
    if (kh == 0) {
      a.pb1[pb * F + col] = sb;
      a.pd1[pb * F + col] = sd;
    }
    float* Ps = a.psm + pb * (H + 3 * F + 2);
    const float t_w2 = s_w2 + __shfl_xor(s_w2, 32, 64);
    const float t_w1 = s_w1 + __shfl_xor(s_w1, 32, 64), t_b1 = s_b1 + __shfl_xor(s_b1, 32, 64);
    if (kh == 0) {
      Ps[H + col] = t_w2;
      Ps[H + F + col] = t_w1;
      Ps[H + 2 * F + col] = t_b1;
    }
    if (wave == 0) {
      const float tg = wave_sum(s_g), td = wave_sum(s_d);
      if (lane == 0) {
        Ps[H + 3 * F] = tg;
        Ps[H + 3 * F + 1] = td;
      }
    }
  } else {
    float* P2 = a.pw2 + pb * H * F;
    const float f = pow2(max(E2 + e3 - 28, -200));
#pragma unroll
    for (int mb = 0; mb < HMB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * mb + c_row(r, lane);
        if (m < H) P2[(size_t)m * F + col] = accw2[mb][r] * f;
      }
    const float sb = bsum + __shfl_xor(bsum, 32, 64);
    if (nb < HMB && kh == 0 && (int)col < H) a.pb2[pb * H + col] = sb;
    if (nb < KHS) {  // layers.2.weight: sum this half-wave's 32 rows
      float* Ps = a.psm + pb * (H + 3 * F + 2);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = o3acc[e];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        const int m = 16 * nb + 8 * kh + e;
        if (j == 0 && m < H) Ps[m] = v;
      }
    }
  }
}

// The two roles are separate instantiations of the body (different register sets); the branch is wave-uniform and
// both sides execute the same barriers.
template <int NW>
__global__ __launch_bounds__(128 * NW) void k_ncsn_bwd_fused(NcsnFusedArgs a) {
  if ((int)(threadIdx.x >> 6) < NW) ncsn_bwd_body<NW, true>(a, a.S, a.h);
  else ncsn_bwd_body<NW, false>(a, a.S, a.h);
}
// both heads of a DDM step in one launch: blockIdx.y = head
// (the two argument sets as ONE kernel argument indexed by blockIdx.y: a block loads its own set from the kernarg
// segment; as two arguments with a select between them the compiler kept BOTH sets in scalar registers and spilled 64)
struct NcsnFusedPair { NcsnFusedArgs a[2]; };
template <int NW>
__global__ __launch_bounds__(128 * NW) void k_ncsn_bwd_fused2(NcsnFusedPair pr, const int32_t* __restrict__ dyn_S,
                                                              const int32_t* __restrict__ dyn_view) {
  const NcsnFusedArgs& a = pr.a[blockIdx.y];
  // capacity launch (see k_ncsn_fwd2): real row count; head 1's features start *dyn_view rows into the shared tensor
  const int S = dyn_count(a.S, dyn_S);
  const float* h = a.h;
  if (dyn_view != nullptr && blockIdx.y == 1) h += (size_t)(*dyn_view) * (32 * NW);
  if ((int)(threadIdx.x >> 6) < NW) ncsn_bwd_body<NW, true>(a, S, h);
  else ncsn_bwd_body<NW, false>(a, S, h);
}

// The five fixed-order reductions of the block partials (k_reduce_multi's arithmetic) and, in the blocks behind them, the
// narrow gradients (k_ncsn_small_reduce's: one wave per output scalar, lanes stride over the blocks, fixed butterfly) -
// one launch per head instead of two.
__device__ __forceinline__ void ncsn_reduce_all_block(const ReduceMulti& m, int nblk, int accumulate,
                                                      const float* __restrict__ psm, int F, const GeosslNcsnGrads& g,
                                                      float (*red)[64]) {
  const int mb = m.xoff[m.nseg];
  if ((int)blockIdx.x < mb) {
    reduce_multi_block(m, nblk, accumulate, red, 0);
    return;
  }
  const int H = F / 2, len = H + 3 * F + 2;
  const int i = ((int)blockIdx.x - mb) * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= len) return;
  float s = 0.0f;
  for (int b = lane; b < nblk; b += 64) s += psm[(size_t)b * len + i];
  s = wave_sum(s);
  if (lane == 0) {
    float* dst;
    if (i < H) dst = g.o3_w + i;
    else if (i < H + F) dst = g.in_w2 + (i - H);
    else if (i < H + 2 * F) dst = g.in_w1 + (i - H - F);
    else if (i < H + 3 * F) dst = g.in_b1 + (i - H - 2 * F);
    else if (i == H + 3 * F) dst = g.o3_b;
    else dst = g.in_b2;
    *dst = accumulate ? *dst + s : s;
  }
}
__global__ __launch_bounds__(256) void k_ncsn_reduce_all(ReduceMulti m, int nblk, int accumulate,
                                                         const float* __restrict__ psm, int F, GeosslNcsnGrads g) {
  __shared__ float red[4][64];
  ncsn_reduce_all_block(m, nblk, accumulate, psm, F, g, red);
}
// both heads of a two-head launch: blockIdx.y = head
__global__ __launch_bounds__(256) void k_ncsn_reduce_all2(ReduceMulti m0, ReduceMulti m1, int nblk, int accumulate,
                                                          const float* __restrict__ psm0,
                                                          const float* __restrict__ psm1, int F, GeosslNcsnGrads g0,
                                                          GeosslNcsnGrads g1) {
  __shared__ float red[4][64];
  if (blockIdx.y == 0) ncsn_reduce_all_block(m0, nblk, accumulate, psm0, F, g0, red);
  else ncsn_reduce_all_block(m1, nblk, accumulate, psm1, F, g1, red);
}

inline int fused_blocks(int64_t S) {
  const int64_t ntiles = (S + TR - 1) / TR;
  return (int)(ntiles < 256 ? ntiles : 256);  // one block per CU
}

}  // namespace

#ifdef NB_TIMING
extern "C" int geossl_ncsn_bwd_debug_read(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nb_dbg), sizeof(long long) * 2 * 32 * 8);
}
#endif

extern "C" int64_t geossl_ddm_loss_bwd_fused_workspace_floats(int64_t S, int F) {
  const int64_t nb = fused_blocks(S), H = F / 2;
  return nb * ((int64_t)F * F + 2 * F + H * F + H + (H + 3 * F + 2));
}

extern "C" int geossl_ddm_loss_bwd_fused(const float* h, const int64_t* sei0, const int64_t* sei1, int64_t S, int64_t N,
                                         int F, const GeosslNcsnWeights* w, const GeosslNcsnSaved* saved,
                                         const int64_t* stats_divisor, float out_scale, const float* gout, float* dfeat,
                                         float* demb, float* grow, const GeosslNcsnGrads* grads, float* workspace,
                                         int accumulate, hipStream_t stream) {
  if (S <= 0) return 0;
  if (F != 32 && F != 64 && F != 128) return (int)hipErrorInvalidValue;
  // 32-bit byte offsets inside the kernel
  if (S * (int64_t)F * 4 >= ((int64_t)1 << 32) || N * (int64_t)F * 4 >= ((int64_t)1 << 32)) return (int)hipErrorInvalidValue;
  const int nb = fused_blocks(S), H = F / 2;
  NcsnFusedArgs a;
  a.h = h; a.sei0 = sei0; a.sei1 = sei1; a.S = (int)S; a.w = *w; a.sv = *saved; a.divisor = stats_divisor;
  a.out_scale = out_scale; a.gout = gout; a.dfeat = dfeat; a.demb = demb; a.grow = grow;
  a.pw1 = workspace;
  a.pd1 = a.pw1 + (size_t)nb * F * F;
  a.pb1 = a.pd1 + (size_t)nb * F;
  a.pw2 = a.pb1 + (size_t)nb * F;
  a.pb2 = a.pw2 + (size_t)nb * H * F;
  a.psm = a.pb2 + (size_t)nb * H;
#define LAUNCH(NWV)                                                                                      \
  do {                                                                                                   \
    allow_big_lds(&k_ncsn_bwd_fused<NWV>);                                                               \
    hipLaunchKernelGGL((k_ncsn_bwd_fused<NWV>), dim3(nb), dim3(128 * NWV), NbLds<NWV>::bytes(), stream, a); \
  } while (0)
  if (F == 128) LAUNCH(4); else if (F == 64) LAUNCH(2); else LAUNCH(1);
#undef LAUNCH
  GEOSSL_CHECK_LAUNCH();
  ReduceMulti rm;  // the five fixed-order partial sums in one launch
  float* o1w[1] = {grads->o1_w};
  float* o1c[1] = {grads->o1_w + F};
  float* o1b[1] = {grads->o1_b};
  float* o2w[1] = {grads->o2_w};
  float* o2b[1] = {grads->o2_b};
  rm.add(a.pw1, F * F, F, F + 1, 1, o1w, 1);
  rm.add(a.pd1, F, F, F, F + 1, o1c, 1);
  rm.add(a.pb1, F, F, F, 1, o1b, 1);
  rm.add(a.pw2, H * F, F, F, 1, o2w, 1);
  rm.add(a.pb2, H, H, H, 1, o2b, 1);
  hipLaunchKernelGGL(k_ncsn_reduce_all, dim3(rm.blocks() + (H + 3 * F + 2 + 3) / 4, 1), dim3(256), 0, stream, rm, nb,
                     accumulate, a.psm, F, *grads);
  GEOSSL_CHECK_LAUNCH();
  return 0;
}

namespace geossl {
int launch_incidence_gather2(const float* dfeat0, const float* dfeat1, const int64_t* inc_ptr, const int32_t* inc_idx,
                             int64_t N, int F, float* dh0, float* dh1, hipStream_t stream, const int32_t* dyn_view);
}

extern "C" int geossl_ddm_loss_bwd_fused2(const GeosslNcsnHeadBwd* heads, const int64_t* sei0, const int64_t* sei1,
                                          int64_t S, int64_t N, int F, const int64_t* stats_divisor, const float* gout,
                                          const int64_t* inc_ptr, const int32_t* inc_idx, int accumulate,
                                          hipStream_t stream) {
  return geossl_ddm_loss_bwd_fused2_dyn(heads, sei0, sei1, S, N, F, stats_divisor, gout, inc_ptr, inc_idx, accumulate,
                                        nullptr, nullptr, stream);
}

extern "C" int geossl_ddm_loss_bwd_fused2_dyn(const GeosslNcsnHeadBwd* heads, const int64_t* sei0, const int64_t* sei1,
                                              int64_t S, int64_t N, int F, const int64_t* stats_divisor,
                                              const float* gout, const int64_t* inc_ptr, const int32_t* inc_idx,
                                              int accumulate, const int32_t* dyn_S, const int32_t* dyn_view,
                                              hipStream_t stream) {
  if (S <= 0) return 0;
  if (heads == nullptr || (F != 32 && F != 64 && F != 128)) return (int)hipErrorInvalidValue;
  if (S * (int64_t)F * 4 >= ((int64_t)1 << 32) || N * (int64_t)F * 4 >= ((int64_t)1 << 32)) return (int)hipErrorInvalidValue;
  // half of the chip per head: the two heads' blocks are resident together (one block of 8 x 256 registers per CU), and
  // a block's fixed costs - formatting its weight slices, 100 KB of partial sums - are paid by half as many blocks
  const int nb = (fused_blocks(S) + 1) / 2, H = F / 2;
  NcsnFusedPair pr;
  NcsnFusedArgs* a = pr.a;
  for (int k = 0; k < 2; ++k) {
    const GeosslNcsnHeadBwd& hd = heads[k];
    if (hd.h == nullptr || hd.dfeat == nullptr || hd.workspace == nullptr) return (int)hipErrorInvalidValue;
    NcsnFusedArgs& x = a[k];
    x.h = hd.h; x.sei0 = sei0; x.sei1 = sei1; x.S = (int)S; x.w = hd.w; x.sv = hd.saved; x.divisor = stats_divisor;
    x.out_scale = hd.out_scale; x.gout = gout; x.dfeat = hd.dfeat; x.demb = hd.demb; x.grow = hd.grow;
    x.pw1 = hd.workspace;
    x.pd1 = x.pw1 + (size_t)nb * F * F;
    x.pb1 = x.pd1 + (size_t)nb * F;
    x.pw2 = x.pb1 + (size_t)nb * F;
    x.pb2 = x.pw2 + (size_t)nb * H * F;
    x.psm = x.pb2 + (size_t)nb * H;
  }
#define LAUNCH2(NWV)                                                                                                \
  do {                                                                                                              \
    allow_big_lds(&k_ncsn_bwd_fused2<NWV>);                                                                         \
    hipLaunchKernelGGL((k_ncsn_bwd_fused2<NWV>), dim3(nb, 2), dim3(128 * NWV), NbLds<NWV>::bytes(), stream, pr,         \
                       dyn_S, dyn_view);                                                                            \
  } while (0)
  if (F == 128) LAUNCH2(4); else if (F == 64) LAUNCH2(2); else LAUNCH2(1);
#undef LAUNCH2
  GEOSSL_CHECK_LAUNCH();
  ReduceMulti rm[2];  // the fixed-order reductions of both heads in one launch
  for (int k = 0; k < 2; ++k) {
    const GeosslNcsnGrads* grads = &heads[k].grads;
    float* o1w[1] = {grads->o1_w};
    float* o1c[1] = {grads->o1_w + F};
    float* o1b[1] = {grads->o1_b};
    float* o2w[1] = {grads->o2_w};
    float* o2b[1] = {grads->o2_b};
    rm[k].add(a[k].pw1, F * F, F, F + 1, 1, o1w, 1);
    rm[k].add(a[k].pd1, F, F, F, F + 1, o1c, 1);
    rm[k].add(a[k].pb1, F, F, F, 1, o1b, 1);
    rm[k].add(a[k].pw2, H * F, F, F, 1, o2w, 1);
    rm[k].add(a[k].pb2, H, H, H, 1, o2b, 1);
  }
  hipLaunchKernelGGL(k_ncsn_reduce_all2, dim3(rm[0].blocks() + (H + 3 * F + 2 + 3) / 4, 2), dim3(256), 0, stream, rm[0],
                     rm[1], nb, accumulate, a[0].psm, a[1].psm, F, heads[0].grads, heads[1].grads);
  GEOSSL_CHECK_LAUNCH();
  if (heads[0].dh != nullptr && heads[1].dh != nullptr)
    return launch_incidence_gather2(heads[0].dfeat, heads[1].dfeat, inc_ptr, inc_idx, N, F, heads[0].dh, heads[1].dh, stream,
                                    dyn_view);
  return 0;
}
