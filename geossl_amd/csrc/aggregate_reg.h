// The register form of the neighbour aggregation (schnet.py:190,194-195: propagate(aggr="add") over the pair-slot layout)
// as a device function: one wave walks one molecule.  Shared by the aggregation kernels (schnet.hip) and the layer loop
// (chain.hip: k_layer_loop).
#pragma once
#include "common.h"

namespace geossl {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------- K4, register form
// The same walk with the molecule's rows of x and the accumulators in REGISTERS: the loops over the row atom a and its
// partners b are unrolled for a size class NMAX >= n, so every row is a named register, there is no LDS, no barrier and
// no address arithmetic beyond one scalar multiply-add per filter row (~12 instructions per pair slot instead of ~44).
// A wave's walk is bound by instruction issue, not by bytes, so this shortens the serial walk of a large molecule -
// which bounds a launch over ragged molecules from below - several times.  Positions (a, b) with b >= n are executed
// on a clamped filter row with their flag bits clear (the flags of a row atom are ballot masks indexed by b): no
// branches, so the filter-row requests run RING positions ahead of their use in one basic block.  Same summation
// order as k_aggregate (separate multiply and add, ascending source per target): results are bit-identical.
template <int NMAX, int RING_ = 0>
__device__ __forceinline__ void aggregate_reg_body(const float* __restrict__ x, const float* __restrict__ Wf,
                                                   const uint8_t* __restrict__ pair_flag, int a0, int n, int base,
                                                   int lane, int f, int F, int swap, float* __restrict__ out) {
#pragma clang fp contract(off)
  typedef f32x2 V;
  // filter rows in flight per wave: a wave's walk takes (positions / RING) memory round trips (~2 us each under load),
  // so the large classes - whose single wave bounds a ragged launch from below - get the deeper ring
#ifndef AGG_RING_SMALL
#define AGG_RING_SMALL 16
#endif
  constexpr int RING = RING_ > 0 ? RING_ : (NMAX > 26 ? 24 : (NMAX > 20 ? 20 : (NMAX > 18 ? 12 : AGG_RING_SMALL)));
  constexpr int NPOS = NMAX * (NMAX - 1) / 2;            // positions of the unrolled walk
  // filter row of a slot = uniform base (scalar registers) + this lane's fixed column offset: the requests then use
  // the scalar-base addressing form and no per-request vector address is ever computed (or kept alive)
  const float* __restrict__ wbase = Wf + (size_t)base * F;
  // slot of position (a, b): a*n - a(a+1)/2 + b - a - 1; invalid positions read slot 0 of the molecule (n >= 2) or, for
  // a one-atom molecule, nothing at all
  const bool has_cols = f >= 0;  // a lane without channels (F = 64: lanes 32..63) only serves the flag ballots
  f = has_cols ? f : 0;
  if (n < 2) {
    if (n == 1 && has_cols) *reinterpret_cast<V*>(out + (size_t)a0 * F + f) = V(0.0f);
    return;
  }
  V xr[NMAX], acc[NMAX], ring[RING];
#pragma unroll
  for (int i = 0; i < NMAX; ++i) {
    xr[i] = *reinterpret_cast<const V*>(x + (size_t)(a0 + min(i, n - 1)) * F + f);
    acc[i] = V(0.0f);
  }
  // Filter row of position (a, b) of the request stream: slot = (first slot of row a) + b - a - 1, the row's first slot
  // carried along in a scalar register and advanced when the stream moves to the next row.  The opaque statement keeps
  // this arithmetic AT the request: left free, the compiler evaluated the closed form a*n - a(a+1)/2 + ... of all
  // NMAX(NMAX-1)/2 positions ahead of the walk and spilled them (134 scalar spills at NMAX = 18, thousands in the kernel
  // that holds every size class).
  const int nu = __builtin_amdgcn_readfirstlane(n);
  int rs_req = 0;
  auto load_row = [&](int a, int b) {
    asm volatile("" : "+s"(rs_req));
    const int slot = b < nu ? rs_req + (b - a - 1) : 0;
    const float* rowp = wbase + (size_t)slot * F;  // uniform
    return __builtin_nontemporal_load(reinterpret_cast<const V*>(rowp + f));
  };
  auto next_pos = [&](int& a, int& b) {  // advance the request stream by one position
    if (++b == NMAX) {
      rs_req += nu - a - 1;
      ++a;
      b = a + 1;
    }
  };
  // Flags of every row atom, requested before anything else: lane b holds the flag byte of slot (a, b).  A flag load
  // inside the walk is the YOUNGEST request of the wave when its ballot needs it, and requests complete in order: it
  // drained the whole ring of filter rows once per row atom (17 memory round trips per 18-atom molecule - the launch
  // was latency bound at 4.6 TB/s where the same stream of filter rows alone reaches 6.5 TB/s).
  unsigned flr[NMAX - 1];
#pragma unroll
  for (int a = 0; a < NMAX - 1; ++a) {
    const bool mine = lane > a && lane < n;  // (clamped address + select below: a predicated load would become a branch)
    flr[a] = pair_flag[base + (mine ? a * n - a * (a + 1) / 2 + lane - a - 1 : 0)];
  }
  int q = 0, ap = 0, bp = 1;
#pragma unroll
  for (int k = 0; k < RING && k < NPOS; ++k) {  // prologue: the first RING positions; then (ap, bp) = position q + RING
    ring[k] = load_row(ap, bp);
    next_pos(ap, bp);
  }
#pragma unroll
  for (int a = 0; a < NMAX - 1; ++a) {
    // flags of row atom a as two ballot masks over the partner index b (the loaded value pinned: with control flow in
    // the walk the arithmetic is sunk below every later request of the block)
    const bool mine = lane > a && lane < n;
    unsigned fl = flr[a];
    asm volatile("" : "+v"(fl));
    fl = mine ? fl : 0u;
    if (swap) fl = ((fl & 1u) << 1) | ((fl >> 1) & 1u);
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64((fl & 1u) != 0u);  // edge b -> a
    const unsigned long long m1 = __builtin_amdgcn_ballot_w64((fl & 2u) != 0u);  // edge a -> b
    V acc_a = acc[a];
    const V xa = xr[a];
#pragma unroll
    for (int b = a + 1; b < NMAX; ++b, ++q) {
      const V w = ring[q % RING];
      if (q + RING < NPOS) {
        ring[q % RING] = load_row(ap, bp);
        next_pos(ap, bp);
      }
      const V t0 = xr[b] * w;
      const V s0 = acc_a + t0;
      acc_a = ((m0 >> b) & 1ull) ? s0 : acc_a;
      const V t1 = xa * w;
      const V s1 = acc[b] + t1;
      acc[b] = ((m1 >> b) & 1ull) ? s1 : acc[b];
      // Keep the schedule as written.  Left alone, the arithmetic (pure register code, only needed by the final
      // stores) is sunk below every request of the block and the live filter rows spill: the volatile statement
      // pins this position's sums in program order, the memory clobber keeps the next request behind it.
      asm volatile("" : "+v"(acc_a.x), "+v"(acc_a.y), "+v"(acc[b].x), "+v"(acc[b].y) : : "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    acc[a] = acc_a;
  }
#pragma unroll
  for (int i = 0; i < NMAX; ++i)
    if (i < n && has_cols) *reinterpret_cast<V*>(out + (size_t)(a0 + i) * F + f) = acc[i];
}

// ---------------------------------------------------------------------------------------- K4, block form
// ONE molecule (1 .. 33 atoms) aggregated by the FOUR waves of a block (the layer loop over the ragged molecules of a
// small batch, chain.hip: k_layer_loop<0>, where a block owns one molecule and three of its waves would watch the
// fourth walk n(n-1)/2 pair slots): wave w forms the sums of the target atoms a = w, w + 4, ...; a lane owns two
// adjacent feature columns (F = 128).  For a target the filter rows of ALL its partners b != a are requested at once
// (at most 32: one memory round trip per target, the next target's requests issued before this one's sums) and added in
// ascending source order with separate multiply and add - the rounding sequence of the walks above: bit-identical.
// A filter row is read by both of its atoms' waves (through L2).  No size classes, no unrolled walks: the molecule's
// rows of x sit in LDS (`smem`: 33 x 128 floats + 528 flag bytes; the caller's block barrier follows).
// NW: waves of the block that take part (4, or 8 in the wide form of the layer loop: the walk is bound by instruction
// issue - one wave per SIMD issues an instruction every four to five cycles at best - so twice the waves halve it).
template <int NW = 4>
__device__ __forceinline__ void aggregate_block_body(const float* __restrict__ x, const float* __restrict__ Wf,
                                                     const uint8_t* __restrict__ pair_flag, int a0, int n, int base,
                                                     int swap, float* __restrict__ out, uint8_t* smem) {
#pragma clang fp contract(off)
  typedef f32x2 V;
  constexpr int F = 128, NP = 32;  // partners of a target: n - 1 <= 32
  float* xs = reinterpret_cast<float*>(smem);   // [n][F]
  uint8_t* fls = smem + 33 * F * sizeof(float);  // [n(n-1)/2] edge flags (exchanged when swap)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nu = __builtin_amdgcn_readfirstlane(n), np = nu * (nu - 1) / 2;
  for (int i = tid; i < nu * (F / 4); i += 64 * NW) {
    const int row = i >> 5, q = i & 31;
    *reinterpret_cast<f32x4*>(xs + row * F + 4 * q) = *reinterpret_cast<const f32x4*>(x + (size_t)(a0 + row) * F + 4 * q);
  }
  for (int p = tid; p < np; p += 64 * NW) {
    unsigned fl = pair_flag[base + p];
    if (swap) fl = ((fl & 1u) << 1) | ((fl >> 1) & 1u);
    fls[p] = (uint8_t)fl;
  }
  __syncthreads();
  if (nu < 2) {
    // a single atom has no pair slot: pair_ptr[m] may equal P (the molecule is the last one), and a request for "slot 0"
    // would read one row past the [L, P, F] filter tensor - the sum over no partners is written without any request
    if (nu == 1 && wave == 0) *reinterpret_cast<V*>(out + (size_t)a0 * F + 2 * lane) = V(0.0f);
    return;
  }
  const float* __restrict__ wcol = Wf + (size_t)base * F + 2 * lane;
  const float* xl = xs + 2 * lane;
  // per target: lane u < n - 1 describes partner b = u + (u >= a): its pair slot and whether b sends to a
  int slot_l = 0, b_l = 0;
  unsigned long long mask = 0ull;
  auto describe = [&](int a) {
    const int u = lane < nu - 1 ? lane : 0;
    const int b = u + (u >= a ? 1 : 0);
    const int i = min(a, b), j = max(a, b);
    slot_l = nu > 1 ? i * nu - i * (i + 1) / 2 + (j - i - 1) : 0;
    b_l = nu > 1 ? b : 0;
    const unsigned fl = nu > 1 ? (unsigned)fls[slot_l] : 0u;
    const bool sends = lane < nu - 1 && ((a < b ? (fl & 1u) : (fl & 2u)) != 0u);   // pair (i < j): bit 0 = j -> i, bit 1 = i -> j
    mask = __builtin_amdgcn_ballot_w64(sends);
  };
  auto request = [&](V (&w)[NP]) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int s = __builtin_amdgcn_readlane(slot_l, u);  // (lanes past the partner count repeat partner 0: clamped, dropped below)
      w[u] = __builtin_nontemporal_load(reinterpret_cast<const V*>(wcol + (size_t)s * F));
    }
  };
  auto reduce = [&](const V (&w)[NP], int a, unsigned long long m, int bl) {
    V acc = V(0.0f);
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int b = __builtin_amdgcn_readlane(bl, u);
      const V xb = *reinterpret_cast<const V*>(xl + b * F);
      const V t = xb * w[u];
      const V s = acc + t;
      acc = ((m >> u) & 1ull) ? s : acc;
    }
    *reinterpret_cast<V*>(out + (size_t)(a0 + a) * F + 2 * lane) = acc;
  };
  // two targets in flight: the requests of target a + 4 are issued before the sums of target a are formed
  V w0[NP], w1[NP];
  int a = wave;
  if (a < nu) {
    describe(a);
    request(w0);
  }
  while (a < nu) {
    const unsigned long long m0 = mask;
    const int bl0 = b_l;
    const int an = a + NW;
    if (an < nu) {
      describe(an);
      request(w1);
    }
    reduce(w0, a, m0, bl0);
    if (an >= nu) break;
    const unsigned long long m1 = mask;
    const int bl1 = b_l;
    const int an2 = an + NW;
    if (an2 < nu) {
      describe(an2);
      request(w0);
    }
    reduce(w1, an, m1, bl1);
    a = an2;
  }
}

// ---------------------------------------------------------------------------------------- K4, target list form
// Molecules ABOVE the largest size class (34 .. 255 atoms: Molecule3D with hydrogens, where the 32-neighbour cap of
// radius_graph makes the edge flags asymmetric): the sums of the target atoms [t0, t1) of one molecule by ONE wave, no
// size classes, no LDS.  Per target the partners b != a in ascending order, 32 at a time: lane u of a chunk describes
// partner u (its pair slot and whether b sends to a: bit 0 of slot (i < j) = edge j -> i, bit 1 = i -> j), the 32 filter
// rows and the 32 rows of x are requested together (x through L2: a molecule's rows are re-read by its other targets),
// then added with separate multiply and add in ascending source order - the rounding sequence of the walks above and of
// a sequential index_add over the canonical edge list: bit-identical.  A filter row is read by both of its atoms'
// targets (twice per launch instead of once: the price of having no per-class walk).
__device__ __forceinline__ void aggregate_targets(const float* __restrict__ x, const float* __restrict__ Wf,
                                                  const uint8_t* __restrict__ pair_flag, int a0, int n, int base,
                                                  int lane, int f, int F, int swap, float* __restrict__ out, int t0,
                                                  int t1) {
#pragma clang fp contract(off)
  typedef f32x2 V;
  constexpr int NP = 32;
  const bool has_cols = f >= 0;
  f = has_cols ? f : 0;
  const int nu = __builtin_amdgcn_readfirstlane(n);
  if (nu < 2) {
    if (nu == 1 && t0 == 0 && t1 > 0 && has_cols) *reinterpret_cast<V*>(out + (size_t)a0 * F + f) = V(0.0f);
    return;
  }
  const float* __restrict__ wcol = Wf + (size_t)base * F + f;
  const float* __restrict__ xcol = x + (size_t)a0 * F + f;
  const uint8_t* __restrict__ flg = pair_flag + base;
  for (int a = t0; a < t1; ++a) {
    V acc = V(0.0f);
    for (int c0 = 0; c0 < nu - 1; c0 += NP) {
      const int u = min(c0 + (lane & (NP - 1)), nu - 2);     // partner index (clamped: dropped by the mask)
      const int b = u + (u >= a ? 1 : 0);
      const int i = min(a, b), j = max(a, b);
      const int slot = i * nu - i * (i + 1) / 2 + (j - i - 1);
      unsigned fl = flg[slot];
      if (swap) fl = ((fl & 1u) << 1) | ((fl >> 1) & 1u);
      const bool sends = lane < NP && c0 + lane < nu - 1 && ((a < b ? (fl & 1u) : (fl & 2u)) != 0u);
      const unsigned long long m = __builtin_amdgcn_ballot_w64(sends);
      V w[NP], xb[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int sk = __builtin_amdgcn_readlane(slot, k), bk = __builtin_amdgcn_readlane(b, k);
        w[k] = *reinterpret_cast<const V*>(wcol + (size_t)sk * F);
        xb[k] = *reinterpret_cast<const V*>(xcol + (size_t)bk * F);
      }
      // ALL 64 requests of the chunk before the first use: left alone, the scheduler trades registers for occupancy and
      // interleaves the sums with the requests - the generated code had ten requests each followed by a full wait
      // (vmcnt(0)): a dozen dependent round trips per target instead of one
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const V t = xb[k] * w[k];
        const V s2 = acc + t;
        acc = ((m >> k) & 1ull) ? s2 : acc;
      }
    }
    if (has_cols) *reinterpret_cast<V*>(out + (size_t)(a0 + a) * F + f) = acc;
  }
}

// One TARGET GROUP of a large molecule: the walk of aggregate_reg_body restricted to the positions that touch a target
// atom in [G0, G1) = part P of K equal parts of the size class - rows a < G0 only at their partners b in the group
// (edge a -> b), rows a in the group at all their partners.  Every sum of a target is still formed by one wave in
// the order of the full walk (positions are dropped, never reordered): results stay bit-identical.  K waves then share
// a molecule whose single walk would outlast the launch (ragged batches: the 26..33-atom molecules), each with
// 1 - (1 - 1/K)^2 of its filter rows (0.75 for K = 2, 0.44 for K = 4); with fewer accumulators the ring is deeper.
template <int NMAX, int K, int P>
__device__ __forceinline__ void aggregate_reg_part(const float* __restrict__ x, const float* __restrict__ Wf,
                                                   const uint8_t* __restrict__ pair_flag, int a0, int n, int base,
                                                   int lane, int f, int F, int swap, float* __restrict__ out) {
#pragma clang fp contract(off)
  typedef f32x2 V;
  constexpr int G0 = (P * NMAX) / K, G1 = ((P + 1) * NMAX) / K, NG = G1 - G0;
  constexpr int AEND = G1 < NMAX - 1 ? G1 : NMAX - 1;     // rows a in [0, AEND) have positions
  constexpr int RING = 40;
  // positions of row a: partners b in [bs(a), be(a))
#define AGG_BS(a) ((a) < G0 ? G0 : (a) + 1)
#define AGG_BE(a) ((a) < G0 ? G1 : NMAX)
  constexpr int NPOS = G0 * NG + (AEND - G0) * NMAX - ((AEND * (AEND + 1)) / 2 - (G0 * (G0 + 1)) / 2);
  const bool has_cols = f >= 0;  // (as in aggregate_reg_body)
  f = has_cols ? f : 0;
  const float* __restrict__ wbase = Wf + (size_t)base * F;
  V xr[NMAX], acc[NG], ring[RING];
#pragma unroll
  for (int i = 0; i < NMAX; ++i) xr[i] = *reinterpret_cast<const V*>(x + (size_t)(a0 + min(i, n - 1)) * F + f);
#pragma unroll
  for (int i = 0; i < NG; ++i) acc[i] = V(0.0f);
  // (request stream as in aggregate_reg_body: the first slot of the stream's row in a scalar register)
  const int nu = __builtin_amdgcn_readfirstlane(n);
  int rs_req = 0;
  auto load_row = [&](int a, int b) {
    asm volatile("" : "+s"(rs_req));
    const int slot = b < nu ? rs_req + (b - a - 1) : 0;
    const float* rowp = wbase + (size_t)slot * F;  // uniform
    return __builtin_nontemporal_load(reinterpret_cast<const V*>(rowp + f));
  };
  unsigned flr[AEND];  // flags of every row atom of this part, requested first (see aggregate_reg_body)
#pragma unroll
  for (int a = 0; a < AEND; ++a) {
    const bool mine = lane > a && lane < n;
    flr[a] = pair_flag[base + (mine ? a * n - a * (a + 1) / 2 + lane - a - 1 : 0)];
  }
  auto next_pos = [&](int& a, int& b) {
    if (++b == AGG_BE(a)) {
      rs_req += nu - a - 1;
      ++a;
      b = AGG_BS(a);
    }
  };
  int q = 0, ap = 0, bp = AGG_BS(0);
#pragma unroll
  for (int k = 0; k < RING && k < NPOS; ++k) {  // prologue: the first RING positions; then (ap, bp) = position q + RING
    ring[k] = load_row(ap, bp);
    next_pos(ap, bp);
  }
  // flags of row atom a as two ballot masks over the partner index b (as in aggregate_reg_body)
  auto row_flags = [&](int a, unsigned long long& m0, unsigned long long& m1) {
    const bool mine = lane > a && lane < n;
    unsigned fl = flr[a];
    asm volatile("" : "+v"(fl));
    fl = mine ? fl : 0u;
    if (swap) fl = ((fl & 1u) << 1) | ((fl >> 1) & 1u);
    m0 = __builtin_amdgcn_ballot_w64((fl & 1u) != 0u);  // edge b -> a
    m1 = __builtin_amdgcn_ballot_w64((fl & 2u) != 0u);  // edge a -> b
  };
  auto next_row = [&]() {  // filter row of position q, and the request for position q + RING
    const V w = ring[q % RING];
    if (q + RING < NPOS) {
      ring[q % RING] = load_row(ap, bp);
      next_pos(ap, bp);
    }
    ++q;
    return w;
  };
  // (two loop nests with plain bounds: the unroller needs the trip counts)
  // ---- rows before the group: only the edges a -> b into the group's targets
#pragma unroll
  for (int a = 0; a < G0; ++a) {
    unsigned long long m0, m1;
    row_flags(a, m0, m1);
    const V xa = xr[a];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int b = G0 + i;
      const V w = next_row();
      const V t1 = xa * w;
      const V s1 = acc[i] + t1;
      acc[i] = ((m1 >> b) & 1ull) ? s1 : acc[i];
      asm volatile("" : "+v"(acc[i].x), "+v"(acc[i].y) : : "memory");  // keep the schedule as written (see above)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- rows of the group: target a from every partner b, and target b while b is in the group
#pragma unroll
  for (int a = G0; a < AEND; ++a) {
    unsigned long long m0, m1;
    row_flags(a, m0, m1);
    V acc_a = acc[a - G0];
    const V xa = xr[a];
#pragma unroll
    for (int b = a + 1; b < NMAX; ++b) {
      const V w = next_row();
      const V t0 = xr[b] * w;
      const V s0 = acc_a + t0;
      acc_a = ((m0 >> b) & 1ull) ? s0 : acc_a;
      if (b < G1) {
        const V t1 = xa * w;
        const V s1 = acc[b - G0] + t1;
        acc[b - G0] = ((m1 >> b) & 1ull) ? s1 : acc[b - G0];
        asm volatile("" : "+v"(acc[b - G0].x), "+v"(acc[b - G0].y) : : "memory");
      }
      asm volatile("" : "+v"(acc_a.x), "+v"(acc_a.y) : : "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    acc[a - G0] = acc_a;
  }
#undef AGG_BS
#undef AGG_BE
#pragma unroll
  for (int i = 0; i < NG; ++i)
    if (G0 + i < n && has_cols) *reinterpret_cast<V*>(out + (size_t)(a0 + G0 + i) * F + f) = acc[i];
}

}  // namespace geossl
