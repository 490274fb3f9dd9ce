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

}  // namespace geossl
