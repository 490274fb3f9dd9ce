// fp32 GEMMs on the bf16 matrix pipe of gfx950 ("3 x bf16 split").
//
// Measured on MI355X (scratch microbenchmarks): v_mfma_f32_32x32x2_f32 runs on the same FP32 lanes as
// the vector ALU - an f32 MFMA and the exp / softplus / address arithmetic of ANY wave on the SIMD serialise (the
// time of a kernel is MFMA time + VALU time), and 157 TFLOP/s is the ceiling of both together.  The bf16 MFMA has
// its own pipe (2.4 PFLOP/s measured) and does overlap with vector work issued by the same wave.
//
// Every fp32 operand is therefore split EXACTLY into three bf16 pieces, x = h + m + l (3 x 8 significant bits;
// bf16 has the exponent range of fp32, so no scaling is involved), and a product sum_k a_k b_k is accumulated in
// fp32 by six MFMAs over the piece products of weight 2^0 .. 2^-16:
//        l*h, h*l, m*m, m*h, h*m, h*h          (smallest first)
// The three dropped products (m*l, l*m, l*l) are <= 2^-24 relative to |a_k||b_k| - the size of the one rounding an
// fp32 FMA commits per product - so the result carries fp32-GEMM accuracy (tests: 1e-6 relative against an fp64
// evaluation, like the f32-MFMA path it replaces).  Six 32x32x16 bf16 MFMAs cost 192 matrix-pipe cycles per 16 k;
// eight 32x32x2 f32 MFMAs cost 512 cycles of the shared FP32 lanes.
//
// Operand maps of v_mfma_f32_32x32x16_bf16 (A and B are symmetric):
//     A: lane l holds A[row = l&31][k = 8 consecutive indices of half (l>>5)]      (one 16-byte register quad)
//     B: lane l holds B[k = same 8 indices][col = l&31]
//     C/D: as for the f32 MFMA: acc[reg], row = (reg&3) + 8*(reg>>2) + 4*(l>>5), col = l&31
// Which 8 contraction indices a (half, element) pair stands for is free as long as A and B agree.  `kperm` is the
// choice that makes the C layout of one product (register = 4 consecutive rows per group of 8) directly the operand
// layout of the next, so chained products need no transposition.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace geossl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// element e (0..7) of half kh of a 16-wide k-step <-> index (e&3) + 8*(e>>2) + 4*kh of the step: registers
// 0..7 / 8..15 of a C-layout accumulator are then elements 0..7 of two consecutive k-steps
__device__ __forceinline__ int kperm(int e, int kh) { return (e & 3) + 8 * (e >> 2) + 4 * kh; }

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {  // round-to-nearest-even pack: low half = a
  bf16x2 p = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, p);
}
// exact 3-way split of two fp32 values into packed bf16 pairs: v = h + m + l
__device__ __forceinline__ void split2(float v0, float v1, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = pk_bf16(v0, v1);
  float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
  m = pk_bf16(r0, r1);
  r0 -= __uint_as_float(m << 16);
  r1 -= __uint_as_float(m & 0xffff0000u);
  l = pk_bf16(r0, r1);
}
struct Frag3 {
  u32x4 h, m, l;
};
// 8 fp32 values (one lane's share of a 16-wide k-step) -> three bf16x8 fragments
__device__ __forceinline__ Frag3 split8(const float (&v)[8]) {
  Frag3 f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t h, m, l;
    split2(v[2 * q], v[2 * q + 1], h, m, l);
    f.h[q] = h;
    f.m[q] = m;
    f.l[q] = l;
  }
  return f;
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
  const f32x16 r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                           0, 0);
#ifdef GEOSSL_MFMA_PAD
  asm volatile("s_nop %0" ::"n"(GEOSSL_MFMA_PAD));
#endif
  return r;
}
// acc += A * B from piece fragments, smallest products first
__device__ __forceinline__ void mma6(f32x16& acc, const Frag3& a, const Frag3& b) {
  acc = mfma_bf16(a.l, b.h, acc);
  acc = mfma_bf16(a.h, b.l, acc);
  acc = mfma_bf16(a.m, b.m, acc);
  acc = mfma_bf16(a.m, b.h, acc);
  acc = mfma_bf16(a.h, b.m, acc);
  acc = mfma_bf16(a.h, b.h, acc);
}

// Same six products on two accumulators (their sum is the result): back-to-back MFMAs that chain through ONE
// accumulator pay a short issue bubble each (measured: -1.6 % on the filter backward when split), two alternating
// chains do not.  Caller adds acc_lo + acc_hi once at the end.
__device__ __forceinline__ void mma6x2(f32x16& acc_lo, f32x16& acc_hi, const Frag3& a, const Frag3& b) {
  acc_lo = mfma_bf16(a.l, b.h, acc_lo);
  acc_hi = mfma_bf16(a.m, b.h, acc_hi);
  acc_lo = mfma_bf16(a.h, b.l, acc_lo);
  acc_hi = mfma_bf16(a.h, b.m, acc_hi);
  acc_lo = mfma_bf16(a.m, b.m, acc_lo);
  acc_hi = mfma_bf16(a.h, b.h, acc_hi);
}

// ---------------------------------------------------------------------------------------------------------------
// Two fp16 pieces (filter forward): x = h + l with h = fp16(x), l = fp16(x - h) carries 22
// significant bits when |x| >= 2^-3 and an absolute error <= 2^-25 below that (fp16's subnormal spacing), and a
// product needs THREE MFMAs (l*h, h*l, h*h; the dropped l*l is <= 2^-22 relative) instead of the six of the bf16
// scheme - half the matrix-pipe work, which is what bounds these kernels once the clock has settled under load.
// fp16 has 5 exponent bits: the caller scales operands by powers of two (exact) so that the largest magnitude is in
// [2^13, 2^14) and undoes the scales on the fp32 result.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
struct Frag2 {
  u32x4 h, l;
};
// The split of a pair of values, plain form (7 vector instructions per value once the compiler has had its way with it:
// it converts each value twice).  Kept as the definition the fused forms below are tested against
// (tools/probes/split_probe.hip).
__device__ __forceinline__ void split2h_plain(float v0, float v1, uint32_t& h, uint32_t& l) {
  const f16x2 p = {(_Float16)v0, (_Float16)v1};  // v_cvt_pk_f16_f32, round to nearest even
  h = __builtin_bit_cast(uint32_t, p);
  const f16x2 q = {(_Float16)(v0 - (float)p[0]), (_Float16)(v1 - (float)p[1])};
  l = __builtin_bit_cast(uint32_t, q);
}
__device__ __forceinline__ Frag2 split8h_plain(const float (&v)[8]) {
  Frag2 f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t h, l;
    split2h_plain(v[2 * q], v[2 * q + 1], h, l);
    f.h[q] = h;
    f.l[q] = l;
  }
  return f;
}
// The same two pieces in 3 instructions per PAIR: v - h is exact in fp32 (h is the nearest fp16 of v), so the mixed-
// precision fused multiply-add h * (-1) + v, rounded once to fp16 into the low / high half of the destination
// (v_fma_mixlo_f16 / v_fma_mixhi_f16), gives the bits of "convert back, subtract, convert".  The operands enter the
// instruction as registers: whatever arithmetic produced v has been rounded to fp32 before - a compiler contraction
// cannot reach into one of the two conversions and not the other (painn_mma.hip's note on inexact products).
__device__ __forceinline__ void split2h(float v0, float v1, uint32_t& h, uint32_t& l) {
  const f16x2 p = {(_Float16)v0, (_Float16)v1};
  h = __builtin_bit_cast(uint32_t, p);
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(v0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(v1));
}
__device__ __forceinline__ Frag2 split8h(const float (&v)[8]) {
  Frag2 f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t h, l;
    split2h(v[2 * q], v[2 * q + 1], h, l);
    f.h[q] = h;
    f.l[q] = l;
  }
  return f;
}
// Pieces of v * s for a power of two s (the operand scales of the two-piece kernels), the multiplication inside the
// instructions: h = fp16(v s), l = fp16(v s - h) - 4 instructions per pair, scale included.  (For any other s the
// pieces would be those of the EXACT product, not of its fp32 rounding.)
__device__ __forceinline__ void split2h_scaled(float v0, float v1, float s, uint32_t& h, uint32_t& l) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(v0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(v1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(v1), "v"(s), "v"(h));
}
__device__ __forceinline__ Frag2 split8h_scaled(const float (&v)[8], float s) {
  Frag2 f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t h, l;
    split2h_scaled(v[2 * q], v[2 * q + 1], s, h, l);
    f.h[q] = h;
    f.l[q] = l;
  }
  return f;
}
__device__ __forceinline__ f32x16 mfma_f16(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// acc += A * B from two-piece fragments (the dropped l * l is <= 2^-22 of the product), smallest products first
__device__ __forceinline__ void mma3(f32x16& acc, const Frag2& a, const Frag2& b) {
  acc = mfma_f16(a.l, b.h, acc);
  acc = mfma_f16(a.h, b.l, acc);
  acc = mfma_f16(a.h, b.h, acc);
}
// the same on two accumulators (see mma6x2); caller adds acc_lo + acc_hi
__device__ __forceinline__ void mma3x2(f32x16& acc_lo, f32x16& acc_hi, const Frag2& a, const Frag2& b) {
  acc_lo = mfma_f16(a.l, b.h, acc_lo);
  acc_hi = mfma_f16(a.h, b.h, acc_hi);
  acc_lo = mfma_f16(a.h, b.l, acc_lo);
}
// exponent e of a magnitude m = f * 2^e, f in [0.5, 1), floored at -100 (keeps 2^(14-e) finite).  frexp gives 0 for
// m = 0; an all-zero tile must not look like a tile of magnitude one - a RUNNING exponent that starts from it would
// push later operands of size 1e-8 into fp16's subnormals - so zero maps to the floor as well.
__device__ __forceinline__ int mag_exponent(float m) {
  return m > 0.0f ? max(__builtin_amdgcn_frexp_expf(m), -100) : -100;
}
// power-of-two scale that brings a magnitude m into [2^13, 2^14) (2^114 for m = 0), and the exponent it used
__device__ __forceinline__ float pow2_scale_to_2p14(float m, int& e) {
  e = mag_exponent(m);
  return __builtin_amdgcn_ldexpf(1.0f, 14 - e);
}

}  // namespace geossl
