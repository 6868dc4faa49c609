// Helpers shared by the pyramid-column MSDeformAttn kernels (msda_forward_col.hip: fp32 values, 4 points;
// msda_forward_col16.hip: 16-bit values, 4 or 8 points).
#pragma once
#include "msda_win_common.hpp"

namespace pct {

typedef int col_i32x4 __attribute__((ext_vector_type(4)));
typedef float col_f32x2 __attribute__((ext_vector_type(2)));
typedef float col_f32x4 __attribute__((ext_vector_type(4)));

// first pixel x of column c (of C) on a level W pixels wide: the pixels whose centre (x + 0.5) / W lies in
// [c / C, (c + 1) / C); every pixel belongs to exactly one column, col_lo(C) == W
__device__ __forceinline__ int col_lo(const int c, const int W, const int C) { return (2 * c * W + C - 1) / (2 * C); }

// a wave-uniform float, pinned to a scalar register: gfx950 has no scalar float unit, so a uniform float expression is
// evaluated on the vector unit and -- hoisted out of the item loop -- would otherwise occupy a vector register for the
// whole kernel (dozens of them here: per-level sizes, reciprocals, clamps)
__device__ __forceinline__ float uni(const float v)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ double uni_d(const double v)
{
  // (explicit v_readfirstlane: the builtin is folded away on a value the compiler already knows to be uniform, and the
  // vector copy then stays alive across the item loop -- five of these were spilled)
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned ul = (unsigned)u, uh = (unsigned)(u >> 32);
  unsigned lo, hi;
  asm volatile("v_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3" : "=s"(lo), "=s"(hi) : "v"(ul), "v"(uh));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// a uniform (x, y) pair in an aligned scalar register pair: a packed FMA takes it as an operand directly (built in vector
// registers instead, the per-level constant pairs were hoisted out of the item loop and spilled)
__device__ __forceinline__ col_f32x2 uni_pair(const float a, const float b)
{
  const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
  unsigned lo, hi;
  asm volatile("v_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3" : "=s"(lo), "=s"(hi) : "v"(ua), "v"(ub));
  return __builtin_bit_cast(col_f32x2, ((unsigned long long)hi << 32) | lo);
}

// floor(n / d) for uniform 0 <= n < 2^31 given inv = 1.0 / d: (n + 0.5) / d is at least 0.5 / d away from an integer and
// the double product is off by less than 2^-20 of that margin.  An integer division by a run-time divisor costs ~35
// dependent instructions here (no hardware divide); the item decode had twenty of them per item.
__device__ __forceinline__ int udiv_by(const int n, const double inv)
{
  return __builtin_amdgcn_readfirstlane((int)(((double)n + 0.5) * inv));
}

}  // namespace pct
