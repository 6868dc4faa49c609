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
// the widest column: col_lo(c) = floor(c * s + d) with s = W / C and d = (C - 1) / (2 C) < 1/2, so consecutive columns
// differ by floor(s) or floor(s) + 1, and as the C widths add up to W the larger one occurs whenever s is not an integer:
// max_c (col_lo(c + 1) - col_lo(c)) = ceil(W / C).  (The kernels used to walk all columns of all levels with two integer
// divisions each, in every workgroup, several times over: tens of microseconds before the first item of a launch.)
__device__ __forceinline__ int col_max_cell(const int W, const int C) { return (W + C - 1) / C; }

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

// floor(n / d) for uniform 0 <= n < 2^31, 1 <= d < 2^31, entirely on the scalar unit (s_mul_hi_u32): with
// m = floor((2^32 - 1) / d) the estimate q' = floor(n * m / 2^32) falls short of n / d by less than n / 2^32 < 1/2, so it
// is the quotient or one below it, and one comparison of the remainder settles which.  (The double-precision form
// above costs four vector instructions and a readfirstlane; an item decode holds some twenty divisions.)
struct UDiv {
  unsigned d, m;
};
__device__ __forceinline__ UDiv make_udiv(const int d)
{
  const unsigned du = (unsigned)__builtin_amdgcn_readfirstlane(d);
  return UDiv{du, (unsigned)__builtin_amdgcn_readfirstlane((int)(0xFFFFFFFFu / du))};
}
__device__ __forceinline__ int udiv_s(const int n, const UDiv u)
{
  const unsigned q = __umulhi((unsigned)n, u.m);
  const unsigned r = (unsigned)n - q * u.d;
  return (int)(q + (r >= u.d ? 1u : 0u));
}

// floor to int in one instruction (the compiler emits v_floor_f32 + v_cvt_i32_f32)
__device__ __forceinline__ int cvt_flr(const float x)
{
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// A wave's bounding box in ONE reduction.  lo = packed (u16, u16) low corner, hi = packed high corner: the maximum is
// taken as the minimum of complements, and after a half-wave swap lanes 0-31 reduce the low corners of all 64 lanes
// while lanes 32-63 reduce the complemented high corners.  Returns, in lane 31, min(lo) and, in lane 63, ~max(hi)
// (other lanes: partial results).  12 vector instructions against 30 for two separate reductions.
__device__ __forceinline__ unsigned wave_reduce_box(const unsigned lo, const unsigned hi)
{
  const auto s = __builtin_amdgcn_permlane32_swap(lo, ~hi, false, false);
  unsigned v = pk_min(s[0], s[1]);
  v = pk_min(v, dpp_u<0xB1>(v));    // quad_perm [1,0,3,2]
  v = pk_min(v, dpp_u<0x4E>(v));    // quad_perm [2,3,0,1]
  v = pk_min(v, dpp_u<0x141>(v));   // row_half_mirror
  v = pk_min(v, dpp_u<0x140>(v));   // row_mirror
  v = pk_min(v, dpp_u<0x142>(v));   // row_bcast:15 -- rows 1 and 3 take in lane 15 of rows 0 and 2
  return v;
}

// The workgroup's box from the NW per-wave records {min lo, ~max hi} (8 bytes each, contiguous, 16-byte aligned): every
// lane reads all of them (LDS broadcast) and reduces in registers; the result is uniform.
template <int NW>
__device__ __forceinline__ void block_box(const unsigned *rec, unsigned &lo, unsigned &hi)
{
  unsigned a = 0xFFFFFFFFu, b = 0xFFFFFFFFu;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    a = pk_min(a, rec[2 * w]);
    b = pk_min(b, rec[2 * w + 1]);
  }
  lo = (unsigned)__builtin_amdgcn_readfirstlane((int)a);
  hi = ~(unsigned)__builtin_amdgcn_readfirstlane((int)b);
}

// ---- quad-cooperative record access ------------------------------------------------------------------------------------
// A lane's records (128 B of sampling locations, 64 B of weights, 64 B of output per (query, head)) are 512 B - 1 KB apart
// from its neighbours': loaded lane by lane, every 16-byte access of a wave instruction touches a different 128-byte line
// and the texture-addresser serialises them (64 tag look-ups per instruction; measured: 16 K of the 77 K cycles an
// item took, and everything behind it in the queue waits).  So the 4 lanes of a quad fetch 64 CONSECUTIVE bytes of ONE
// record per instruction (16 look-ups), taking the quad's four records in turn, and the 4 x 4 block of 16-byte pieces is
// transposed in registers: instruction s gives lane i piece (s - i) % 4 of record s; the lane rotates its four registers
// by its own index (two layers of v_cndmask) and a quad_perm rotation by k delivers piece k of its own record.
template <int K>
__device__ __forceinline__ col_f32x4 quad_rot(const col_f32x4 v)      // lane c receives lane (c - K) % 4's value
{
  constexpr int ctrl = ((0 - K) & 3) | (((1 - K) & 3) << 2) | (((2 - K) & 3) << 4) | (((3 - K) & 3) << 6);
  if constexpr (K == 0) return v;
  return col_f32x4{dpp_f<ctrl>(v[0]), dpp_f<ctrl>(v[1]), dpp_f<ctrl>(v[2]), dpp_f<ctrl>(v[3])};
}
// x[k] <- x[(a + k) % 4] with the per-lane amount a = a0 + 2 * a1
__device__ __forceinline__ void rot_regs(col_f32x4 (&x)[4], const bool a0, const bool a1)
{
  col_f32x4 t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) t[k][e] = a0 ? x[(k + 1) & 3][e] : x[k][e];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) x[k][e] = a1 ? t[(k + 2) & 3][e] : t[k][e];
}
// in: x[s] = what this lane loaded for record s of its quad (piece (s - i) % 4); out: x[k] = piece k of its own record
__device__ __forceinline__ void quad_transpose_in(col_f32x4 (&x)[4], const bool i0, const bool i1)
{
  rot_regs(x, i0, i1);
  x[1] = quad_rot<1>(x[1]);
  x[2] = quad_rot<2>(x[2]);
  x[3] = quad_rot<3>(x[3]);
}

}  // namespace pct
