// Pieces shared by the bf16 MFMA attention kernels (masked_attention.hip: generic operands from global memory;
// cross_attention.hip: the decoder's cross-attention with LDS-staged operands).  One 32-key step of the online softmax lives
// here so that both kernels evaluate the same expressions in the same order: their outputs are bit-identical.
#pragma once
#include <math.h>

#include "msda_common.hpp"

namespace pct {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Reductions over the four 16-lane groups of a wave (lane ^ 16, lane ^ 32) on the vector unit: v_permlane16_swap hands the
// even rows the odd rows' value and the reverse, v_permlane32_swap does the same for the two halves; op(first result, second
// result) is the pair's combination in both.  (Was: two ds_bpermute round trips through the LDS pipe per reduction, four
// per 32-key step, each a full LDS latency on the softmax's critical path.)  Inline asm with the two wait states a vector
// write -> v_permlane read needs inside the string: given the same value twice, the builtins' two results are taken for
// equal by hipcc (ROCm 7.2) and folded.  max and + are commutative: the same bits as the lane ^ 16, lane ^ 32 form.
__device__ __forceinline__ void xgroup_swap16(float &a, float &b)
{
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void xgroup_swap32(float &a, float &b)
{
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xgroup_max(float v)
{
  float a = v, b = v;
  xgroup_swap16(a, b);
  a = b = fmaxf(a, b);
  xgroup_swap32(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float xgroup_sum(float v)
{
  float a = v, b = v;
  xgroup_swap16(a, b);
  a = b = a + b;
  xgroup_swap32(a, b);
  return a + b;
}

// ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns of 16-bit elements, delivered column-major -- lane
// 4 q + p of the group supplies the address of row q, columns 4 p .. 4 p + 3; lane i receives column i of the four rows.
// (EXEC must be all ones: every lane's address takes part.)
__device__ __forceinline__ bf16x4 lds_read_tr16_b64(const unsigned char *p)
{
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  const s16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t *)(__attribute__((address_space(3))) void *)p);
  return __builtin_bit_cast(bf16x4, r);
}

// One 32-key step for 16 queries (MFMA columns) of one head.
//   a0, a1: A operands K[key = k0 + col | k0 + 16 + col][dims 8g .. 8g+7];  qb: B operand Q[query col][dims 8g ..];
//   va: A operand of O^T += V^T . P^T, V^T[vd = col][k-slot 8g + j <-> key (j < 4 ? k0 + 4g + j : k0 + 16 + 4g + j - 4)];
//   mA / mB: four mask bytes for keys k0 + 4g + r / k0 + 16 + 4g + r, nonzero = dead (masked, or past the last key).
// This lane holds, for query `col`, the scores of keys k0 + 4g + r (s0[r]) and k0 + 16 + 4g + r (s1[r]): a query's row
// statistics are 8 in-lane values and two cross-group steps, and the probabilities go from accumulator to operand registers
// without any lane movement.
__device__ __forceinline__ void attn_step(const bf16x8 a0, const bf16x8 a1, const bf16x8 qb, const bf16x8 va, const unsigned mA,
                                          const unsigned mB, const float scale, f32x4 &o, float &m_run, float &l_run)
{
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qb, z, 0, 0, 0);
  const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qb, z, 0, 0, 0);
  float p0[4], p1[4];
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool deadA = ((mA >> (8 * r)) & 0xFFu) != 0;
    const bool deadB = ((mB >> (8 * r)) & 0xFFu) != 0;
    // (the products are made opaque so that they are never contracted with the subtraction below: without a mask the select
    // is gone and hipcc would fuse the two into one FMA, one rounding less than the masked instantiation has; HIP's
    // __fmul_rn is a plain multiplication and does not prevent it)
    float x0 = s0[r] * scale, x1 = s1[r] * scale;
    asm volatile("" : "+v"(x0), "+v"(x1));
    p0[r] = deadA ? -INFINITY : x0;
    p1[r] = deadB ? -INFINITY : x1;
    mx = fmaxf(mx, fmaxf(p0[r], p1[r]));
  }
  mx = xgroup_max(mx);
  const float m_new = fmaxf(m_run, mx);
  const float m_safe = m_new == -INFINITY ? 0.f : m_new;          // nothing attendable yet: keep everything 0
  const float alpha = __expf(m_run - m_safe);                     // exp(-inf) = 0 on the first live step
  float rs = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    p0[r] = __expf(p0[r] - m_safe);
    p1[r] = __expf(p1[r] - m_safe);
    rs += p0[r] + p1[r];
  }
  rs = xgroup_sum(rs);
  l_run = l_run * alpha + rs;
  m_run = m_new;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] *= alpha;
  const bf16x8 pb = {(__bf16)p0[0], (__bf16)p0[1], (__bf16)p0[2], (__bf16)p0[3],
                     (__bf16)p1[0], (__bf16)p1[1], (__bf16)p1[2], (__bf16)p1[3]};
  o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o, 0, 0, 0);
}

}  // namespace pct
