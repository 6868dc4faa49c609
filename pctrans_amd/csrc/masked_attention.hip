// Fused masked multi-head attention core on CDNA4 MFMA (gfx950, wave64), bf16 inputs / fp32 accumulate.
//
// Replaces the body of the reference's multi_head_attention_forward for the PCTrans decoder
// (transformer_decoder/attention.py:271-387):  q * head_dim^-0.5 -> bmm(q, k^T) -> masked_fill(-inf) -> softmax ->
// bmm(p, v), which torch runs as ~9 launches that materialise the [N*heads, Q, HW] score tensor three times
// (bmm output, masked copy, softmax output; 210 MB at HW = 4096, N = 16).  Under bf16 autocast (the configuration the
// reference trains/infers with, trainer.py:140) those bmm's take bf16 operands; here:
//
//   * one wave owns (image n, head h, 16 queries) and walks the keys 32 at a time with an online softmax; nothing but
//     q, k, v^T, the byte mask and the output ever touches HBM;
//   * S^T = K . Q^T is computed (keys on MFMA rows, queries on MFMA columns: v_mfma_f32_16x16x32_bf16, K = head_dim
//     = 32 -> one MFMA per 16x16 tile), so a query's scores live in ONE lane column: row statistics are 8 in-lane
//     values + two cross-lane-group steps, and the accumulator tile is already laid out as the B operand (P^T) of
//     O^T += V^T . P^T -- no LDS, no transposes, no barriers;
//   * the k-slot -> key assignment of that second MFMA is permuted (slot 8g+j <-> key 4g+j | 16+4g+(j-4)) so that P goes
//     from accumulator registers to operand registers without any lane movement; V^T is read with the same permutation.
//
// Layouts (all contiguous): q [Q, N, heads*HD] bf16, k [S, N, heads*HD] bf16, vT [N, heads*VD, S] bf16,
// mask [N, Q, S] bytes (nonzero = may not attend; shared by the heads) or NULL, out [Q, N, heads*VD] bf16 / fp32.
// Fully masked rows give NaN exactly like softmax over all -inf does in the reference (callers un-mask such rows
// first, mask2former_transformer_decoder.py:561).
#include "attn_common.hpp"

namespace pct {

// HD: q/k head dim (32 or 16), VD = 16.  OutT: __bf16 or float.
template <int HD, typename OutT>
__global__ __launch_bounds__(64) void masked_attention_kernel(const __bf16 *__restrict__ q, const __bf16 *__restrict__ k,
                                                              const __bf16 *__restrict__ vT,
                                                              const unsigned char *__restrict__ mask, const int N,
                                                              const int heads, const int Q, const int S,
                                                              const float scale, OutT *__restrict__ out)
{
  constexpr int VD = 16;
  const int lane = threadIdx.x, col = lane & 15, g = lane >> 4;
  const int qtiles = (Q + 15) / 16;
  // XCD-chunked block order: the query tiles of a head, and the heads of an image (whose K rows share 512-B lines),
  // run on one XCD and hit one L2 instead of eight
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = (int)(lb % (unsigned)qtiles);
  const int nh = (int)(lb / (unsigned)qtiles);
  const int h = nh % heads, n = nh / heads;
  const int E = heads * HD, EV = heads * VD;

  // B operand of S^T = K . Q^T : Q[query = q0 + col][dims]  (constant for the whole wave)
  const int qi = min(qt * 16 + col, Q - 1);
  const __bf16 *qrow = q + ((size_t)qi * N + n) * E + h * HD;
  bf16x8 qb;
  if constexpr (HD == 32) {
    qb = *reinterpret_cast<const bf16x8 *>(qrow + 8 * g);
  } else {   // HD == 16: k-slots 16..31 are zero padding
    qb = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (g < 2) qb = *reinterpret_cast<const bf16x8 *>(qrow + 8 * g);
  }

  const __bf16 *kbase = k + (size_t)n * E + h * HD;                 // + key * N * E
  const __bf16 *vrow = vT + ((size_t)n * EV + h * VD + col) * S;    // V^T[vd = col][key]
  const unsigned char *mrow = mask ? mask + ((size_t)n * Q + qi) * S : nullptr;

  f32x4 o = {0.f, 0.f, 0.f, 0.f};          // O^T[vd = 4g + r][query = col]
  float m_run = -INFINITY, l_run = 0.f;

  for (int key0 = 0; key0 < S; key0 += 32) {
    // ---- A operands of S^T: keys key0 + [0,16) and key0 + [16,32) ------------------------------------------------------
    bf16x8 a0, a1;
    {
      const int ka = min(key0 + col, S - 1), kb = min(key0 + 16 + col, S - 1);
      if constexpr (HD == 32) {
        a0 = *reinterpret_cast<const bf16x8 *>(kbase + (size_t)ka * N * E + 8 * g);
        a1 = *reinterpret_cast<const bf16x8 *>(kbase + (size_t)kb * N * E + 8 * g);
      } else {
        a0 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        a1 = a0;
        if (g < 2) {
          a0 = *reinterpret_cast<const bf16x8 *>(kbase + (size_t)ka * N * E + 8 * g);
          a1 = *reinterpret_cast<const bf16x8 *>(kbase + (size_t)kb * N * E + 8 * g);
        }
      }
    }
    // this lane holds, for query `col`, keys key0 + 4g + r (first tile) and key0 + 16 + 4g + r (second tile)
    const int kA = key0 + 4 * g, kB = key0 + 16 + 4 * g;
    unsigned mA = 0, mB = 0;
    if (mrow) {
      const bool al = (S & 3) == 0;
      if (al && kA + 3 < S) mA = *reinterpret_cast<const unsigned *>(mrow + kA);
      else
        for (int r = 0; r < 4; ++r) mA |= (kA + r < S ? (unsigned)(mrow[kA + r] != 0) : 1u) << (8 * r);
      if (al && kB + 3 < S) mB = *reinterpret_cast<const unsigned *>(mrow + kB);
      else
        for (int r = 0; r < 4; ++r) mB |= (kB + r < S ? (unsigned)(mrow[kB + r] != 0) : 1u) << (8 * r);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                   // keys past the end are dead
      if (kA + r >= S) mA |= 0xFFu << (8 * r);
      if (kB + r >= S) mB |= 0xFFu << (8 * r);
    }
    // ---- A operand of O^T += V^T . P^T : k-slot 8g+j <-> key (j < 4 ? kA + j : kB + j - 4), same permutation on both operands --
    bf16x8 va;
    if (kB + 3 < S && (S & 3) == 0) {
      const bf16x4 va0 = *reinterpret_cast<const bf16x4 *>(vrow + kA);
      const bf16x4 va1 = *reinterpret_cast<const bf16x4 *>(vrow + kB);
      va = bf16x8{va0[0], va0[1], va0[2], va0[3], va1[0], va1[1], va1[2], va1[3]};
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        va[r] = vrow[min(kA + r, S - 1)];        // dead keys have p = 0; clamped loads keep the address valid
        va[4 + r] = vrow[min(kB + r, S - 1)];
      }
    }
    attn_step(a0, a1, qb, va, mA, mB, scale, o, m_run, l_run);
  }

  if (qt * 16 + col < Q) {
    const float inv = 1.0f / l_run;                                  // l_run == 0 -> NaN, as softmax(all -inf)
    OutT *dst = out + ((size_t)(qt * 16 + col) * N + n) * EV + h * VD + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = (OutT)(o[r] * inv);
  }
}

// out_dtype: 0 = fp32, 2 = bf16
int launch_masked_attention(const void *q, const void *k, const void *vT, const unsigned char *mask, int N, int heads,
                            int Q, int S, int head_dim, int v_head_dim, float scale, int out_dtype, void *out,
                            hipStream_t stream)
{
  if (v_head_dim != 16 || (head_dim != 32 && head_dim != 16)) return -4;
  if (N == 0 || Q == 0) return 0;
  if (S <= 0) return -1;
  const long long nblk = (long long)N * heads * ((Q + 15) / 16);
  if (nblk > 0x7fffffffLL) return -4;
  const dim3 grid((unsigned)nblk), block(64);
  const __bf16 *qq = static_cast<const __bf16 *>(q), *kk = static_cast<const __bf16 *>(k);
  const __bf16 *vv = static_cast<const __bf16 *>(vT);
#define PCT_MA(HD_, OUT_)                                                                                        \
  hipLaunchKernelGGL((masked_attention_kernel<HD_, OUT_>), grid, block, 0, stream, qq, kk, vv, mask, N, heads, Q, S, \
                     scale, static_cast<OUT_ *>(out))
  if (head_dim == 32 && out_dtype == 2) PCT_MA(32, __bf16);
  else if (head_dim == 32 && out_dtype == 0) PCT_MA(32, float);
  else if (head_dim == 16 && out_dtype == 2) PCT_MA(16, __bf16);
  else if (head_dim == 16 && out_dtype == 0) PCT_MA(16, float);
  else return -1;
#undef PCT_MA
  return (int)hipGetLastError();
}

}  // namespace pct
