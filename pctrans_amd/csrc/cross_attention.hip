// Position-guided masked cross-attention core of the PCTrans decoder on CDNA4 MFMA (gfx950, wave64), bf16 operands / fp32
// accumulation -- the CrossAttentionLayer's own operand form (mask2former_transformer_decoder.py:130-183 -> attention.py:
// 271-387): per head the query and key are [content (16) | position (16)] halves coming from DIFFERENT projections, so the
// kernel takes the halves as separate tensors in the layout the projection GEMMs write ([tokens, N, heads * 16]) and the
// per-head concatenations (dec.py:160-172: two torch.cat passes over the keys per layer) never exist; the value tensor is
// read as the projection writes it too (no V^T GEMM).
//
// Same arithmetic, in the same order, as masked_attention.hip (S^T = K . Q^T with v_mfma_f32_16x16x32_bf16, online softmax
// over 32-key steps, P^T as the B operand of O^T += V^T . P^T with the k-slot permutation 8g+j <-> key 4g+j | 16+4g+(j-4)):
// outputs are bit-identical to that kernel on the concatenated operands (tests/test_cross_attention_gpu.py).  What changes
// is how the operands reach the matrix cores.  There one wave per (image, head, 16 queries) fetched its fragments straight
// from global memory: 64-byte head slices of a key 64 KB apart, value rows per channel, mask bytes per query row -- about
// 290 tag look-ups in the texture addresser per 32-key step and wave, the K / V bytes of an (image, head) fetched seven
// times over, 0.66 ms at 4 096 keys and batch 128 (2.4 % of the MFMA rate, 9 % of the HBM rate).  Here:
//   * workgroup = (image, 4 heads, up to 4 query tiles of 16): 4 waves = 4 heads, every wave walking all the keys for its
//     head and keeping up to four query tiles' softmax state and output accumulators in registers (four independent MFMA /
//     exp chains per wave to interleave);
//   * the keys arrive in chunks of 64: each thread fetches 16-byte pieces of whole 128-byte rows (4 heads x 16 dims: full
//     lines) of K-content, K-position and V, and of the 64 x 64 mask bytes, one chunk AHEAD into registers while the
//     current chunk is computed, then stores them into a double-buffered LDS image (one barrier per chunk);
//   * LDS image: [key][8 pieces of 16 B], piece c of key k stored at c ^ (k % 8): the ds_read_b128 of an A operand (16 keys x
//     one head's 32 bytes) and the ds_read_b64_tr_b16 of a V block (4 keys x 16 channels, delivered transposed: the
//     hardware does the V^T) are then bank-conflict free; mask rows as 16 dwords rotated by 2 * (query / 2);
//   * the two workgroups of an (image, head group) -- query tiles 0-3 and 4-6 at 100 queries -- sit next to each other in
//     the XCD-chunked block order, so the second one finds the K / V lines in L2.
#include "attn_common.hpp"

namespace pct {

constexpr int XA_HG = 4;      // heads per workgroup (one wave each)
constexpr int XA_T = 4;       // query tiles per wave
constexpr int XA_CH = 64;     // keys per chunk

typedef unsigned xa_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned xa_u32x2 __attribute__((ext_vector_type(2)));

// q_c, q_s [Q, N, heads*16]; k_c, k_p, v [S, N, heads*16] bf16; mask [N, Q, S] bytes (nonzero = may not attend) or NULL;
// row_open [N, Q] bytes or NULL: nonzero = this query's mask row is ignored (it attends everywhere) -- the decoder's "a query
// whose mask rules out every pixel attends everywhere instead" (dec.py:561) without re-writing the [N, Q, S] mask;
// out [Q, N, heads*16] bf16.  S % 64 == 0, heads % 4 == 0.
struct XaShared {
  xa_u32x4 sKc[2][XA_CH * 8], sKp[2][XA_CH * 8], sV[2][XA_CH * 8];
  unsigned sM[2][XA_T * 16][16];
};

// NT: the query tiles this workgroup's waves carry (1 .. XA_T; compile time, so that the tiles' chains are straight-line code
// the scheduler can interleave -- with a run-time count each tile became its own basic block)
template <bool HAS_MASK, int NT>
__device__ __forceinline__ void cross_attention_body(XaShared &sh, const __bf16 *__restrict__ q_c, const __bf16 *__restrict__ q_s,
                                                     const __bf16 *__restrict__ k_c, const __bf16 *__restrict__ k_p,
                                                     const __bf16 *__restrict__ v, const unsigned char *__restrict__ mask,
                                                     const unsigned char *__restrict__ row_open, const int N, const int heads,
                                                     const int Q, const int S, const float scale, const int tgroups,
                                                     __bf16 *__restrict__ out)
{
  auto &sKc = sh.sKc;
  auto &sKp = sh.sKp;
  auto &sV = sh.sV;
  auto &sM = sh.sM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int C = heads * 16;
  const int hgroups = heads / XA_HG;
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int tg = (int)(lb % (unsigned)tgroups);
  const int hg = (int)((lb / (unsigned)tgroups) % (unsigned)hgroups);
  const int n = (int)(lb / (unsigned)(tgroups * hgroups));
  const int h = hg * XA_HG + wave;
  const int tile0 = tg * XA_T;

  // ---- B operands of S^T = K . Q^T: Q[query][k-slot 8g .. 8g+7] = content dims 8g.. (g < 2) | position dims 8(g-2).. ------
  bf16x8 qb[NT];
  unsigned keep[NT];                                     // all ones, or 0 for a query whose mask is to be ignored (row_open)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int qi = min((tile0 + t) * 16 + col, Q - 1);
    keep[t] = (HAS_MASK && row_open && row_open[(size_t)n * Q + qi]) ? 0u : 0xFFFFFFFFu;
    const __bf16 *src = (g < 2 ? q_c : q_s) + ((size_t)qi * N + n) * C + h * 16 + 8 * (g & 1);
    qb[t] = *reinterpret_cast<const bf16x8 *>(src);
  }
  f32x4 o[NT];
  float m_run[NT], l_run[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_run[t] = -INFINITY;
    l_run[t] = 0.f;
  }

  // ---- staging: which 16-byte pieces of a chunk this thread moves ---------------------------------------------------------
  // K: 64 keys x {content, position} x 8 pieces = 1 024 pieces, 4 per thread; V: 512 pieces, 2 per thread; mask: 64 query rows
  // x 4 pieces of 16 keys, 1 per thread.  piece p of a tensor: key = p / 8, c = p % 8 -> the 128-byte row of the head group
  const size_t row_elems = (size_t)N * C;                                   // elements between two keys of one image
  const __bf16 *kc_img = k_c + (size_t)n * C + hg * (XA_HG * 16);
  const __bf16 *kp_img = k_p + (size_t)n * C + hg * (XA_HG * 16);
  const __bf16 *v_img = v + (size_t)n * C + hg * (XA_HG * 16);
  const int sk_key = tid >> 3, sk_c = tid & 7;                              // + 32 keys for the second piece
  const int sm_q = tid >> 2, sm_part = tid & 3;
  const unsigned char *m_row = nullptr;
  if constexpr (HAS_MASK)
    m_row = mask + ((size_t)n * Q + min(tile0 * 16 + sm_q, Q - 1)) * S + sm_part * 16;
  xa_u32x4 pkc[2], pkp[2], pv[2], pm;
  auto fetch = [&](const int key0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const size_t off = (size_t)(key0 + sk_key + 32 * i) * row_elems + sk_c * 8;
      pkc[i] = *reinterpret_cast<const xa_u32x4 *>(kc_img + off);
      pkp[i] = *reinterpret_cast<const xa_u32x4 *>(kp_img + off);
      pv[i] = *reinterpret_cast<const xa_u32x4 *>(v_img + off);
    }
    if constexpr (HAS_MASK) pm = *reinterpret_cast<const xa_u32x4 *>(m_row + key0);
  };
  auto stash = [&](const int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = sk_key + 32 * i;
      const int pos = key * 8 + (sk_c ^ (key & 7));
      sKc[buf][pos] = pkc[i];
      sKp[buf][pos] = pkp[i];
      sV[buf][pos] = pv[i];
    }
    if constexpr (HAS_MASK) {
      // row sm_q, dwords 4 part .. 4 part + 3, rotated by 2 * ((query % 16) / 2): two aligned dword pairs
      const int rot = 2 * ((sm_q & 15) >> 1);
      unsigned *row = sM[buf][sm_q];
      *reinterpret_cast<xa_u32x2 *>(row + ((4 * sm_part + rot) & 15)) = xa_u32x2{pm[0], pm[1]};
      *reinterpret_cast<xa_u32x2 *>(row + ((4 * sm_part + 2 + rot) & 15)) = xa_u32x2{pm[2], pm[3]};
    }
  };

  // per-lane LDS addresses of the fragments (fixed for the launch; + 16 keys / + 32 keys by immediate offsets)
  // A operand: key kb + col, piece 2 wave + (g & 1) of content (g < 2) or position
  const int a_c = 2 * wave + (g & 1);
  // V block of group g: rows kb + 4 g + (l / 4), pieces 2 wave + (l % 4) / 2, half (l % 2), l = lane % 16
  const int v_r = 4 * g + (col >> 2), v_c = 2 * wave + ((col & 3) >> 1), v_h = col & 1;
  const int m_rot = 2 * (col >> 1);

  const int nch = S / XA_CH;
  fetch(0);
  stash(0);
  __syncthreads();
  if (nch > 1) fetch(XA_CH);
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    const unsigned char *kt = reinterpret_cast<const unsigned char *>(g < 2 ? sKc[buf] : sKp[buf]);
    const unsigned char *vt = reinterpret_cast<const unsigned char *>(sV[buf]);
#pragma unroll
    for (int j = 0; j < XA_CH / 32; ++j) {
      const int kb = 32 * j;
      // ---- fragments of this 32-key step, shared by the wave's query tiles ------------------------------------------------
      const int ka = kb + col, kb2 = kb + 16 + col;
      const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(kt + (ka * 8 + (a_c ^ (ka & 7))) * 16);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(kt + (kb2 * 8 + (a_c ^ (kb2 & 7))) * 16);
      const int va_k = kb + v_r, vb_k = kb + 16 + v_r;
      const bf16x4 va0 = lds_read_tr16_b64(vt + (va_k * 8 + (v_c ^ (va_k & 7))) * 16 + v_h * 8);
      const bf16x4 va1 = lds_read_tr16_b64(vt + (vb_k * 8 + (v_c ^ (vb_k & 7))) * 16 + v_h * 8);
      const bf16x8 va = {va0[0], va0[1], va0[2], va0[3], va1[0], va1[1], va1[2], va1[3]};
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        unsigned mA = 0, mB = 0;
        if constexpr (HAS_MASK) {
          const unsigned *mr = sM[buf][t * 16 + col];
          mA = mr[(8 * j + g + m_rot) & 15] & keep[t];
          mB = mr[(8 * j + 4 + g + m_rot) & 15] & keep[t];
        }
        attn_step(a0, a1, qb[t], va, mA, mB, scale, o[t], m_run[t], l_run[t]);
      }
    }
    if (ch + 1 < nch) {
      stash(buf ^ 1);               // (everyone left that buffer before the barrier that ended the previous chunk)
      __syncthreads();
      if (ch + 2 < nch) fetch((ch + 2) * XA_CH);
    }
  }

#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int qo = (tile0 + t) * 16 + col;
    if (qo < Q) {
      const float inv = 1.0f / l_run[t];                             // l_run == 0 -> NaN, as softmax(all -inf)
      __bf16 *dst = out + ((size_t)qo * N + n) * C + h * 16 + 4 * g;
      bf16x4 r = {(__bf16)(o[t][0] * inv), (__bf16)(o[t][1] * inv), (__bf16)(o[t][2] * inv), (__bf16)(o[t][3] * inv)};
      *reinterpret_cast<bf16x4 *>(dst) = r;
    }
  }
}

template <bool HAS_MASK>
__global__ __launch_bounds__(256, 2) void cross_attention_kernel(const __bf16 *__restrict__ q_c, const __bf16 *__restrict__ q_s,
                                                                 const __bf16 *__restrict__ k_c, const __bf16 *__restrict__ k_p,
                                                                 const __bf16 *__restrict__ v,
                                                                 const unsigned char *__restrict__ mask,
                                                                 const unsigned char *__restrict__ row_open, const int N,
                                                                 const int heads, const int Q, const int S, const float scale,
                                                                 const int tgroups, __bf16 *__restrict__ out)
{
  __shared__ __attribute__((aligned(16))) XaShared sh;
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int tg = (int)(lb % (unsigned)tgroups);
  const int ntile = min(XA_T, (Q + 15) / 16 - tg * XA_T);            // uniform over the workgroup
  static_assert(XA_T == 4, "dispatch below");
  if (ntile == 4) cross_attention_body<HAS_MASK, 4>(sh, q_c, q_s, k_c, k_p, v, mask, row_open, N, heads, Q, S, scale, tgroups, out);
  else if (ntile == 3) cross_attention_body<HAS_MASK, 3>(sh, q_c, q_s, k_c, k_p, v, mask, row_open, N, heads, Q, S, scale, tgroups, out);
  else if (ntile == 2) cross_attention_body<HAS_MASK, 2>(sh, q_c, q_s, k_c, k_p, v, mask, row_open, N, heads, Q, S, scale, tgroups, out);
  else cross_attention_body<HAS_MASK, 1>(sh, q_c, q_s, k_c, k_p, v, mask, row_open, N, heads, Q, S, scale, tgroups, out);
}

// -100: geometry not covered (the caller concatenates the halves and uses pct_masked_attention_bf16)
int launch_cross_attention(const void *q_c, const void *q_s, const void *k_c, const void *k_p, const void *v,
                           const unsigned char *mask, const unsigned char *row_open, int N, int heads, int Q, int S,
                           float scale, void *out, hipStream_t stream)
{
  if (heads <= 0 || heads % XA_HG != 0 || S <= 0 || S % XA_CH != 0) return -100;
  if (N == 0 || Q == 0) return 0;
  if ((long long)S * N * heads * 16 >= 0x7fffffffLL) return -100;
  if ((((uintptr_t)q_c | (uintptr_t)q_s | (uintptr_t)k_c | (uintptr_t)k_p | (uintptr_t)v) & 15u) || ((uintptr_t)out & 7u)) return -100;
  if (mask && (((uintptr_t)mask & 15u) || (S & 15))) return -100;
  const int qtiles = (Q + 15) / 16, tgroups = (qtiles + XA_T - 1) / XA_T;
  const long long nblk = (long long)N * (heads / XA_HG) * tgroups;
  if (nblk > 0x7fffffffLL) return -100;
  const dim3 grid((unsigned)nblk), block(256);
  const __bf16 *qc = static_cast<const __bf16 *>(q_c), *qs = static_cast<const __bf16 *>(q_s);
  const __bf16 *kc = static_cast<const __bf16 *>(k_c), *kp = static_cast<const __bf16 *>(k_p);
  const __bf16 *vv = static_cast<const __bf16 *>(v);
  if (mask)
    hipLaunchKernelGGL(cross_attention_kernel<true>, grid, block, 0, stream, qc, qs, kc, kp, vv, mask, row_open, N, heads, Q, S,
                       scale, tgroups, static_cast<__bf16 *>(out));
  else
    hipLaunchKernelGGL(cross_attention_kernel<false>, grid, block, 0, stream, qc, qs, kc, kp, vv, mask, row_open, N, heads, Q, S,
                       scale, tgroups, static_cast<__bf16 *>(out));
  return (int)hipGetLastError();
}

}  // namespace pct
