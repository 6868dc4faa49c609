// Pieces shared by the bf16 dynamic-mask-head kernels (dyn_mask_head_mfma.hip: MLP + streaming resize in two launches;
// dyn_mask_head_fused.hip: one launch).  The blend and mask expressions live here so that both paths evaluate the SAME
// floating-point expression trees (explicit fmaf: nothing is left to the compiler's contraction choices) and produce
// bit-identical outputs -- tests/test_dmh_fused_gpu.py holds them to that.
#pragma once
#include "msda_common.hpp"

#ifndef PCT_DMH_L2_MFMA
#define PCT_DMH_L2_MFMA 1
#endif

namespace pct {

typedef __bf16 dm_bf16x4 __attribute__((ext_vector_type(4)));
typedef short dm_s16x4 __attribute__((ext_vector_type(4)));
typedef float dm_f32x4 __attribute__((ext_vector_type(4)));
typedef float dm_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 dm_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned dm_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned dm_u32x4 __attribute__((ext_vector_type(4)));

// two bf16 (round to nearest even) in one dword: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b)
{
  return __builtin_bit_cast(unsigned, __builtin_convertvector(dm_f32x2{a, b}, dm_bf16x2));
}
// two v_cvt_pk_bf16_f32, no per-element conversions and byte permutes
__device__ __forceinline__ dm_s16x4 pack_bf16x4(float a, float b, float c, float d)
{
  const dm_u32x2 v = {pack_bf16x2(a, b), pack_bf16x2(c, d)};
  return __builtin_bit_cast(dm_s16x4, v);
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }

// relu as ONE instruction the compiler knows: a signed-integer max with 0 on the bit pattern (a negative float, -0
// included, is a negative int32; a positive one is unchanged).  (fmaxf / v_med3_f32 on an MFMA result make hipcc emit a
// canonicalising v_max_f32 x, x, x first; an inline-asm v_max_f32 is worse: hipcc pads no hazards for an asm statement, and
// a vector instruction that reads an MFMA result needs wait states behind the MFMA -- whether it got them depended on what
// the scheduler happened to place in between, and in the one-launch kernel it did not: garbage activations.)
__device__ __forceinline__ float dm_relu(float x)
{
  return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0));
}
// relu of four values already packed as bf16: a signed 16-bit max with 0 (a negative bf16, -0 included, is a negative
// int16) -- the same bits as rounding relu(x), in two instructions instead of four
typedef short dm_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dm_s16x4 dm_relu_packed(dm_s16x4 v)
{
  return __builtin_elementwise_max(v, dm_s16x4{0, 0, 0, 0});
}

// PyTorch upsample_bilinear2d source index (align_corners = False)
__device__ __forceinline__ void dm_bilinear_src(int dst, float scale, int in_size, int &i0, int &i1, float &lam)
{
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  lam = src - (float)i0;
}

// one bilinear output:  h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d)   (a, b: the upper source row's two columns;
// c, d: the lower row's) -- the order PyTorch's upsample_bilinear2d kernel evaluates it in
__device__ __forceinline__ float dm_blend_x(float w0, float w1, float a, float b) { return fmaf(w1, b, w0 * a); }
__device__ __forceinline__ float dm_blend_y(float h0, float h1, float t0, float t1) { return fmaf(h1, t1, h0 * t0); }
__device__ __forceinline__ float dm_blend(float h0, float h1, float w0, float w1, float a, float b, float c, float d)
{
  return dm_blend_y(h0, h1, dm_blend_x(w0, w1, a, b), dm_blend_x(w0, w1, c, d));
}

// attention-mask byte of one target pixel: bilinear value -> bf16 (torch's interpolate output under autocast) ->
// sigmoid in bf16 -> "< 0.5" = may not attend (mask2former_transformer_decoder.py:689-691)
__device__ __forceinline__ unsigned char dm_mask_byte(float ly, float lx, float a, float b, float c, float d)
{
  float v = dm_blend(1.f - ly, ly, 1.f - lx, lx, a, b, c, d);
  v = (float)(__bf16)v;
  const float s = (float)(__bf16)(1.f / (1.f + expf(-v)));
  return s < 0.5f ? 1 : 0;
}

// ---- the per-query 3-layer MLP on the bf16 matrix cores (shared by both bf16 paths) --------------------------------------
// One MFMA tile = 2 queries x 8 hidden rows (M = 16) by 16 pixels (N = 16).  lane = (g = lane / 16, col = lane % 16):
// accumulator rows 4g .. 4g+3 of pixel column `col`; groups 0, 1 belong to the pair's first query, groups 2, 3 to its second.
// Layer 0 is ONE v_mfma_f32_16x16x32_bf16 (a 16x16x16 costs the same 16 cycles for half the depth): k-slots 0 .. 15 are the
// 16 feature channels, slots 16 .. 27 the relative-coordinate term -- wx * (ref_x - loc_x) + wy * (ref_y - loc_y) splits into
// a per-query constant (folded into the bias) and a per-pixel part that goes through the matrix core EXACTLY: a pixel
// coordinate (an integer below 2^16) is the sum of two bf16 pieces, a weight the sum of three, and the six products per axis
// are twelve k-slots (weight piece i times coordinate piece j; exact products, fp32 accumulation).  Layer 1 (block-diagonal
// 8 x 8 per query) is a v_mfma_f32_16x16x16_bf16 whose B operand is layer 0's accumulator, relu'd and packed in place;
// layer 2 (8 MACs per query) is 4 FMAs per lane and one cross-group add.
constexpr int DMH_C = 16, DMH_HID = 8;
typedef __bf16 dm_bf16x8 __attribute__((ext_vector_type(8)));

// B operand of layer 0 for one pixel (the lane's pixel column of a tile): k = 8g .. 8g+7
template <bool REL>
__device__ __forceinline__ dm_u32x4 dmh_pixel_operand(const float *__restrict__ fimg, const int HW, const int W, const int px,
                                                      const int g, const int stride)
{
  dm_u32x4 b = {0u, 0u, 0u, 0u};
  if (g < 2) {
    const float *fp = fimg + (size_t)(8 * g) * HW + px;
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = pack_bf16x2(fp[(size_t)(2 * j) * HW], fp[(size_t)(2 * j + 1) * HW]);
  } else if constexpr (REL) {
    const float half = (float)(stride / 2);
    const int y = px / W, x = px - y * W;
    const float lxf = (float)(x * stride) + half, lyf = (float)(y * stride) + half;
    const float xh = (float)(__bf16)lxf, yh = (float)(__bf16)lyf;
    const float xl = lxf - xh, yl = lyf - yh;                  // exact, and exactly representable (integers < 2^16)
    const unsigned px2 = pack_bf16x2(xh, xl), py2 = pack_bf16x2(yh, yl);
    b = g == 2 ? dm_u32x4{px2, px2, px2, py2} : dm_u32x4{py2, py2, 0u, 0u};
  }
  return b;
}

// The generated parameters of one query PAIR as this lane needs them (params [N, Q, G] in parse_dynamic_params order:
// w0 | w1 | w2 | b0 | b1 | b2), and the MLP of pixel tiles.
template <bool REL>
struct DmhPair {
  static constexpr int C = DMH_C, HID = DMH_HID, CIN = REL ? C + 2 : C;
  static constexpr int G = CIN * HID + HID * HID + HID + HID + HID + 1;
  static constexpr int OFF_W1 = CIN * HID, OFF_W2 = OFF_W1 + HID * HID, OFF_B0 = OFF_W2 + HID;
  static constexpr int OFF_B1 = OFF_B0 + HID, OFF_B2 = OFF_B1 + HID;
  static constexpr int PREP_DWORDS = 20;                          // prepared form: a0 (4), a1 (2), b0 (4), b1 (4), w2 (4), b2, pad
  dm_u32x4 a0;
  dm_s16x4 a1, a2;
  float b0v[4], b1v[4], w2v[4], b2;
  // layer 2 (8 -> 1 per query) as a third MFMA: A rows 0 and 4 carry the first query's weights, rows 8 and 12 the second's, the
  // other rows are zero -- the accumulator register 0 of lane (pixel, g) is then row 4 g = its own query's logit.  This lane's A
  // slots k = 4g .. 4g+3 are the hidden units its w2v[] belong to.
  __device__ __forceinline__ void make_a2(const int col, const int g)
  {
    a2 = ((col & 3) == 0 && (col >> 3) == (g >> 1)) ? pack_bf16x4(w2v[0], w2v[1], w2v[2], w2v[3]) : dm_s16x4{0, 0, 0, 0};
  }

  // the pair's parameters as loaded (fetch) and as the MFMAs take them (prepare)
  struct Raw {
    float w0[8], w1[4], b0[4], b1[4], w2[4], wx[4], wy[4], b2, rx, ry, wxr, wyr;
  };
  static __device__ __forceinline__ Raw fetch(const float *__restrict__ params, const float *__restrict__ ref, const int n,
                                              const int Q, const int pr, const int col, const int g)
  {
    Raw w;
    // the query this lane's accumulator rows (4g .. 4g+3) belong to, and the query of my A-operand row `col`
    const int q_acc = min(2 * pr + (g >> 1), Q - 1);
    const int q_row = min(2 * pr + (col >> 3), Q - 1);
    const float *pa = params + ((size_t)n * Q + q_acc) * G;      // for accumulator-side constants
    const float *prw = params + ((size_t)n * Q + q_row) * G;     // for A-operand rows
    const int hr = col & 7;                                      // hidden row of A-operand row `col`
    const int r0 = (4 * g) & 7;                                  // first hidden row of my accumulator rows
    const float *w0r = prw + hr * CIN + (REL ? 2 : 0) + 8 * (g & 1);   // layer 0 (groups 0, 1): W0feat[q_row][hr][ch = 8g + j]
    const float *w1r = prw + OFF_W1 + hr * HID + ((4 * g) & 7);  // layer 1: W1[q_row][hr][(4g + j) & 7]
#pragma unroll
    for (int j = 0; j < 8; ++j) w.w0[j] = w0r[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      w.w1[r] = w1r[r];
      w.b0[r] = pa[OFF_B0 + r0 + r];
      w.b1[r] = pa[OFF_B1 + r0 + r];
      w.w2[r] = pa[OFF_W2 + r0 + r];
      w.wx[r] = REL ? pa[(r0 + r) * CIN + 0] : 0.f;
      w.wy[r] = REL ? pa[(r0 + r) * CIN + 1] : 0.f;
    }
    w.b2 = pa[OFF_B2];
    w.rx = w.ry = w.wxr = w.wyr = 0.f;
    if constexpr (REL) {
      w.rx = ref[((size_t)n * Q + q_acc) * 2];
      w.ry = ref[((size_t)n * Q + q_acc) * 2 + 1];
      w.wxr = prw[hr * CIN + 0];
      w.wyr = prw[hr * CIN + 1];
    }
    return w;
  }
  __device__ __forceinline__ void prepare(const Raw &w, const int col, const int g, const int H, const int W, const int stride)
  {
    // A operand of layer 0, k = 8g .. 8g+7: feature weights (groups 0, 1) or minus the coordinate weights' pieces (2, 3)
    a0 = dm_u32x4{pack_bf16x2(w.w0[0], w.w0[1]), pack_bf16x2(w.w0[2], w.w0[3]), pack_bf16x2(w.w0[4], w.w0[5]),
                  pack_bf16x2(w.w0[6], w.w0[7])};
    // layer 1 is block-diagonal: row `col` (query col>>3, hidden hr) x k = 4g + j (query g>>1, hidden (4g+j)&7)
    a1 = (col >> 3) == (g >> 1) ? pack_bf16x4(w.w1[0], w.w1[1], w.w1[2], w.w1[3]) : dm_s16x4{0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      b0v[r] = w.b0[r];
      b1v[r] = w.b1[r];
      w2v[r] = w.w2[r];
    }
    b2 = w.b2;
    make_a2(col, g);
    if (g >= 2) a0 = dm_u32x4{0u, 0u, 0u, 0u};
    if constexpr (REL) {
      const float rx = w.rx * (float)(W * stride), ry = w.ry * (float)(H * stride);
#pragma unroll
      for (int r = 0; r < 4; ++r) b0v[r] = fmaf(w.wy[r], ry, fmaf(w.wx[r], rx, b0v[r]));
      const float wxr = w.wxr, wyr = w.wyr;
      const float x1 = (float)(__bf16)wxr, x2 = (float)(__bf16)(wxr - x1), x3 = (float)(__bf16)((wxr - x1) - x2);
      const float y1 = (float)(__bf16)wyr, y2 = (float)(__bf16)(wyr - y1), y3 = (float)(__bf16)((wyr - y1) - y2);
      if (g == 2) a0 = dm_u32x4{pack_bf16x2(-x1, -x1), pack_bf16x2(-x2, -x2), pack_bf16x2(-x3, -x3), pack_bf16x2(-y1, -y1)};
      if (g == 3) a0 = dm_u32x4{pack_bf16x2(-y2, -y2), pack_bf16x2(-y3, -y3), 0u, 0u};
    }
  }
  __device__ __forceinline__ void load(const float *__restrict__ params, const float *__restrict__ ref, const int n,
                                       const int Q, const int pr, const int col, const int g, const int H, const int W,
                                       const int stride)
  {
    prepare(fetch(params, ref, n, Q, pr, col, g), col, g, H, W, stride);
  }
  // the prepared form as PREP_DWORDS dwords per lane, [k / 4][lane][k % 4]: five coalesced 16-byte accesses per lane
  __device__ __forceinline__ void store_prepared(unsigned *__restrict__ dst, const int lane) const
  {
    dm_u32x4 *d = reinterpret_cast<dm_u32x4 *>(dst) + lane;
    const dm_u32x2 a1u = __builtin_bit_cast(dm_u32x2, a1);
    d[0] = a0;
    d[64] = dm_u32x4{a1u[0], a1u[1], __builtin_bit_cast(unsigned, b2), 0u};
    d[128] = dm_u32x4{__builtin_bit_cast(unsigned, b0v[0]), __builtin_bit_cast(unsigned, b0v[1]),
                      __builtin_bit_cast(unsigned, b0v[2]), __builtin_bit_cast(unsigned, b0v[3])};
    d[192] = dm_u32x4{__builtin_bit_cast(unsigned, b1v[0]), __builtin_bit_cast(unsigned, b1v[1]),
                      __builtin_bit_cast(unsigned, b1v[2]), __builtin_bit_cast(unsigned, b1v[3])};
    d[256] = dm_u32x4{__builtin_bit_cast(unsigned, w2v[0]), __builtin_bit_cast(unsigned, w2v[1]),
                      __builtin_bit_cast(unsigned, w2v[2]), __builtin_bit_cast(unsigned, w2v[3])};
  }
  struct Prepared {
    dm_u32x4 v[5];
  };
  static __device__ __forceinline__ Prepared fetch_prepared(const unsigned *__restrict__ src, const int lane)
  {
    const dm_u32x4 *d = reinterpret_cast<const dm_u32x4 *>(src) + lane;
    Prepared p;
#pragma unroll
    for (int k = 0; k < 5; ++k) p.v[k] = d[64 * k];
    return p;
  }
  __device__ __forceinline__ void take(const Prepared &p)
  {
    // (whole-vector casts: __builtin_bit_cast(float, v[r]) on an ELEMENT of an ext_vector yields element 0 for every r with
    // hipcc of ROCm 7.2 -- found by comparing these fields against directly loaded ones inside the kernel)
    a0 = p.v[0];
    a1 = __builtin_bit_cast(dm_s16x4, dm_u32x2{p.v[1][0], p.v[1][1]});
    const dm_f32x4 m1 = __builtin_bit_cast(dm_f32x4, p.v[1]);
    const dm_f32x4 f0 = __builtin_bit_cast(dm_f32x4, p.v[2]), f1 = __builtin_bit_cast(dm_f32x4, p.v[3]);
    const dm_f32x4 f2 = __builtin_bit_cast(dm_f32x4, p.v[4]);
    b2 = m1[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      b0v[r] = f0[r];
      b1v[r] = f1[r];
      w2v[r] = f2[r];
    }
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    make_a2(lane & 15, lane >> 4);
  }

  // fp32 logits of query (g >> 1) at the lane's pixel column of NT tiles: the same value in both groups of a query.
  // Stage by stage over the NT tiles, so that every MFMA -> vector -> MFMA dependency of one tile has the other tiles' work
  // to hide behind (tile by tile, a wave at two per SIMD spent most of a pair waiting on its own matrix results).
  template <int NT>
  __device__ __forceinline__ void tiles(const dm_u32x4 *fb, float *out) const
  {
    dm_f32x4 c0[NT], c1[NT];
    dm_s16x4 xb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
      c0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dm_bf16x8, a0), __builtin_bit_cast(dm_bf16x8, fb[t]),
                                                       dm_f32x4{b0v[0], b0v[1], b0v[2], b0v[3]}, 0, 0, 0);
    // relu + bf16: accumulator rows 4g..4g+3 of pixel `col` == B operand k-slots 4g..4g+3 of column `col`
#pragma unroll
    for (int t = 0; t < NT; ++t) xb[t] = dm_relu_packed(pack_bf16x4(c0[t][0], c0[t][1], c0[t][2], c0[t][3]));
#pragma unroll
    for (int t = 0; t < NT; ++t)
      c1[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, xb[t], dm_f32x4{b1v[0], b1v[1], b1v[2], b1v[3]}, 0, 0, 0);
#if PCT_DMH_L2_MFMA
    // layer 2 on the matrix cores too (per tile: 2 conversions + 2 packed maxima + 1 MFMA instead of 4 maxima, 4 FMAs, a lane
    // swap and 2 additions -- the kernel is bound by vector issue, its matrix pipe 18 % busy); its input is rounded to bf16 as
    // the reference's autocast convolution rounds it
    dm_s16x4 xc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) xc[t] = dm_relu_packed(pack_bf16x4(c1[t][0], c1[t][1], c1[t][2], c1[t][3]));
#pragma unroll
    for (int t = 0; t < NT; ++t)
      out[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2, xc[t], dm_f32x4{b2, b2, b2, b2}, 0, 0, 0)[0];
#else
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float part = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) part = fmaf(w2v[r], dm_relu(c1[t][r]), part);
      // rows 4g..4g+3 (+) rows of the partner group (lane ^ 16): v_permlane16_swap hands the even 16-lane rows the odd
      // rows' value and the reverse; the sum of the two results is the pair's total in all four rows (a + b == b + a:
      // the same bits as the ds_bpermute form it replaces, without the trip through the LDS pipe)
      // (inline asm, with the two wait states a vector write -> v_permlane read needs inside the string: given the same
      // value in both operands, the builtin's two results are taken for equal by hipcc (ROCm 7.2) and their sum is
      // folded into 2 x one of them -- also with one operand behind an empty asm)
      float pa = part, pb = part;
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(pa), "+v"(pb));
      out[t] = (pa + pb) + b2;
    }
#endif
  }
};

}  // namespace pct
