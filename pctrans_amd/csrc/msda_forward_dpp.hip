// MSDeformAttn forward, "quad-owner" kernel for MI355X (gfx950, wave64) -- the default path for PCTrans' geometry
// (head dim * element size = 32 or 64 bytes, i.e. D = 16 fp32 / fp16 / bf16).
//
// Same semantics as msda_forward.hip (reference: ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 + :38-89).
//
// Measured fact that shapes this kernel (profiles/r01_*_pmc): the straightforward formulation is bound by VALU
// ISSUE, not by memory -- every lane re-derives every sample's bilinear geometry (~40 VALU ops x L*P samples) and
// masks out-of-map corners with per-channel selects.  So the instruction stream is cut instead:
//   * the QL lanes that share a (query, head) record split the record's points: lane c OWNS points p == c (mod QL)
//     of every level, loads only their (x, y, weight), derives the four corner offsets and the four
//     bilinear*attention weights once, and the siblings receive them through DPP quad_perm (v_mov_b32_dpp /
//     v_add_u32_dpp -- register crossbar, no LDS);
//   * value is read through a buffer descriptor (raw_buffer_load_b128, 32-bit byte offsets): a masked corner gets an
//     out-of-range offset and the hardware bounds check returns zeros -- no selects, no 64-bit address arithmetic,
//     and an Inf/NaN in an unread texel can not leak;
//   * accumulation is a chain of 4 packed FMAs per channel pair straight into the accumulator;
//   * the next level's points are fetched while the current level is gathered.
#include <math.h>

#include <utility>

#include "msda_common.hpp"

namespace pct {

constexpr int DPP_BLOCK = 256;

template <int CTRL>
__device__ __forceinline__ float qbcast_f(float v)
{
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ unsigned qbcast_u(unsigned v)
{
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, true);
}
template <int QL, int SRC>
struct QuadCtrl {   // every lane of a QL-lane group reads lane SRC of its own group
  static constexpr int value = QL == 4 ? (SRC | (SRC << 2) | (SRC << 4) | (SRC << 6))
                                       : (SRC | (SRC << 2) | ((2 + SRC) << 4) | ((2 + SRC) << 6));
};

typedef int i32x4 __attribute__((ext_vector_type(4)));

// QMAJOR: a wave holds 64/QL CONSECUTIVE queries of ONE head (block = M waves = the M heads of those queries).
// Consecutive queries are row-adjacent pixels in PCTrans' encoder, so neighbouring lanes share bilinear corners and
// the wave's working set per level shrinks ~4x -> vector-L1 hit rate up (measured: profiles/r01_*_pmc).
// !QMAJOR: records in memory order (2 queries x 8 heads per wave); used when M > 16.
// FUSED: the module front-end (ops/modules/ms_deform_attn.py:100-109) is folded in -- `loc` holds the raw
// sampling offsets, `attn` the raw attention logits, and the kernel forms
//     weights = softmax over the record's L*P logits,   location = ref[q, l] + offset / (W_l, H_l)
// itself, so the [N,Lq,M,L,P,2] sampling_locations tensor and the softmax output never exist in HBM.
template <typename T, int D, int P, bool QMAJOR, bool FUSED>
__global__ __launch_bounds__(1024) void msda_forward_dpp_kernel(
    const typename Traits<T>::store_t *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const float *__restrict__ loc, const float *__restrict__ attn, const int S,
    const int M, const int L, const int Lq, const long long total_recs, const unsigned value_bytes,
    typename Traits<T>::store_t *__restrict__ out, const float *__restrict__ ref, const long long ref_batch_stride)
{
  using ST = typename Traits<T>::store_t;
  constexpr int VEC = 16 / (int)sizeof(ST);       // channels per lane
  constexpr int QL = D / VEC;                     // lanes per record
  constexpr int PPL = P / QL;                     // points a lane owns per level
  constexpr unsigned ESZ = sizeof(ST);
  constexpr unsigned OOB = 0x80000000u;           // value_bytes < 2^31 (host-checked): always out of range
  static_assert((QL == 2 || QL == 4) && P % QL == 0, "unsupported geometry");

  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  long long rec;
  if constexpr (QMAJOR) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    rec = ((long long)lb * (64 / QL) + lane / QL) * M + wave;     // (global query) * M + head
  } else {
    rec = (long long)lb * (DPP_BLOCK / QL) + (threadIdx.x / QL);
  }
  const bool active = rec < total_recs;
  rec = active ? rec : total_recs - 1;            // idle tail lanes shadow the last record (stores are masked)
  const int c = threadIdx.x & (QL - 1);
  const int m = (int)(rec % M);
  const long long b = rec / ((long long)M * Lq);
  const unsigned MDb = (unsigned)(M * D) * ESZ;   // bytes per pixel
  const unsigned lane_base = (unsigned)((b * S) * (long long)MDb) + (unsigned)(m * D) * ESZ + (unsigned)c * 16u;

  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST *>(value), 0, (int)value_bytes, 0x00020000);

  const float *lrec = loc + rec * ((long long)L * P * 2) + c * 2;
  const float *wrec = attn + rec * ((long long)L * P) + c;

  // FUSED: softmax statistics of this record's L*P logits (each lane sees its own points; the QL lanes combine by DPP)
  float sm_max = 0.f, sm_inv = 1.f;
  const float *rrec = nullptr;
  if constexpr (FUSED) {
    const long long gq = rec / M;                                   // b * Lq + q
    rrec = ref + b * ref_batch_stride + (gq - b * Lq) * ((long long)L * 2);
    float mx = -INFINITY;
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int k = 0; k < PPL; ++k) mx = fmaxf(mx, wrec[l * P + k * QL]);
    mx = fmaxf(mx, qbcast_f<QL == 4 ? 0xB1 : 0xB1>(mx));            // lane ^ 1
    if constexpr (QL == 4) mx = fmaxf(mx, qbcast_f<0x4E>(mx));      // lane ^ 2
    float sum = 0.f;
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int k = 0; k < PPL; ++k) sum += __expf(wrec[l * P + k * QL] - mx);
    sum += qbcast_f<0xB1>(sum);
    if constexpr (QL == 4) sum += qbcast_f<0x4E>(sum);
    sm_max = mx;
    sm_inv = 1.f / sum;
  }

  float acc[VEC], acc2[VEC], acc3[VEC], acc4[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = acc2[e] = acc3[e] = acc4[e] = 0.f;

  // software pipeline over levels: my points of level l+1 are in flight while level l is gathered
  float nx[PPL], ny[PPL], nw[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const vec_t<float, 2> xy = *reinterpret_cast<const vec_t<float, 2> *>(lrec + (k * QL) * 2);
    nx[k] = xy[0];
    ny[k] = xy[1];
    nw[k] = wrec[k * QL];
  }

#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    float cx[PPL], cy[PPL], cw[PPL];
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      cx[k] = nx[k];
      cy[k] = ny[k];
      cw[k] = nw[k];
    }
    if (l + 1 < L) {
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        const vec_t<float, 2> xy =
            *reinterpret_cast<const vec_t<float, 2> *>(lrec + ((l + 1) * P + k * QL) * 2);
        nx[k] = xy[0];
        ny[k] = xy[1];
        nw[k] = wrec[(l + 1) * P + k * QL];
      }
    }
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const unsigned lvl = (unsigned)starts[l] * MDb;
    if constexpr (FUSED) {
      const float rx = rrec[2 * l], ry = rrec[2 * l + 1];
      const float iw = 1.0f / (float)W, ih = 1.0f / (float)H;      // uniform: one reciprocal per level
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        cx[k] = fmaf(cx[k], iw, rx);                                // reference_points + offsets / (W_l, H_l)
        cy[k] = fmaf(cy[k], ih, ry);
        cw[k] = __expf(cw[k] - sm_max) * sm_inv;                    // softmax weight
      }
    }

    // owner side: corner byte offsets (relative to lane_base) and bilinear * attention weights of my points
    unsigned o1[PPL], o2[PPL], o3[PPL], o4[PPL];
    float g1[PPL], g2[PPL], g3[PPL], g4[PPL];
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      const float h_im = cy[k] * H - 0.5f, w_im = cx[k] * W - 0.5f;
      const bool gate = h_im > -1 && w_im > -1 && h_im < H && w_im < W;     // false for NaN
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int y0 = gate ? (int)hf : 0, x0 = gate ? (int)wf : 0;
      const float lh = gate ? h_im - hf : 0.f, lw = gate ? w_im - wf : 0.f;  // gated-out sample contributes
      const float wgt = gate ? cw[k] : 0.f;                                  // exactly 0 (cuh:290-296)
      const float hh = 1.f - lh, hw = 1.f - lw;
      const bool top = gate && y0 >= 0, bot = gate && y0 + 1 <= H - 1;
      const bool lft = x0 >= 0, rgt = x0 + 1 <= W - 1;
      const unsigned a = lvl + (unsigned)(y0 * W + x0) * MDb;
      o1[k] = (top && lft) ? a : OOB;
      o2[k] = (top && rgt) ? a + MDb : OOB;
      o3[k] = (bot && lft) ? a + (unsigned)W * MDb : OOB;
      o4[k] = (bot && rgt) ? a + (unsigned)W * MDb + MDb : OOB;
      g1[k] = hh * hw * wgt;
      g2[k] = hh * lw * wgt;
      g3[k] = lh * hw * wgt;
      g4[k] = lh * lw * wgt;
    }

    // consumer side: all lanes walk the P points; point p's geometry comes from its owner lane through DPP
    auto consume = [&](auto pc) {
      constexpr int p = decltype(pc)::value;
      constexpr int k = p / QL;
      constexpr int ctrl = QuadCtrl<QL, p % QL>::value;
      const unsigned a1 = qbcast_u<ctrl>(o1[k]) + lane_base, a2 = qbcast_u<ctrl>(o2[k]) + lane_base;
      const unsigned a3 = qbcast_u<ctrl>(o3[k]) + lane_base, a4 = qbcast_u<ctrl>(o4[k]) + lane_base;
      const float w1 = qbcast_f<ctrl>(g1[k]), w2 = qbcast_f<ctrl>(g2[k]);
      const float w3 = qbcast_f<ctrl>(g3[k]), w4 = qbcast_f<ctrl>(g4[k]);
      const i32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)a1, 0, 0);
      const i32x4 r2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)a2, 0, 0);
      const i32x4 r3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)a3, 0, 0);
      const i32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)a4, 0, 0);
      const vec_t<ST, VEC> v1 = __builtin_bit_cast(vec_t<ST, VEC>, r1), v2 = __builtin_bit_cast(vec_t<ST, VEC>, r2);
      const vec_t<ST, VEC> v3 = __builtin_bit_cast(vec_t<ST, VEC>, r3), v4 = __builtin_bit_cast(vec_t<ST, VEC>, r4);
      // one accumulator per corner: independent FMA chains (no back-to-back dependent FMAs)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        acc[e] = fmaf(w1, Traits<T>::to_acc(v1[e]), acc[e]);
        acc2[e] = fmaf(w2, Traits<T>::to_acc(v2[e]), acc2[e]);
        acc3[e] = fmaf(w3, Traits<T>::to_acc(v3[e]), acc3[e]);
        acc4[e] = fmaf(w4, Traits<T>::to_acc(v4[e]), acc4[e]);
      }
    };
    [&]<int... Ps>(std::integer_sequence<int, Ps...>) {
      (consume(std::integral_constant<int, Ps>{}), ...);
    }(std::make_integer_sequence<int, P>{});
  }

  if (active) {
    vec_t<ST, VEC> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = Traits<T>::from_acc((acc[e] + acc2[e]) + (acc3[e] + acc4[e]));
    *reinterpret_cast<vec_t<ST, VEC> *>(out + rec * D + c * VEC) = o;
  }
}

// returns -100 when this geometry is not covered (caller falls through to the generic kernel).
// ref == nullptr: plain op (loc = sampling locations, attn = attention weights);
// ref != nullptr: fused front-end (loc = raw offsets, attn = raw logits, ref = [N or 1, Lq, L, 2] reference points).
template <typename T>
int launch_msda_forward_dpp(const void *value, const int64_t *shapes, const int64_t *starts, const void *loc,
                            const void *attn, int N, int S, int M, int D, int L, int Lq, int P, void *out,
                            hipStream_t stream, const float *ref, long long ref_batch_stride)
{
  using ST = typename Traits<T>::store_t;
  constexpr int VEC = 16 / (int)sizeof(ST);
  if (D != 16 || (P != 4 && P != 8)) return -100;
  if ((((uintptr_t)value | (uintptr_t)out) & 15u) || (((uintptr_t)loc) & 7u) || (((uintptr_t)attn) & 3u)) return -100;
  const long long vbytes = (long long)N * S * M * D * (long long)sizeof(ST);
  if (vbytes >= 0x7fffffffLL) return -100;                          // 32-bit buffer offsets, OOB sentinel at 2^31
  constexpr int QL = 16 / VEC;
  const long long total_recs = (long long)N * Lq * M;
  if (total_recs == 0) return 0;
  const bool qmajor = M <= 16;
  const long long per_blk = qmajor ? (long long)(64 / QL) * M : DPP_BLOCK / QL;    // records per block
  const long long nblk = qmajor ? ((long long)N * Lq + 64 / QL - 1) / (64 / QL) : (total_recs + per_blk - 1) / per_blk;
  if (nblk > 0x7fffffffLL) return -100;
  const dim3 grid((unsigned)nblk), block(qmajor ? 64 * M : DPP_BLOCK);
  const ST *v = static_cast<const ST *>(value);
  const float *lc = static_cast<const float *>(loc), *at = static_cast<const float *>(attn);
  ST *o = static_cast<ST *>(out);
#define PCT_DPP(P_, QM_, FU_)                                                                                      \
  hipLaunchKernelGGL((msda_forward_dpp_kernel<T, 16, P_, QM_, FU_>), grid, block, 0, stream, v, shapes, starts, lc, \
                     at, S, M, L, Lq, total_recs, (unsigned)vbytes, o, ref, ref_batch_stride)
  if (ref) {
    if (P == 4 && qmajor) PCT_DPP(4, true, true);
    else if (P == 4) PCT_DPP(4, false, true);
    else if (qmajor) PCT_DPP(8, true, true);
    else PCT_DPP(8, false, true);
  } else {
    if (P == 4 && qmajor) PCT_DPP(4, true, false);
    else if (P == 4) PCT_DPP(4, false, false);
    else if (qmajor) PCT_DPP(8, true, false);
    else PCT_DPP(8, false, false);
  }
#undef PCT_DPP
  return (int)hipGetLastError();
}

template int launch_msda_forward_dpp<float>(const void *, const int64_t *, const int64_t *, const void *,
                                            const void *, int, int, int, int, int, int, int, void *, hipStream_t,
                                            const float *, long long);
template int launch_msda_forward_dpp<half_bits>(const void *, const int64_t *, const int64_t *, const void *,
                                                const void *, int, int, int, int, int, int, int, void *, hipStream_t,
                                                const float *, long long);
template int launch_msda_forward_dpp<bf16_bits>(const void *, const int64_t *, const int64_t *, const void *,
                                                const void *, int, int, int, int, int, int, int, void *, hipStream_t,
                                                const float *, long long);

}  // namespace pct
