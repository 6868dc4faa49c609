// MSDeformAttn backward, windowed-LDS accumulation for MI355X (gfx950, wave64).
//
// Same semantics as msda_backward.hip (reference: ops/src/cuda/ms_deform_im2col_cuda.cuh:92-164 per sample, channel
// sums as in cuh:306-408).
//
// Why: the reference -- and msda_backward.hip -- scatter grad_value with one global float atomic per (sample, corner,
// channel): L*P*4*D = 1024 atomics per (query, head), each wave instruction touching 16 different 64-B lines four
// bytes at a time.  Measured on MI355X that path sustains 0.33 TB/s of atomic bytes and is >95 % of the backward's
// time.  Neighbouring queries sample neighbouring texels (PCTrans' encoder self-attention: Lq == S, offsets of a few
// pixels), so here, with the work item = (image, query tile, head) of msda_forward_win.hip:
//
//   * the value window of each level is staged in LDS exactly as in the forward (LDS-DMA, zero apron) and a second,
//     zero-initialised window of the same geometry accumulates grad_value.  LDS *float* atomics are unusable for this
//     (ds_add_f32: 193 cycles per wave instruction per CU measured, any layout; ds_add_u32: 4.3), so the window holds
//     int32 fixed point with one power-of-two scale per tile, derived from max|grad_out| * max|attn| over the tile
//     (resolution 2^-21 of that bound, no overflow possible, order-independent sums); channel-planar so that the
//     32-lane groups of every ds_add_u32 fall on 32 different banks;
//   * when the tile's samples are done the grad window is flushed once: one global atomic per window dword, lanes on
//     consecutive dwords (a wave instruction covers four whole 64-B head-pixels).  A texel shared by k samples of the
//     tile costs one global atomic instead of k, and none of them is a partial line;
//   * grad_sampling_loc / grad_attn_weight: the three channel sums are DPP butterflies inside the query's 4-lane
//     group; the lane that owns point p keeps them and stores 8 + 4 bytes per level, coalesced across the group;
//   * a level whose window does not fit (coarse-level tiles looking at a fine level, scattered locations) uses the
//     direct path for that level only: global gathers + global atomics, with the same owner-computed geometry.
//
// fp32, D == 16, P == 4 (the shapes PCTrans runs); anything else goes to msda_backward.hip.
#include <math.h>
#include <stdlib.h>

#include <utility>

#include "msda_win_common.hpp"

namespace pct {

// pixels per pool: NS = 1 -> 3 workgroups per CU (3 x 53.8 KB), NS = 2 -> 2 per CU (2 x 74.2 KB)
__host__ __device__ constexpr int bwd_win_pool_px(int ns) { return ns == 1 ? 418 : 578; }

template <int L, int NS>
__global__ __launch_bounds__(WIN_BLOCK, 2) void msda_backward_win_kernel(
    const float *__restrict__ grad_out, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const float *__restrict__ loc, const float *__restrict__ attn, const int N,
    const int S, const int M, const int Lq, const int pyramid, float *__restrict__ grad_value,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn, unsigned *__restrict__ queue)
{
  constexpr int D = 16, P = 4, VEC = 4, QL = 4;
  constexpr int TQ = WIN_BLOCK / QL;              // queries per slot (64)
  constexpr int TW = TQ / WIN_TH;                 // 8 pixels wide
  constexpr int SH = TQ / TW;                     // 8 rows per slot
  constexpr int THT = SH * NS;                    // tile height
  constexpr int PXB = QL * 16;                    // bytes per head-pixel
  // Window capacity in pixels.  The grad windows are channel-planar: plane ch holds GPX floats, so that the lanes of
  // one ds_add_u32 (32-lane groups = 8 queries x 4 channel quarters, banks = dword mod 32) fall on 8*c + pixel, and
  // the lanes of the flush's ds_read_b32 (2 pixels x 16 channels) on 2*ch + pixel: GPX = 2 (mod 32).
  constexpr int GPX = bwd_win_pool_px(NS);
  constexpr int pool_px = GPX;
  static_assert(GPX % 32 == 2, "plane stride must keep the LDS atomics conflict-free");
  static_assert(L <= WIN_MAXL, "too many levels");
  using v4f = vec_t<float, 4>;
  typedef float f32x2 __attribute__((ext_vector_type(2)));

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char *vpool = smem_raw;                                  // value windows
  constexpr unsigned gofs = (unsigned)GPX * PXB;                   // grad windows: 16 planes of GPX floats
  int *gpool = reinterpret_cast<int *>(smem_raw + gofs);           // fixed-point accumulators (see `scale` below)
  unsigned *bb = reinterpret_cast<unsigned *>(smem_raw + 2 * (size_t)gofs);   // [4 waves][L][2]
  unsigned *mx = bb + (WIN_BLOCK / 64) * WIN_MAXL * 2;              // [4 waves][2]: max |grad_out| bits, max |attn| bits
  unsigned *next_idx = mx + (WIN_BLOCK / 64) * 2;                   // the workgroup's next item (index in the XCD's chunk)

  const int tid = threadIdx.x, wave = tid >> 6;
  const int c = tid & (QL - 1);
  const int j = tid / QL;
  const int MD = M * D;

  int Hs[L], Ws[L], St[L], tiles_before[L + 1];
  tiles_before[0] = 0;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
    St[l] = (int)starts[l];
    tiles_before[l + 1] = tiles_before[l] + ((Hs[l] + THT - 1) / THT) * ((Ws[l] + TW - 1) / TW);
  }
  const int T_img = pyramid ? tiles_before[L] : (Lq + TQ * NS - 1) / (TQ * NS);
  const int items = N * T_img * M;

  // pixels 0 and 1 of both pools serve gated-out samples: they read zeros and add zeros (never flushed)
  if (tid < 2 * QL) reinterpret_cast<v4f *>(vpool)[tid] = v4f{0.f, 0.f, 0.f, 0.f};

  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int chunk = (items + 7) / 8;
  const int item_end = min((xcd + 1) * chunk, items);

  // items: first round static, then from the XCD's self-resetting counter (msda_win_common.hpp: win_queue_slot)
  int item = xcd * chunk + slot0;
  while (item < item_end) {
    unsigned rfetch = 0u;                                            // looked at right before barrier (1)
    if (queue && tid == 0) rfetch = atomicAdd(queue + xcd * WIN_QUEUE_STRIDE, 1u);
    const int m = item % M;
    const int bt = item / M;
    const int t = bt % T_img;
    const int b = bt / T_img;

    // ---- this lane's NS queries (record index or -1) -----------------------------------------------------------
    long long recs[NS];
    {
      int Hq = Hs[0], Wq = Ws[0], Sq = St[0], tb = 0;
      if (pyramid) {
#pragma unroll
        for (int l = 1; l < L; ++l)
          if (t >= tiles_before[l]) { Hq = Hs[l]; Wq = Ws[l]; Sq = St[l]; tb = tiles_before[l]; }
      }
      const int tpr = (Wq + TW - 1) / TW;
      const int tl = t - tb;
      const int ty = tl / tpr, tx = tl - ty * tpr;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        int q;
        bool ok;
        if (pyramid) {
          const int qy = ty * THT + s * SH + j / TW, qx = tx * TW + (j % TW);
          ok = qy < Hq && qx < Wq;
          q = Sq + qy * Wq + qx;
        } else {
          q = (t * NS + s) * TQ + j;
          ok = q < Lq;
        }
        recs[s] = ok ? ((long long)b * Lq + q) * M + m : -1;
      }
    }

    // ---- pre-pass: per-level bounding box (incl. 1-pixel apron) of every corner the tile touches ---------------
    f32x2 pxy[NS][L];                                               // my point (p == c) of every level, kept
    float pw[NS][L];
    v4f tops[NS];                                                   // my 4 channels of grad_out
    {
      unsigned gmax = 0u, amax = 0u;                                // |x| as bits: Inf / NaN compare largest
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const long long r = recs[s] < 0 ? 0 : recs[s];
        const float *lrec = loc + r * (L * P * 2) + c * 2;
        const float *wrec = attn + r * (L * P) + c;
#pragma unroll
        for (int l = 0; l < L; ++l) {
          pxy[s][l] = *reinterpret_cast<const f32x2 *>(lrec + l * P * 2);
          pw[s][l] = wrec[l * P];
        }
        tops[s] = *reinterpret_cast<const v4f *>(grad_out + r * D + c * VEC);
        if (recs[s] < 0) tops[s] = v4f{0.f, 0.f, 0.f, 0.f};
        else {
#pragma unroll
          for (int l = 0; l < L; ++l) amax = max(amax, __float_as_uint(pw[s][l]) & 0x7fffffffu);
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float tk = tops[s][k];
          gmax = max(gmax, __float_as_uint(tk) & 0x7fffffffu);
        }
      }
      gmax = wave_reduce_umax(gmax);
      amax = wave_reduce_umax(amax);
      if ((tid & 63) == 0) {
        mx[wave * 2] = gmax;
        mx[wave * 2 + 1] = amax;
      }
#pragma unroll
      for (int l = 0; l < L; ++l) {
        unsigned lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float h_im = pxy[s][l][1] * Hs[l] - 0.5f, w_im = pxy[s][l][0] * Ws[l] - 0.5f;
          const bool gate = recs[s] >= 0 && h_im > -1 && w_im > -1 && h_im < Hs[l] && w_im < Ws[l];
          const unsigned xa = (unsigned)((int)floorf(w_im) + 1), ya = (unsigned)((int)floorf(h_im) + 1);
          lo = gate ? pk_min(lo, xa | (ya << 16)) : lo;
          hi = gate ? pk_max(hi, (xa + 1) | ((ya + 1) << 16)) : hi;
        }
        lo = wave_reduce_pk<true>(lo);
        hi = wave_reduce_pk<false>(hi);
        if ((tid & 63) == 0) {
          bb[(wave * L + l) * 2] = lo;
          bb[(wave * L + l) * 2 + 1] = hi;
        }
      }
    }
    if (tid == 0) {
      unsigned fetched = (unsigned)(item - xcd * chunk + nslots);   // static stride when there is no queue
      if (queue) {
        if (rfetch + 1u >= (unsigned)(item_end - xcd * chunk)) atomicExch(queue + xcd * WIN_QUEUE_STRIDE, 0u);
        fetched = (unsigned)nslots + rfetch;
      }
      next_idx[0] = fetched;
    }
    __syncthreads();                                               // (1) boxes visible; previous item's flush done
    int next_item;
    {
      const unsigned nxt = __builtin_amdgcn_readfirstlane(next_idx[0]);   // (rewritten only after everyone passed (2))
      next_item = nxt < (unsigned)(item_end - xcd * chunk) ? xcd * chunk + (int)nxt : item_end;
    }

    // ---- fixed-point scale of the LDS accumulators ------------------------------------------------------------
    // A window dword receives at most NS*64*P <= 512 contributions w_corner * attn * grad_out, each bounded by
    // C = max|attn| * max|grad_out| over the tile (bilinear weights are <= 1).  With C < 2^e every contribution is
    // rounded to a multiple of 2^(e-21) -- i.e. to 2^-21..2^-20 of C, the resolution an fp32 sum of that size has
    // anyway -- and the int32 sum cannot overflow (512 * 2^21 = 2^30).  Integer sums are order-independent, so the
    // LDS stage is deterministic.  Non-finite or huge bounds take the direct (float) path for the whole tile.
    bool fixed_ok;
    float scale, inv_scale;
    {
      unsigned gb = 0u, ab = 0u;
#pragma unroll
      for (int w = 0; w < WIN_BLOCK / 64; ++w) {
        gb = max(gb, mx[w * 2]);
        ab = max(ab, mx[w * 2 + 1]);
      }
      gb = __builtin_amdgcn_readfirstlane(gb);
      ab = __builtin_amdgcn_readfirstlane(ab);
      const float C = __builtin_bit_cast(float, gb) * __builtin_bit_cast(float, ab);
      int e;
      (void)frexpf(C, &e);                                          // C < 2^e
      fixed_ok = gb < 0x7f800000u && ab < 0x7f800000u && C < 1e30f;
      const int shift = min(21 - e, 100);
      scale = ldexpf(1.f, shift);
      inv_scale = ldexpf(1.f, -shift);
    }

    // ---- windows (identical in every lane; kept in SGPRs) -------------------------------------------------------
    int wx0[L], wy0[L], wwid[L], wbase[L], wsize[L], in_lds[L];
    int used = 0;
#pragma unroll
    for (int ll = 0; ll < L; ++ll) {
      const int l = L - 1 - ll;                                     // finest level first, as in the forward
      const LevelWindow w = read_window(bb, L, l);
      wx0[l] = w.x0;
      wy0[l] = w.y0;
      wwid[l] = w.wid;
      wsize[l] = w.size;
      in_lds[l] = (fixed_ok && used + w.size <= pool_px - 2) ? 1 : 0;
      wbase[l] = used + 2;
      used += in_lds[l] ? w.size : 0;
    }

    // ---- stage the value windows by LDS-DMA and zero the grad windows ------------------------------------------
    const long long img_off = (long long)b * S * MD + m * D;        // this image, this head
    const float *vimg = value + img_off;
    float *gimg = grad_value + img_off;
    auto stage_window = [&](const int H, const int W, const int st, const int wb, const int wx, const int wy,
                            const int ww, const int wsz) {
      const float inv_w = 1.0f / (float)ww;
      const int n16 = wsz * QL;
      const float *vlev = vimg + (long long)st * MD;
      unsigned char *dst = vpool + (size_t)wb * PXB;
      for (int it = 0; it * WIN_BLOCK < n16; ++it) {
        const int i = it * WIN_BLOCK + tid;
        if (i < n16) {
          const int px = i / QL, cc = i & (QL - 1);
          const int r = (int)(((float)px + 0.5f) * inv_w);
          const int y = wy + r, x = wx + px - r * ww;
          const bool inside = y >= 0 && y < H && x >= 0 && x < W;
          const float *src = inside ? vlev + (long long)(y * W + x) * MD + cc * VEC : g_zero16;
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void *)(src),
              (__attribute__((address_space(3))) void *)(dst + (size_t)(it * WIN_BLOCK + (tid & ~63)) * 16), 16, 0, 0);
        }
      }
    };
    auto zero_grad_pool = [&]() {
      for (int i = tid; i < GPX * 4; i += WIN_BLOCK)               // all 16 planes, 16 bytes at a time
        reinterpret_cast<v4f *>(gpool)[i] = v4f{0.f, 0.f, 0.f, 0.f};
    };
    // one global atomic per window dword, lanes on consecutive dwords
    auto flush_window = [&](const int H, const int W, const int st, const int wb, const int wx, const int wy,
                            const int ww, const int wsz) {
      const float inv_w = 1.0f / (float)ww;
      const int n4 = wsz * (PXB / 4);
      const int *srcw = gpool + wb;
      float *glev = gimg + (long long)st * MD;
      for (int i = tid; i < n4; i += WIN_BLOCK) {
        const int px = i >> 4, ch = i & 15;
        const int r = (int)(((float)px + 0.5f) * inv_w);
        const int y = wy + r, x = wx + px - r * ww;
        const bool inside = y >= 0 && y < H && x >= 0 && x < W;
        const int gi = srcw[ch * GPX + px];
        const float g = (float)gi * inv_scale;
#if defined(PCT_BWD_KO_FLUSH)      // timing experiments only (wrong results)
        if (inside && gi == 0x12345678) unsafeAtomicAdd(glev + (long long)(y * W + x) * MD + ch, g);
#elif defined(PCT_BWD_PLAIN_FLUSH)
        if (inside && gi != 0) glev[(long long)(y * W + x) * MD + ch] = g;
#else
        if (inside && gi != 0) unsafeAtomicAdd(glev + (long long)(y * W + x) * MD + ch, g);
#endif
      }
    };
#pragma unroll
    for (int l = 0; l < L; ++l)
      if (in_lds[l] && wsize[l] > 0) stage_window(Hs[l], Ws[l], St[l], wbase[l], wx0[l], wy0[l], wwid[l], wsize[l]);
    zero_grad_pool();
    __syncthreads();                                               // (2) windows staged / zeroed

    // ---- per slot: every lane walks the L*P samples of its query, geometry from the owner lane by DPP ----------
    const unsigned char *vlane = vpool + c * 16;
    int *glane = gpool + c * 4 * GPX;
    const f32x2 sc2 = {scale, scale};
    const f32x2 magic2 = {12582912.f, 12582912.f};                  // 1.5 * 2^23: fma(x, 1, magic) rounds x to an integer
    // slot s of this lane: record, its point (p == c) of every level, its 4 channels of grad_out
    auto select_slot = [&](const int s, long long &rec, f32x2 (&sxy)[L], float (&sw)[L], f32x2 (&top)[2]) {
      rec = recs[0];
      v4f tg = tops[0];
#pragma unroll
      for (int l = 0; l < L; ++l) {
        sxy[l] = pxy[0][l];
        sw[l] = pw[0][l];
      }
#pragma unroll
      for (int u = 1; u < NS; ++u) {
        rec = s == u ? recs[u] : rec;
#pragma unroll
        for (int l = 0; l < L; ++l) {
          sxy[l] = s == u ? pxy[u][l] : sxy[l];
          sw[l] = s == u ? pw[u][l] : sw[l];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) tg[k] = s == u ? tops[u][k] : tg[k];
      }
      top[0] = f32x2{tg[0], tg[1]};
      top[1] = f32x2{tg[2], tg[3]};
    };
    // all P samples of one level of one slot.  lds: the level's window (base wb, origin wx/wy, width ww) is staged
    auto do_level = [&](const int l, const int H, const int W, const int st, const long long rec, const f32x2 pt,
                        const float wt, const f32x2 (&top)[2], const bool lds, const int wb, const int wx,
                        const int wy, const int ww) {
        const bool qvalid = rec >= 0;
        // owner side: my point of this level
        const float h_im = pt[1] * H - 0.5f, w_im = pt[0] * W - 0.5f;
        const bool gate = qvalid && h_im > -1 && w_im > -1 && h_im < H && w_im < W;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int y0 = gate ? (int)hf : 0, x0 = gate ? (int)wf : 0;
        const float o_lh = gate ? h_im - hf : 0.f, o_lw = gate ? w_im - wf : 0.f;   // gated-out: zero gradients
        const float o_wgt = gate ? wt : 0.f;
        int o1, o2, o3 = 0, o4 = 0;
        if (lds) {
          o1 = gate ? wb + __mul24(y0 - wy, ww) + (x0 - wx) : 0;                      // pixel index in the pool
          o2 = gate ? ww : 0;                                                      // row step (pixels)
        } else {
          // element offsets from this image/head base; out-of-map corners -> -1
          const bool tp = gate && y0 >= 0, bt2 = gate && y0 + 1 <= H - 1;
          const bool lf = x0 >= 0, rg = x0 + 1 <= W - 1;
          const int a = (st + y0 * W + x0) * MD;
          o1 = (tp && lf) ? a : -1;
          o2 = (tp && rg) ? a + MD : -1;
          o3 = (bt2 && lf) ? a + W * MD : -1;
          o4 = (bt2 && rg) ? a + W * MD + MD : -1;
        }
        float my_gw = 0.f, my_gh = 0.f, my_ga = 0.f;                 // results of the point I own

        auto consume = [&](auto pc, auto lds_c) {
          constexpr int p = decltype(pc)::value;
          constexpr bool LDS = decltype(lds_c)::value;
          constexpr int ctrl = BcastCtrl<QL, p>::value;
          const int a1 = dpp_i<ctrl>(o1);
          const int a2 = dpp_i<ctrl>(o2);
          int a3 = 0, a4 = 0;
          const float lh = dpp_f<ctrl>(o_lh), lw = dpp_f<ctrl>(o_lw), wgt = dpp_f<ctrl>(o_wgt);
          const float hh = 1.f - lh, hw = 1.f - lw;
          const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
          v4f v1, v2, v3, v4;
          if constexpr (LDS) {
            const unsigned char *pa = vlane + (unsigned)a1 * PXB, *pb = pa + (unsigned)a2 * PXB;
            v1 = *reinterpret_cast<const v4f *>(pa);
            v2 = *reinterpret_cast<const v4f *>(pa + PXB);
            v3 = *reinterpret_cast<const v4f *>(pb);
            v4 = *reinterpret_cast<const v4f *>(pb + PXB);
          } else {
            a3 = dpp_i<ctrl>(o3);
            a4 = dpp_i<ctrl>(o4);
            const float *vl = vimg + c * VEC;
            v1 = *reinterpret_cast<const v4f *>(vl + max(a1, 0));
            v2 = *reinterpret_cast<const v4f *>(vl + max(a2, 0));
            v3 = *reinterpret_cast<const v4f *>(vl + max(a3, 0));
            v4 = *reinterpret_cast<const v4f *>(vl + max(a4, 0));
            if (a1 < 0) v1 = v4f{0.f, 0.f, 0.f, 0.f};
            if (a2 < 0) v2 = v4f{0.f, 0.f, 0.f, 0.f};
            if (a3 < 0) v3 = v4f{0.f, 0.f, 0.f, 0.f};
            if (a4 < 0) v4 = v4f{0.f, 0.f, 0.f, 0.f};
          }
          const f32x2 ww1 = {w1, w1}, ww2 = {w2, w2}, ww3 = {w3, w3}, ww4 = {w4, w4};
          const f32x2 hh2 = {hh, hh}, hw2 = {hw, hw}, lh2 = {lh, lh}, lw2 = {lw, lw}, wg2 = {wgt, wgt};
          f32x2 s_w = {0.f, 0.f}, s_h = {0.f, 0.f}, s_a = {0.f, 0.f};
          f32x2 add1[2], add2[2], add3[2], add4[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const f32x2 x1 = {v1[2 * e], v1[2 * e + 1]}, x2 = {v2[2 * e], v2[2 * e + 1]};
            const f32x2 x3 = {v3[2 * e], v3[2 * e + 1]}, x4 = {v4[2 * e], v4[2 * e + 1]};
            const f32x2 tgv = top[e] * wg2;                         // top_grad_value (cuh:112)
            // d bilinear / d h, d w in the reference's summation order (cuh:115-146)
            f32x2 gh = -(hw2 * x1);
            gh = __builtin_elementwise_fma(-lw2, x2, gh);
            gh = __builtin_elementwise_fma(hw2, x3, gh);
            gh = __builtin_elementwise_fma(lw2, x4, gh);
            f32x2 gw = -(hh2 * x1);
            gw = __builtin_elementwise_fma(hh2, x2, gw);
            gw = __builtin_elementwise_fma(-lh2, x3, gw);
            gw = __builtin_elementwise_fma(lh2, x4, gw);
            f32x2 val = ww1 * x1;
            val = __builtin_elementwise_fma(ww2, x2, val);
            val = __builtin_elementwise_fma(ww3, x3, val);
            val = __builtin_elementwise_fma(ww4, x4, val);
            s_a = __builtin_elementwise_fma(top[e], val, s_a);
            s_w = __builtin_elementwise_fma(gw, tgv, s_w);
            s_h = __builtin_elementwise_fma(gh, tgv, s_h);
            if constexpr (LDS) {   // round(w * tgv * scale) sits in the low mantissa bits of the sum with `magic`
              const f32x2 tgs = tgv * sc2;
              add1[e] = __builtin_elementwise_fma(ww1, tgs, magic2);
              add2[e] = __builtin_elementwise_fma(ww2, tgs, magic2);
              add3[e] = __builtin_elementwise_fma(ww3, tgs, magic2);
              add4[e] = __builtin_elementwise_fma(ww4, tgs, magic2);
            } else {
              add1[e] = ww1 * tgv;
              add2[e] = ww2 * tgv;
              add3[e] = ww3 * tgv;
              add4[e] = ww4 * tgv;
            }
          }
          if constexpr (LDS) {
            // channel-planar grad window: plane (4c + k) at gpool + (4c + k) * GPX, same pixel indexing
            int *ga_ = glane + a1;
            int *gb_ = ga_ + a2;
            constexpr int MAGIC_BITS = 0x4B400000;
            // (element reads go through scalars: __builtin_bit_cast on an ext-vector element reads element 0)
            auto fx = [](float f) { return __float_as_int(f) - MAGIC_BITS; };
#if defined(PCT_BWD_KO_LDSADD)
            if (a1 == 0x7fffffff)
#endif
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const float p0 = add1[e].x, p1 = add1[e].y, q0 = add2[e].x, q1 = add2[e].y;
              const float r0 = add3[e].x, r1 = add3[e].y, t0 = add4[e].x, t1 = add4[e].y;
              lds_add(ga_ + (2 * e) * GPX, fx(p0));
              lds_add(ga_ + (2 * e) * GPX + 1, fx(q0));
              lds_add(gb_ + (2 * e) * GPX, fx(r0));
              lds_add(gb_ + (2 * e) * GPX + 1, fx(t0));
              lds_add(ga_ + (2 * e + 1) * GPX, fx(p1));
              lds_add(ga_ + (2 * e + 1) * GPX + 1, fx(q1));
              lds_add(gb_ + (2 * e + 1) * GPX, fx(r1));
              lds_add(gb_ + (2 * e + 1) * GPX + 1, fx(t1));
            }
          } else {
            float *gl_ = gimg + c * VEC;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                if (a1 >= 0) unsafeAtomicAdd(gl_ + a1 + 2 * e + h, add1[e][h]);
                if (a2 >= 0) unsafeAtomicAdd(gl_ + a2 + 2 * e + h, add2[e][h]);
                if (a3 >= 0) unsafeAtomicAdd(gl_ + a3 + 2 * e + h, add3[e][h]);
                if (a4 >= 0) unsafeAtomicAdd(gl_ + a4 + 2 * e + h, add4[e][h]);
              }
          }
          // channel sums over the 4 lanes of the group; the owner of point p keeps them
          float g_w = (s_w[0] + s_w[1]) * (float)W, g_h = (s_h[0] + s_h[1]) * (float)H, g_a = s_a[0] + s_a[1];
          g_w += dpp_f<0xB1>(g_w);
          g_h += dpp_f<0xB1>(g_h);
          g_a += dpp_f<0xB1>(g_a);
          g_w += dpp_f<0x4E>(g_w);
          g_h += dpp_f<0x4E>(g_h);
          g_a += dpp_f<0x4E>(g_a);
          my_gw = c == p ? g_w : my_gw;
          my_gh = c == p ? g_h : my_gh;
          my_ga = c == p ? g_a : my_ga;
        };
        if (lds) {
          [&]<int... Ps>(std::integer_sequence<int, Ps...>) {
            (consume(std::integral_constant<int, Ps>{}, std::true_type{}), ...);
          }(std::make_integer_sequence<int, P>{});
        } else {
          [&]<int... Ps>(std::integer_sequence<int, Ps...>) {
            (consume(std::integral_constant<int, Ps>{}, std::false_type{}), ...);
          }(std::make_integer_sequence<int, P>{});
        }
        if (qvalid) {
          // a gated-out sample is skipped by the reference (cuh:352): exact zeros even when grad_out is not finite
          *reinterpret_cast<f32x2 *>(grad_loc + (rec * (L * P) + l * P + c) * 2) =
              gate ? f32x2{my_gw, my_gh} : f32x2{0.f, 0.f};
          grad_attn[rec * (L * P) + l * P + c] = gate ? my_ga : 0.f;
        }
    };

    // a level whose tile-wide window did not fit is retried slot by slot (a slot's 64 queries cover half / a quarter
    // of the footprint) once the pool is free again; only what still does not fit takes the direct path
    bool defer[L];
#pragma unroll
    for (int l = 0; l < L; ++l)
      defer[l] = fixed_ok && !in_lds[l] && wsize[l] > 0 && wsize[l] <= 2 * NS * (pool_px - 2);   // else hopeless

    // ---- main pass: per slot, every lane walks the samples of its query, geometry from the owner lane by DPP ----
#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
      long long rec;
      f32x2 sxy[L], top[2];
      float sw[L];
      select_slot(s, rec, sxy, sw, top);
#pragma unroll
      for (int l = 0; l < L; ++l)
        if (!defer[l])
          do_level(l, Hs[l], Ws[l], St[l], rec, sxy[l], sw[l], top, in_lds[l] != 0, wbase[l], wx0[l], wy0[l], wwid[l]);
    }
    __syncthreads();                                               // (3) every add of this tile is in the windows
#pragma unroll
    for (int l = 0; l < L; ++l)
      if (in_lds[l] && wsize[l] > 0) flush_window(Hs[l], Ws[l], St[l], wbase[l], wx0[l], wy0[l], wwid[l], wsize[l]);

    // ---- deferred levels, one slot at a time ---------------------------------------------------------------------
#pragma unroll
    for (int l = 0; l < L; ++l) {
      if (defer[l]) {                                               // uniform
#pragma unroll 1
        for (int s = 0; s < NS; ++s) {
          long long rec;
          f32x2 sxy[L], top[2];
          float sw[L];
          select_slot(s, rec, sxy, sw, top);
          {
            const float h_im = sxy[l][1] * Hs[l] - 0.5f, w_im = sxy[l][0] * Ws[l] - 0.5f;
            const bool gate = rec >= 0 && h_im > -1 && w_im > -1 && h_im < Hs[l] && w_im < Ws[l];
            const unsigned xa = (unsigned)((int)floorf(w_im) + 1), ya = (unsigned)((int)floorf(h_im) + 1);
            const unsigned lo = wave_reduce_pk<true>(gate ? (xa | (ya << 16)) : 0xFFFFFFFFu);
            const unsigned hi = wave_reduce_pk<false>(gate ? ((xa + 1) | ((ya + 1) << 16)) : 0u);
            if ((tid & 63) == 0) {
              bb[(wave * L + l) * 2] = lo;
              bb[(wave * L + l) * 2 + 1] = hi;
            }
          }
          __syncthreads();                                         // slot boxes visible; earlier flushes issued
          const LevelWindow w = read_window(bb, L, l);
          __syncthreads();                                         // everyone has read the boxes
          const bool fits = w.size > 0 && w.size <= pool_px - 2;
          if (fits) {
            stage_window(Hs[l], Ws[l], St[l], 2, w.x0, w.y0, w.wid, w.size);
            zero_grad_pool();
            __syncthreads();
          }
          do_level(l, Hs[l], Ws[l], St[l], rec, sxy[l], sw[l], top, fits, 2, w.x0, w.y0, w.wid);
          if (fits) {
            __syncthreads();
            flush_window(Hs[l], Ws[l], St[l], 2, w.x0, w.y0, w.wid, w.size);
          }
        }
      }
    }
    // no barrier here: the next item's barrier (1) orders this flush before its windows are zeroed
    item = next_item;
  }
}

// returns -100 when this geometry is not covered (caller uses msda_backward.hip)
int launch_msda_backward_win(const float *value, const int64_t *shapes, const int64_t *starts, const float *loc,
                             const float *attn, const float *grad_out, int N, int S, int M, int D, int L, int Lq,
                             int P, float *grad_value, float *grad_loc, float *grad_attn, hipStream_t stream)
{
  if ((((uintptr_t)value | (uintptr_t)grad_out) & 15u) || (((uintptr_t)loc | (uintptr_t)grad_loc) & 7u)) return -100;
  if (D != 16 || P != 4 || L < 3 || L > 5) return -100;
  if ((long long)N * Lq * M < 32768) return -100;
  if ((long long)N * ((long long)S + 64 * L) * M >= 0x7fffffffLL) return -100;
  if ((long long)S * M * D >= 0x7fffffffLL) return -100;            // per-image element offsets are 32-bit
  static const int ns_env = [] { const char *e = getenv("PCT_BWD_NS"); return e ? atoi(e) : 2; }();
  const int NS = ns_env == 1 ? 1 : 2;
  const int pool_px = bwd_win_pool_px(NS);                           // value pool + grad pool + boxes per workgroup
  const size_t lds = 2 * (size_t)pool_px * 64 + (WIN_BLOCK / 64) * (WIN_MAXL + 1) * 2 * sizeof(unsigned) + 16;
  const int wg_fit = (int)((160 * 1024) / lds);
  const int wg_per_cu = wg_fit < 1 ? 1 : (wg_fit > 3 ? 3 : wg_fit);
  const int pyramid = Lq == S ? 1 : 0;
  const dim3 grid(256 * wg_per_cu), block(WIN_BLOCK);
  unsigned *queue = win_queue_slot(stream);                        // nullptr: static item stride
#define PCT_BWIN(L_, NS_)                                                                                          \
  hipLaunchKernelGGL((msda_backward_win_kernel<L_, NS_>), grid, block, lds, stream, grad_out, value, shapes, starts, \
                     loc, attn, N, S, M, Lq, pyramid, grad_value, grad_loc, grad_attn, queue)
#define PCT_BWIN_L(NS_)              \
  if (L == 3) PCT_BWIN(3, NS_);      \
  else if (L == 4) PCT_BWIN(4, NS_); \
  else PCT_BWIN(5, NS_)
  if (NS == 1) { PCT_BWIN_L(1); }
  else { PCT_BWIN_L(2); }
#undef PCT_BWIN_L
#undef PCT_BWIN
  return (int)hipGetLastError();
}

}  // namespace pct
