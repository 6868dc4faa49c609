// MSDeformAttn forward, "pyramid-column" kernel for MI355X (gfx950, wave64): the default for PCTrans' encoder
// self-attention (Lq == S, fp32, D = 16, P = 4) once the problem fills the chip.
//
// Same semantics as msda_forward.hip (reference: ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 + :38-89).
//
// What it changes against the windowed kernel of round 1 (msda_forward_win.hip: 16 x 16 tile of ONE level x one head per
// item, 4 lanes per (query, head), 0.33 of the HBM roofline with 1.87x the algorithmic bytes on the memory side and
// ~2 600 vector instructions per wave and item):
//   * work item = (image, pyramid COLUMN, head): the queries of ALL levels whose pixel centres fall into one cell of a
//     CX x CY grid over the image -- e.g. 32 x 22 pixels of the 128^2 level + 16 x 11 + 8 x 6 + 4 x 3 of the coarser
//     ones, <= 1024 queries.  Every query of the column samples, on each level, around the same spatial cell, so one
//     window per level serves the whole column: ~1.9 staged head-pixels per query instead of 3.3 (fine tiles re-staging
//     the coarse levels), and a coarse-level query looking at a fine level needs no window of its own;
//   * one lane = one (query, head): it loads its whole 128-byte sampling-location record and 64-byte weight record
//     (nothing is shared between lanes, so no DPP broadcasts and no owner/consumer split), derives each sample's
//     geometry once and accumulates all D = 16 channels -- ~45 % of the vector instructions per sample;
//   * a 64-byte head-pixel is read as four ds_read_b128 whose piece order is ROTATED per 8-lane block
//     (piece (j + lane/8) % 4 in instruction j): the 16 lanes the LDS serves together would otherwise meet on 4 of
//     the 16 bank groups (pixels are 64 B apart) -- 4-way conflicts; the rotation only permutes which accumulator
//     register holds which 4 channels, which the final store undoes with address arithmetic;
//   * one workgroup of 1024 threads per CU with a pool of up to ~150 KB of LDS, windows = per-level bounding boxes of
//     the column's samples (packed-u16 min/max: DPP inside a wave, one LDS hop across the 16 waves) staged by LDS-DMA
//     with a zero apron; levels that do not fit the pool together are staged and gathered in successive PHASES
//     (accumulators stay in registers), a level whose box alone exceeds the pool is gathered from global memory
//     through a bounds-checked buffer descriptor;
//   * persistent grid, items handed out per XCD from self-resetting counters with the 8 heads of a column adjacent
//     (the heads share every 128-byte line of the column's windows, two heads per line).
#include <math.h>
#include <stdlib.h>

#include <utility>

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

template <int L, bool FUSED, int BLOCK>
__global__ __launch_bounds__(BLOCK) void msda_forward_col_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const float *__restrict__ loc, const float *__restrict__ attn, const int N, const int S, const int M,
    const int pool_px, float *__restrict__ out, const float *__restrict__ ref, const long long ref_batch_stride,
    unsigned *__restrict__ queue)
{
  constexpr int P = 4, D = 16, PXB = 64, NW = BLOCK / 64;
  static_assert(L >= 1 && L <= 5 && NW <= 16, "unsupported geometry");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char *pool = smem_raw;                                              // level windows, 64 B per head-pixel
  unsigned *bb = reinterpret_cast<unsigned *>(smem_raw + (size_t)pool_px * PXB);   // [NW][L][2] per-wave boxes
  unsigned *next_idx = bb + NW * L * 2;                                        // the workgroup's next item

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int MD = M * D;

  // ---- level geometry (uniform) -------------------------------------------------------------------------------------
  int Hs[L], Ws[L], St[L];
  float fH[L], fW[L], invW[L], invH[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
    St[l] = (int)starts[l];
    fH[l] = uni((float)Hs[l]);
    fW[l] = uni((float)Ws[l]);
    invW[l] = uni(1.0f / (float)Ws[l]);                                       // FUSED: offset / W as offset * (1 / W)
    invH[l] = uni(1.0f / (float)Hs[l]);
  }

  // ---- column grid (uniform; every workgroup derives the same one): CX x CY cells such that no column holds more than
  // BLOCK queries, cells about square on the finest level with a width near a multiple of 8 lanes ----------------------
  int CX, CY;
  {
    int Hf = Hs[0], Wf = Ws[0];
#pragma unroll
    for (int l = 1; l < L; ++l)
      if (Hs[l] * Ws[l] > Hf * Wf) { Hf = Hs[l]; Wf = Ws[l]; }
    const float area = (float)BLOCK * (float)(Hf * Wf) / (float)S;           // finest-level pixels per full column
    const int side = (int)sqrtf(area);
    const int nxt = min(Wf, max(8, (side + 7) & ~7));
    CX = (Wf + nxt - 1) / nxt;
    const int nx0 = (Wf + CX - 1) / CX;
    const int nyt = max(1, (int)(area / (float)nx0));
    CY = min(Hf, (Hf + nyt - 1) / nyt);
    for (int guard = 0; guard < 4096; ++guard) {
      int maxq = 0;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        int mx = 0, my = 0;
        for (int c = 0; c < CX; ++c) mx = max(mx, col_lo(c + 1, Ws[l], CX) - col_lo(c, Ws[l], CX));
        for (int c = 0; c < CY; ++c) my = max(my, col_lo(c + 1, Hs[l], CY) - col_lo(c, Hs[l], CY));
        maxq += mx * my;
      }
      if (maxq <= BLOCK) break;
      if (CY < Hf) ++CY;
      else if (CX < Wf) ++CX;
      else break;                                                              // (L pixels per column: cannot exceed BLOCK)
    }
  }
  const int ncol = CX * CY;
  const int items = N * ncol * M;

  // pixels 0 and 1 of the pool are zeros: gated-out samples read them (weight 0 times a guaranteed-finite value)
  if (tid < 8) reinterpret_cast<col_f32x4 *>(pool)[tid] = col_f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane rotation of the four 16-byte pieces of a head-pixel (see the header)
  const unsigned rho = (unsigned)(lane >> 3) & 3u;
  unsigned rot[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rot[j] = ((j + rho) & 3u) << 4;

  // ---- persistent, XCD-chunked walk over the items (msda_win_common.hpp: queue protocol) ---------------------------
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int chunk = (items + 7) / 8;
  const int item_end = min((xcd + 1) * chunk, items);
  int item = xcd * chunk + slot0;

  while (item < item_end) {
    // ---- the item and this lane's query ------------------------------------------------------------------------------
    const int m = item % M;
    const int bt = item / M;
    const int col = bt % ncol;
    const int b = bt / ncol;
    const int cy = col / CX, cx = col - cy * CX;

    int q = 0;
    bool valid = false;
    {
      int r = tid, q_first = 0;
      bool found = false, have_first = false;
#pragma unroll
      for (int ll = 0; ll < L; ++ll) {
        const int l = L - 1 - ll;                                             // finest level (last in PCTrans) first
        const int xa = col_lo(cx, Ws[l], CX), nx = col_lo(cx + 1, Ws[l], CX) - xa;
        const int ya = col_lo(cy, Hs[l], CY), ny = col_lo(cy + 1, Hs[l], CY) - ya;
        const int cnt = nx * ny;
        if (!have_first && cnt > 0) { have_first = true; q_first = St[l] + ya * Ws[l] + xa; }
        const bool in = !found && r < cnt;
        if (in) {
          const int ly = (int)(((float)r + 0.5f) * uni(1.0f / (float)max(nx, 1)));
          const int lx = r - ly * nx;
          q = St[l] + (ya + ly) * Ws[l] + xa + lx;
          found = true;
        }
        r -= found ? 0 : cnt;
      }
      valid = found;
      if (!found) q = q_first;            // idle lanes shadow a query of the column: they cannot move its boxes
    }
    const long long rec = ((long long)b * S + q) * M + m;

    // ---- the record: sampling locations (FUSED: reference point + offset / (W, H)) ---------------------------------
    col_f32x2 lxy[L][P];
    {
      const col_f32x4 *lp = reinterpret_cast<const col_f32x4 *>(loc + rec * (L * P * 2));
      col_f32x4 raw[L * 2];
#pragma unroll
      for (int i = 0; i < L * 2; ++i) raw[i] = lp[i];
      col_f32x2 rr[L];
      if constexpr (FUSED) {
        const float *rrow = ref + b * ref_batch_stride + (long long)q * (L * 2);
#pragma unroll
        for (int l = 0; l < L; ++l) rr[l] = *reinterpret_cast<const col_f32x2 *>(rrow + 2 * l);
      }
#pragma unroll
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < P; ++k) {
          col_f32x2 v = {raw[l * 2 + k / 2][(k & 1) * 2], raw[l * 2 + k / 2][(k & 1) * 2 + 1]};
          if constexpr (FUSED) v = col_f32x2{fmaf(v[0], invW[l], rr[l][0]), fmaf(v[1], invH[l], rr[l][1])};
          lxy[l][k] = v;
        }
    }

    // ---- pre-pass: per-level bounding box (first corners, biased by +1; the box covers x0 .. x0 + 1) ----------------
#pragma unroll
    for (int l = 0; l < L; ++l) {
      float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
#pragma unroll
      for (int k = 0; k < P; ++k) {
        // same expressions as the gather's, so both sides floor the same number.  A sample is gated in iff
        // -1 < w_im < W: clamping to [-1, W - 0.5] maps a gated-out coordinate onto one a gated-in sample could have
        // (NaN clamps to -1), so it can only widen the box towards the map border, never past the 1-pixel apron.
        const float h_im = fmaf(lxy[l][k][1], fH[l], -0.5f), w_im = fmaf(lxy[l][k][0], fW[l], -0.5f);
        const float wc = __builtin_amdgcn_fmed3f(w_im, -1.f, uni(fW[l] - 0.5f));
        const float hc = __builtin_amdgcn_fmed3f(h_im, -1.f, uni(fH[l] - 0.5f));
        mnx = fminf(mnx, wc);
        mxx = fmaxf(mxx, wc);
        mny = fminf(mny, hc);
        mxy = fmaxf(mxy, hc);
      }
      unsigned lo = (unsigned)((int)floorf(mnx) + 1) | ((unsigned)((int)floorf(mny) + 1) << 16);
      unsigned hi = (unsigned)((int)floorf(mxx) + 2) | ((unsigned)((int)floorf(mxy) + 2) << 16);
      lo = wave_reduce_pk<true>(lo);
      hi = wave_reduce_pk<false>(hi);
      if (lane == 0) {
        bb[(wave * L + l) * 2] = lo;
        bb[(wave * L + l) * 2 + 1] = hi;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // boxes in LDS before the barrier
    __syncthreads();                                                          // (A) boxes visible; pool free

    unsigned rfetch = 0u;
    if (queue && tid == 0) rfetch = atomicAdd(queue + xcd, 1u);

    // ---- windows and phases (uniform) --------------------------------------------------------------------------------
    int wx0[L], wy0[L], wwid[L], whgt[L], wsize[L], wbase[L], phase_of[L];
    int nph;
    {
#pragma unroll
      for (int l = 0; l < L; ++l) {
        unsigned lo = lane < NW ? bb[(lane * L + l) * 2] : 0xFFFFFFFFu;
        unsigned hi = lane < NW ? bb[(lane * L + l) * 2 + 1] : 0u;
        lo = __builtin_amdgcn_readfirstlane(wave_reduce_pk<true>(lo));
        hi = __builtin_amdgcn_readfirstlane(wave_reduce_pk<false>(hi));
        const int x0 = (int)(lo & 0xFFFFu) - 1, y0 = (int)(lo >> 16) - 1;    // un-bias: origin may be -1 (apron)
        const int x1 = (int)(hi & 0xFFFFu) - 1, y1 = (int)(hi >> 16) - 1;
        const bool empty = x0 > x1 || y0 > y1;
        wx0[l] = x0;
        wy0[l] = y0;
        wwid[l] = empty ? 1 : x1 - x0 + 1;
        whgt[l] = empty ? 0 : y1 - y0 + 1;
        wsize[l] = wwid[l] * whgt[l];
      }
      int ph = 0, used = 0;
#pragma unroll
      for (int ll = 0; ll < L; ++ll) {
        const int l = L - 1 - ll;
        if (wsize[l] > pool_px - 2) {                                          // never fits: gathered from global memory
          phase_of[l] = -1;
          wbase[l] = 0;
          continue;
        }
        if (used + wsize[l] > pool_px - 2) {
          ++ph;
          used = 0;
        }
        phase_of[l] = ph;
        wbase[l] = used + 2;                                                   // pixels 0, 1 are the zero pixels
        used += wsize[l];
      }
      nph = ph + 1;
    }

    // ---- weights (FUSED: logits -> softmax).  Issued here, consumed after the first staging barrier ------------------
    float wts[L][P];
    {
      const col_f32x4 *wp = reinterpret_cast<const col_f32x4 *>(attn + rec * (L * P));
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const col_f32x4 t = wp[l];
#pragma unroll
        for (int k = 0; k < P; ++k) wts[l][k] = t[k];
      }
    }

    const float *vimg = value + (long long)b * S * MD + m * D;               // this image, this head
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value + (long long)b * S * MD), 0,
                                                        (int)((unsigned)S * (unsigned)MD * 4u), 0x00020000);

    col_f32x2 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = col_f32x2{0.f, 0.f};

    // one sample from LDS (LDS = true) or through the buffer descriptor (LDS = false)
    auto sample = [&](auto lc, auto kc, auto lds_c) {
      constexpr int l = decltype(lc)::value;
      constexpr int k = decltype(kc)::value;
      constexpr bool LDS = decltype(lds_c)::value;
      const int H = Hs[l], W = Ws[l];
      // (opaque copies: without them the compiler hoists every sample's geometry out of the phase loop -- it is
      // invariant there -- and shares it with the pre-pass, ~200 live registers, hundreds of spills)
      float sx = lxy[l][k][0], sy = lxy[l][k][1];
      asm volatile("" : "+v"(sx), "+v"(sy));
      const float h_im = fmaf(sy, fH[l], -0.5f), w_im = fmaf(sx, fW[l], -0.5f);
      const bool gate = valid && h_im > -1 && w_im > -1 && h_im < fH[l] && w_im < fW[l];   // false for NaN (cuh:290-296)
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int y0 = gate ? (int)hf : 0, x0 = gate ? (int)wf : 0;
      const float lh = gate ? h_im - hf : 0.f, lw = gate ? w_im - wf : 0.f;
      const float wgt = gate ? wts[l][k] : 0.f;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const float g1 = hh * hw * wgt, g2 = hh * lw * wgt, g3 = lh * hw * wgt, g4 = lh * lw * wgt;
      const col_f32x2 ww1 = {g1, g1}, ww2 = {g2, g2}, ww3 = {g3, g3}, ww4 = {g4, g4};
      // one pixel ROW of the sample at a time (two corners = 8 x 16 B per lane in flight): the other three waves of the
      // SIMD cover the LDS latency, and 32 data registers instead of 64 keep the kernel inside its 128-VGPR budget.
      // Corner-major FMAs: consecutive packed FMAs go to different accumulators (8 independent chains).
      // (the empty asm pins the accumulators: without it the compiler sinks a level's 128 packed FMAs below the LDS reads
      // of all four samples and spills the 256 data registers in between)
      auto pin_acc = [&]() {
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]),
                          "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
      };
      auto fma_row = [&](const col_f32x4 (&va)[4], const col_f32x4 (&vb)[4], const col_f32x2 wa, const col_f32x2 wb) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 2; ++e)
            acc[j][e] = __builtin_elementwise_fma(wa, col_f32x2{va[j][2 * e], va[j][2 * e + 1]}, acc[j][e]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 2; ++e)
            acc[j][e] = __builtin_elementwise_fma(wb, col_f32x2{vb[j][2 * e], vb[j][2 * e + 1]}, acc[j][e]);
      };
      if constexpr (LDS) {
        // all four corners lie inside the staged window (out-of-map ones are zeros); a gated-out sample reads the
        // two zero pixels (offset 0, row step 0)
        const unsigned a = gate ? (unsigned)(wbase[l] + __mul24(y0 - wy0[l], wwid[l]) + (x0 - wx0[l])) << 6 : 0u;
        const unsigned rowb = gate ? (unsigned)wwid[l] << 6 : 0u;
        {
          col_f32x4 va[4], vb[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned char *pa = pool + (a + rot[j]);
            va[j] = *reinterpret_cast<const col_f32x4 *>(pa);
            vb[j] = *reinterpret_cast<const col_f32x4 *>(pa + PXB);
          }
          fma_row(va, vb, ww1, ww2);
        }
        pin_acc();
        __builtin_amdgcn_sched_barrier(0);
        {
          col_f32x4 va[4], vb[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned char *pb = pool + (a + rowb + rot[j]);
            va[j] = *reinterpret_cast<const col_f32x4 *>(pb);
            vb[j] = *reinterpret_cast<const col_f32x4 *>(pb + PXB);
          }
          fma_row(va, vb, ww3, ww4);
        }
        pin_acc();
        __builtin_amdgcn_sched_barrier(0);
      } else {
        constexpr unsigned OOB = 0x80000000u;                                  // S * M * D * 4 < 2^31 (C ABI check)
        const bool top = gate && y0 >= 0, bot = gate && y0 + 1 <= H - 1;
        const bool lft = x0 >= 0, rgt = x0 + 1 <= W - 1;
        const unsigned MDb = (unsigned)MD * 4u;
        const unsigned a = (unsigned)(St[l] + y0 * W + x0) * MDb + (unsigned)(m * D) * 4u;
        const unsigned o1 = (top && lft) ? a : OOB, o2 = (top && rgt) ? a + MDb : OOB;
        const unsigned o3 = (bot && lft) ? a + (unsigned)W * MDb : OOB, o4 = (bot && rgt) ? a + (unsigned)W * MDb + MDb : OOB;
        {
          col_f32x4 va[4], vb[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            va[j] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(o1 + rot[j]), 0, 0));
            vb[j] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(o2 + rot[j]), 0, 0));
          }
          fma_row(va, vb, ww1, ww2);
        }
        pin_acc();
        __builtin_amdgcn_sched_barrier(0);
        {
          col_f32x4 va[4], vb[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            va[j] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(o3 + rot[j]), 0, 0));
            vb[j] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(o4 + rot[j]), 0, 0));
          }
          fma_row(va, vb, ww3, ww4);
        }
        pin_acc();
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto gather_level = [&](auto lc, auto lds_c) {
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
        (sample(lc, std::integral_constant<int, Ks>{}, lds_c), ...);
      }(std::make_integer_sequence<int, P>{});
    };

    for (int ph = 0; ph < nph; ++ph) {
      if (ph > 0) __syncthreads();                                            // every wave is done with the pool
      // ---- stage this phase's boxes by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, every piece in flight
      // at once; LDS address of a piece = wave-uniform base + lane * 16; the global source is per lane (apron lanes read
      // a 16-byte zero constant) -----------------------------------------------------------------------------------------
#pragma unroll
      for (int l = 0; l < L; ++l) {
        if (phase_of[l] == ph && wsize[l] > 0) {
          const float inv_w = uni(1.0f / (float)wwid[l]);
          const int n16 = wsize[l] * 4;
          const float *vlev = vimg + (long long)St[l] * MD;
          unsigned char *dst = pool + (size_t)wbase[l] * PXB;
          for (int it = 0; it * BLOCK < n16; ++it) {
            const int i = it * BLOCK + tid;
            if (i < n16) {
              const int px = i >> 2, cc = i & 3;
              const int r = (int)(((float)px + 0.5f) * inv_w);
              const int y = wy0[l] + r, x = wx0[l] + px - r * wwid[l];
              const bool inside = y >= 0 && y < Hs[l] && x >= 0 && x < Ws[l];
              const float *src = inside ? vlev + (long long)(y * Ws[l] + x) * MD + cc * 4 : g_zero16;
              __builtin_amdgcn_global_load_lds(
                  (const __attribute__((address_space(1))) void *)(src),
                  (__attribute__((address_space(3))) void *)(dst + (size_t)(it * BLOCK + (tid & ~63)) * 16), 16, 0, 0);
            }
          }
        }
      }
      if (ph == 0 && tid == 0) {
        unsigned fetched = (unsigned)(item - xcd * chunk + nslots);           // static stride when there is no queue
        if (queue) {
          if (rfetch + 1u >= (unsigned)(item_end - xcd * chunk)) atomicExch(queue + xcd, 0u);
          fetched = (unsigned)nslots + rfetch;
        }
        next_idx[0] = fetched;
      }
      __syncthreads();                                                        // windows staged (vmcnt(0) + barrier)

      if (ph == 0) {
        if constexpr (FUSED) {                                                // softmax over the record's L * P logits
          float mx = -INFINITY;
#pragma unroll
          for (int l = 0; l < L; ++l)
#pragma unroll
            for (int k = 0; k < P; ++k) mx = fmaxf(mx, wts[l][k]);
          float sum = 0.f;
#pragma unroll
          for (int l = 0; l < L; ++l)
#pragma unroll
            for (int k = 0; k < P; ++k) {
              wts[l][k] = __expf(wts[l][k] - mx);
              sum += wts[l][k];
            }
          const float inv = 1.f / sum;
#pragma unroll
          for (int l = 0; l < L; ++l)
#pragma unroll
            for (int k = 0; k < P; ++k) wts[l][k] *= inv;
        }
      }

      [&]<int... Ls>(std::integer_sequence<int, Ls...>) {
        ((phase_of[L - 1 - Ls] == ph ? gather_level(std::integral_constant<int, L - 1 - Ls>{}, std::true_type{}) : (void)0), ...);
      }(std::make_integer_sequence<int, L>{});
    }
    // levels whose box exceeds the pool: straight from global memory (every lane, uniform branch)
    [&]<int... Ls>(std::integer_sequence<int, Ls...>) {
      ((phase_of[Ls] < 0 ? gather_level(std::integral_constant<int, Ls>{}, std::false_type{}) : (void)0), ...);
    }(std::make_integer_sequence<int, L>{});

    if (valid) {
      float *op = out + rec * D;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<col_f32x4 *>(reinterpret_cast<unsigned char *>(op) + rot[j]) =
            col_f32x4{acc[j][0][0], acc[j][0][1], acc[j][1][0], acc[j][1][1]};
    }

    {                                                                          // (written before the staging barrier)
      const unsigned nxt = __builtin_amdgcn_readfirstlane(next_idx[0]);
      item = nxt < (unsigned)(item_end - xcd * chunk) ? xcd * chunk + (int)nxt : item_end;   // never out of the chunk
    }
  }
}

// ---- launcher: returns -100 when this geometry is not covered (caller uses another kernel) ----------------------------
// ref == nullptr: plain op; ref != nullptr: fused front-end (loc = raw offsets, attn = raw logits).
int launch_msda_forward_col(const void *value, const int64_t *shapes, const int64_t *starts, const void *loc,
                            const void *attn, int N, int S, int M, int D, int L, int Lq, int P, void *out,
                            hipStream_t stream, const float *ref, long long ref_batch_stride)
{
  if ((((uintptr_t)value | (uintptr_t)out | (uintptr_t)loc | (uintptr_t)attn) & 15u)) return -100;
  if (ref && (((uintptr_t)ref) & 7u)) return -100;
  if (D != 16 || P != 4 || L < 3 || L > 5 || Lq != S || M < 1) return -100;
  if ((long long)N * ((long long)S + 4096) * M >= 0x7fffffffLL) return -100;   // item / record arithmetic headroom
  constexpr int BLOCK = 1024;
  static const int pool_kb = [] { const char *e = getenv("PCT_COL_POOL_KB"); const int v = e ? atoi(e) : 0;
                                  return (v >= 32 && v <= 158) ? v : 150; }();
  const int pool_px = pool_kb * 1024 / 64;
  const size_t lds = (size_t)pool_px * 64 + ((size_t)(BLOCK / 64) * L * 2 + 4) * sizeof(unsigned);
  const dim3 grid(256), block(BLOCK);
  unsigned *queue = win_queue_slot(stream);                                    // nullptr: static item stride
  const float *v = static_cast<const float *>(value);
  const float *lc = static_cast<const float *>(loc), *at = static_cast<const float *>(attn);
  float *o = static_cast<float *>(out);
#define PCT_COL(L_, FU_)                                                                                                \
  do {                                                                                                                  \
    static const hipError_t attr_rc = hipFuncSetAttribute(                                                              \
        reinterpret_cast<const void *>(&msda_forward_col_kernel<L_, FU_, BLOCK>),                                       \
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                                        \
    if (attr_rc != hipSuccess) return (int)attr_rc;                                                                     \
    hipLaunchKernelGGL((msda_forward_col_kernel<L_, FU_, BLOCK>), grid, block, lds, stream, v, shapes, starts, lc, at,  \
                       N, S, M, pool_px, o, ref, ref_batch_stride, queue);                                              \
  } while (0)
  if (ref) {
    if (L == 3) PCT_COL(3, true);
    else if (L == 4) PCT_COL(4, true);
    else PCT_COL(5, true);
  } else {
    if (L == 3) PCT_COL(3, false);
    else if (L == 4) PCT_COL(4, false);
    else PCT_COL(5, false);
  }
#undef PCT_COL
  return (int)hipGetLastError();
}

}  // namespace pct
