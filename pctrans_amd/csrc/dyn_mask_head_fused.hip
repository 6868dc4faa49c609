// Per-query dynamic mask head for the bf16-autocast configuration in ONE launch (gfx950, wave64):
//   MLP (mask2former_transformer_decoder.py:699-719) -> logits in bf16 -> bilinear x2 upsample (:693-695) and the boolean
//   attention mask  sigmoid(resize to (th, tw)) < 0.5  (:689-691),
// bit-identical to the two launches of dyn_mask_head_mfma.hip (same expression trees, dmh_common.hpp) for finite logits --
// but the [N, Q, H, W] bf16 logits plane (419 MB per call at the north-star shape, written by one kernel and read back by
// the next) never exists: per call the memory side sees the 134 MB of features, the 1.68 GB of upsampled logits and the
// mask bytes, nothing else.
//
// Geometry: feature maps 128 pixels wide (the north-star shape: stride-4 mask features of a 512^2 image), height a
// multiple of 8, attention-mask target an exact 1/2, 1/4 or 1/8 of the map; everything else runs on the two-launch path.
//   * workgroup = (image, band of 8 feature rows), 4 waves, each wave owning TWO full rows (its 16 MFMA pixel tiles: lane
//     column `col` holds 8 consecutive pixels of each row) plus a quarter of the band's two halo rows; the bf16 B operands
//     of all 20 tiles stay in registers for the whole launch while the workgroup walks the query pairs;
//   * per query pair a wave computes the logits of its tiles, rounds them to bf16 and trades rows with its neighbours
//     through a double-buffered 10 KB LDS image (one barrier per pair); the MFMA accumulator layout leaves every query's
//     logit in BOTH lane groups of the query, so group g takes (query g / 2, output-row parity g % 2): all 64 lanes
//     blend, and each stores 2 x 32 contiguous bytes of two output rows -- a 16-lane group writes one whole 512-byte row;
//   * the x2 upsample needs rows y - 1 .. y + 1: the row above a wave's pair and the row below it come from LDS, the
//     columns left and right of a lane's 8 pixels from its neighbours by DPP row shifts (clamped at the map's edges as
//     PyTorch's index clamp does); the 1/2, 1/4, 1/8 attention-mask targets read a 2 x 2 block with weights 1/2 that lies
//     inside one lane's pixels and in two rows a wave holds (its own pair, or the row above + its first row).
#include "dmh_common.hpp"
#include "msda_win_common.hpp"

namespace pct {

#ifndef PCT_DMH_WGS
#define PCT_DMH_WGS 2          /* workgroups per CU the register budget is set for (3: 168 registers, ~30 spilled: 0.36 -> 0.53 ms at batch 64) */
#endif
constexpr int FZ_W = 128;              // feature-map width this kernel is built for (8 tiles of 16 pixels per row)
constexpr int FZ_BAND = 8;             // rows per workgroup
constexpr int FZ_TILES = 20;           // per wave: 2 x 8 own tiles + 4 halo tiles
constexpr int FZ_SLOTS = FZ_BAND + 2;  // LDS rows per buffer: halo above, the band, halo below

// one wave per (image, query pair): the pair's generated parameters in the form DmhPair::take reads them
template <bool REL>
__global__ __launch_bounds__(64) void dmh_prepare_kernel(const float *__restrict__ params, const float *__restrict__ ref,
                                                         const int Q, const int H, const int W, const int stride,
                                                         unsigned *__restrict__ ws)
{
  const int lane = threadIdx.x, col = lane & 15, g = lane >> 4;
  const int npairs = (Q + 1) / 2;
  const int n = blockIdx.x / npairs, pr = blockIdx.x - n * npairs;
  DmhPair<REL> pw;
  pw.load(params, ref, n, Q, pr, col, g, H, W, stride);
  pw.store_prepared(ws + (size_t)blockIdx.x * (DmhPair<REL>::PREP_DWORDS * 64), lane);
}

// up [N, Q, 2H, 256] bf16, amask [N, Q, (H/SC) * (128/SC)] bytes
template <bool REL, int SC>
__global__ __launch_bounds__(256, PCT_DMH_WGS) void dmh_fused_kernel(const float *__restrict__ feat,
                                                           const unsigned *__restrict__ ws, const int Q, const int H,
                                                           const int stride, __bf16 *__restrict__ up,
                                                           unsigned char *__restrict__ amask)
{
  constexpr int W = FZ_W, C = DMH_C;
  __shared__ __attribute__((aligned(16))) unsigned short xch[2][FZ_SLOTS][2][W];   // [buffer][row slot][query of the pair][x]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, g = lane >> 4;                    // MFMA column and lane group
  const int par = g & 1, qsel = g >> 1;                        // output-row parity / query of the pair this lane blends
  const int HW = H * W;
  const int bands = H / FZ_BAND;
  const int n = blockIdx.x / bands, band = blockIdx.x - n * bands;
  const int y0 = band * FZ_BAND + 2 * wave, y1 = y0 + 1;       // the wave's own rows
  // the band's halo rows (clamped like PyTorch's source index): waves 0, 1 share the one above, waves 2, 3 the one below;
  // each computes 4 of its 8 tiles
  const int hy = (wave >> 1) ? min(band * FZ_BAND + FZ_BAND, H - 1) : max(band * FZ_BAND - 1, 0);
  const int ht0 = 4 * (wave & 1);

  // ---- B operands of all tiles (tile t of a row <-> pixel x = 8 col + t), kept for all query pairs -----------------------
  dm_u32x4 fb[FZ_TILES];
  const float *fimg = feat + (size_t)n * C * HW;
#pragma unroll
  for (int t = 0; t < FZ_TILES; ++t) {
    const int row = t < 8 ? y0 : (t < 16 ? y1 : hy);
    const int tt = t < 16 ? (t & 7) : ht0 + (t - 16);
    fb[t] = dmh_pixel_operand<REL>(fimg, HW, W, row * W + 8 * col + tt, g, stride);
  }

  const bool top = y0 == 0;
  // vertical weights of this lane's two output rows: rows (V0, V1) and (V1, V2) of its three-row window (below)
  const float hA0 = par ? 0.75f : (top ? 1.f : 0.25f), hA1 = par ? 0.25f : (top ? 0.f : 0.75f);
  const float hB0 = par ? 0.75f : 0.25f, hB1 = par ? 0.25f : 0.75f;
  const bool left = col == 0;
  const float wl0 = left ? 1.f : 0.25f, wl1 = left ? 0.f : 0.75f;            // even output column of the lane's first pixel
  const int OW = 2 * W;
  const int th_w = W / SC;                                                   // attention-mask row length

#ifdef FZ_STAMP
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = 0;
#define FZ_ST(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); __builtin_amdgcn_sched_barrier(0); if ((i) >= 0) ts[(i) < 0 ? 0 : (i)] += t_ - tp; tp = t_; } while (0)
#else
#define FZ_ST(i) do { } while (0)
#endif
  const int npairs = (Q + 1) / 2;
  // the pairs' parameters in the form the MFMAs take them, written by dmh_prepare_kernel: 5 coalesced 16-byte loads per lane
  // and pair instead of ~40 scattered dword loads and ~100 vector instructions of splitting and packing in EACH of the
  // image's 16 bands (a fifth of a pair's cycles before); the next pair's are in flight while this pair is computed
  const unsigned *wsn = ws + (size_t)n * npairs * (DmhPair<REL>::PREP_DWORDS * 64);
  typename DmhPair<REL>::Prepared prep = DmhPair<REL>::fetch_prepared(wsn, lane);
  // The attention-mask byte of a bf16 value v is  bf16(sigmoid(v)) < 0.5.  sigmoid is monotonic and bf16(.) rounds 0.5 - x/4
  // to 0.5 for every bf16 |x| < 2^-8 and below it for every |x| > 2^-8, so the byte is (v < -2^-8) except AT v = -2^-8, where
  // the fp32 evaluation decides: evaluated once here with the very expression the two-launch path uses
  const bool edge_masked = dm_mask_byte(0.5f, 0.5f, -0.00390625f, -0.00390625f, -0.00390625f, -0.00390625f) != 0;
  for (int pr = 0; pr < npairs; ++pr) {
    FZ_ST(-1);
    DmhPair<REL> pw;
    pw.take(prep);
    prep = DmhPair<REL>::fetch_prepared(wsn + (size_t)min(pr + 1, npairs - 1) * (DmhPair<REL>::PREP_DWORDS * 64), lane);
    const int q_acc = min(2 * pr + qsel, Q - 1);
    const int buf = pr & 1;
    FZ_ST(0);

    // ---- logits of my tiles, rounded to bf16 (what the two-launch path stores and reads back) -----------------------------
    unsigned pk0[4], pk1[4], pkh[2];
    {
      float lg[FZ_TILES];
#ifndef FZ_BATCH
#define FZ_BATCH 4
#endif
#pragma unroll
      for (int t0 = 0; t0 < FZ_TILES; t0 += FZ_BATCH) pw.template tiles<FZ_BATCH>(fb + t0, lg + t0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pk0[k] = pack_bf16x2(lg[2 * k], lg[2 * k + 1]);
        pk1[k] = pack_bf16x2(lg[8 + 2 * k], lg[8 + 2 * k + 1]);
      }
      pkh[0] = pack_bf16x2(lg[16], lg[17]);
      pkh[1] = pack_bf16x2(lg[18], lg[19]);
    }

    FZ_ST(1);
    // ---- trade rows: group parity 0 publishes the wave's first row and its halo quarter, parity 1 the second row -------
    {
      const dm_u32x4 own = par ? dm_u32x4{pk1[0], pk1[1], pk1[2], pk1[3]} : dm_u32x4{pk0[0], pk0[1], pk0[2], pk0[3]};
      *reinterpret_cast<dm_u32x4 *>(&xch[buf][1 + 2 * wave + par][qsel][8 * col]) = own;
      if (!par)
        *reinterpret_cast<dm_u32x2 *>(&xch[buf][(wave >> 1) ? FZ_SLOTS - 1 : 0][qsel][8 * col + ht0]) = dm_u32x2{pkh[0], pkh[1]};
    }
    FZ_ST(2);
    __syncthreads();
    FZ_ST(3);
    // the row above my pair (parity 0) or below it (parity 1)
    const dm_u32x4 hal = *reinterpret_cast<const dm_u32x4 *>(&xch[buf][par ? 2 * wave + 3 : 2 * wave][qsel][8 * col]);

    // ---- this lane's three-row window, columns x0 - 1 .. x0 + 8:  parity 0: (above, row y0, row y1),  parity 1: (row y0,
    // row y1, below).  e[r][1 + k] = pixel x0 + k; e[r][0], e[r][9] come from the neighbouring lanes (clamped at the edges)
    float e[3][10];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned w0 = par ? pk0[k] : hal[k], w1 = par ? pk1[k] : pk0[k], w2 = par ? hal[k] : pk1[k];
      e[0][1 + 2 * k] = bf16_lo(w0);
      e[0][2 + 2 * k] = bf16_hi(w0);
      e[1][1 + 2 * k] = bf16_lo(w1);
      e[1][2 + 2 * k] = bf16_hi(w1);
      e[2][1 + 2 * k] = bf16_lo(w2);
      e[2][2 + 2 * k] = bf16_hi(w2);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      // row_shr:1 hands lane c the value of lane c - 1, row_shl:1 that of lane c + 1; a lane without a source (the first /
      // last of its 16-lane row = the map's edge) keeps `old` = its own edge pixel
      e[r][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e[r][1]),
                                                                       __builtin_bit_cast(int, e[r][8]), 0x111, 0xf, 0xf, false));
      e[r][9] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e[r][8]),
                                                                       __builtin_bit_cast(int, e[r][1]), 0x101, 0xf, 0xf, false));
    }

    FZ_ST(4);
    // ---- x2 upsample: horizontal blends of the three rows once, then the two output rows ---------------------------------
    // even output column 2 (x0 + k): source columns (x0 + k - 1, x0 + k), weights (1/4, 3/4) -- at the map's left edge
    // (1, 0) on the clamped pair; odd column: (x0 + k, x0 + k + 1), weights (3/4, 1/4)
    const bool wr = 2 * pr + qsel < Q;
    __bf16 *uplane = up + ((size_t)n * Q + q_acc) * (size_t)(4 * HW);
    float tx[3][16];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        tx[r][2 * k] = k == 0 ? dm_blend_x(wl0, wl1, e[r][0], e[r][1]) : dm_blend_x(0.25f, 0.75f, e[r][k], e[r][k + 1]);
        tx[r][2 * k + 1] = dm_blend_x(0.75f, 0.25f, e[r][k + 1], e[r][k + 2]);
      }
    dm_u32x4 oa[2], ob[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      oa[k >> 2][k & 3] = pack_bf16x2(dm_blend_y(hA0, hA1, tx[0][2 * k], tx[1][2 * k]),
                                       dm_blend_y(hA0, hA1, tx[0][2 * k + 1], tx[1][2 * k + 1]));
      ob[k >> 2][k & 3] = pack_bf16x2(dm_blend_y(hB0, hB1, tx[1][2 * k], tx[2][2 * k]),
                                       dm_blend_y(hB0, hB1, tx[1][2 * k + 1], tx[2][2 * k + 1]));
    }
#ifdef FZ_DEBUG_NEAREST
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      oa[k >> 2][k & 3] = pack_bf16x2(e[par ? 0 : 1][1 + k], e[par ? 0 : 1][1 + k]);
      ob[k >> 2][k & 3] = pack_bf16x2(e[par ? 1 : 2][1 + k], e[par ? 1 : 2][1 + k]);
    }
#endif
    if (wr) {
      // output rows 2 y0 + par (from window rows 0, 1) and 2 y1 + par (rows 1, 2), columns 16 col .. 16 col + 15
      __bf16 *ra = uplane + (size_t)(2 * y0 + par) * OW + 16 * col;
      __bf16 *rb = uplane + (size_t)(2 * y1 + par) * OW + 16 * col;
      // (non-temporal: 1.68 GB per call that nothing in this kernel reads back -- kept out of the L2's way of the feature
      // tiles and the query pairs' parameters: 0.64-0.66 -> 0.58-0.61 ms per call at batch 128, two same-box alternations)
      __builtin_nontemporal_store(oa[0], reinterpret_cast<dm_u32x4 *>(ra));
      __builtin_nontemporal_store(oa[1], reinterpret_cast<dm_u32x4 *>(ra + 8));
      __builtin_nontemporal_store(ob[0], reinterpret_cast<dm_u32x4 *>(rb));
      __builtin_nontemporal_store(ob[1], reinterpret_cast<dm_u32x4 *>(rb + 8));
    }

    FZ_ST(5);
    // ---- attention mask at 1 / SC of the map: target (ty, tx) reads rows SC ty + SC/2 - 1, + 1 and the same two columns,
    // all four weights 1/2.  SC = 2: my own rows (window rows 1, 2 of parity 0); SC = 4, 8: the row above + my first row
    // (window rows 0, 1), in the waves whose first row is SC/2 (mod SC).
    if ((SC == 2 || (y0 & (SC - 1)) == SC / 2) && !par && wr) {
      constexpr int RA = SC == 2 ? 1 : 0;
      const int ty = SC == 2 ? y0 / 2 : (y0 - SC / 2) / SC;
      unsigned bytes = 0;
#pragma unroll
      for (int j = 0; j < 8 / SC; ++j) {
        const int xl = SC * j + SC / 2 - 1;                     // pixel index inside the lane's 8 (e[][1 + xl])
        const float v = (float)(__bf16)dm_blend(0.5f, 0.5f, 0.5f, 0.5f, e[RA][1 + xl], e[RA][2 + xl], e[RA + 1][1 + xl], e[RA + 1][2 + xl]);
        bytes |= (unsigned)(v < -0.00390625f || (edge_masked && v == -0.00390625f)) << (8 * j);
      }
      unsigned char *mp = amask + ((size_t)n * Q + q_acc) * (size_t)((H / SC) * th_w) + (size_t)ty * th_w + (8 / SC) * col;
      if constexpr (SC == 2) *reinterpret_cast<unsigned *>(mp) = bytes;
      else if constexpr (SC == 4) *reinterpret_cast<unsigned short *>(mp) = (unsigned short)bytes;
      else *mp = (unsigned char)bytes;
    }
    FZ_ST(6);
  }
#ifdef FZ_STAMP
  if (blockIdx.x == 7 && threadIdx.x == 64)
    for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long *>(amask)[i] = ts[i];
#endif
}

// -100: geometry not covered (the caller runs the two-launch path).  `workspace`: N * ceil(Q / 2) * 5120 bytes, 16-byte aligned.
int launch_dyn_mask_head_fused(const float *feat, const float *ref, const float *params, int N, int C, int Q, int H, int W,
                               int stride, int rel_coord, int th, int tw, void *workspace, void *up, unsigned char *amask,
                               hipStream_t stream)
{
  if (C != DMH_C || W != FZ_W || H < FZ_BAND || (H % FZ_BAND) != 0) return -100;
  if (th <= 0 || tw <= 0 || H % th != 0 || W % tw != 0 || H / th != W / tw) return -100;
  const int sc = H / th;
  if (sc != 2 && sc != 4 && sc != 8) return -100;
  const int npairs = (Q + 1) / 2;
  if ((long long)N * (H / FZ_BAND) > 0x7fffffffLL || (long long)N * npairs > 0x7fffffffLL) return -100;
  if (((uintptr_t)up & 15u) || ((uintptr_t)amask & 3u) || ((uintptr_t)workspace & 15u)) return -100;
  if (sc == 2 && (((size_t)(H / 2) * (W / 2)) & 3u)) return -100;          // 4-byte mask stores
  unsigned *ws = static_cast<unsigned *>(workspace);
  if (rel_coord)
    hipLaunchKernelGGL(dmh_prepare_kernel<true>, dim3((unsigned)(N * npairs)), dim3(64), 0, stream, params, ref, Q, H, W, stride, ws);
  else
    hipLaunchKernelGGL(dmh_prepare_kernel<false>, dim3((unsigned)(N * npairs)), dim3(64), 0, stream, params, ref, Q, H, W, stride, ws);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  const dim3 grid((unsigned)(N * (H / FZ_BAND))), block(256);
  __bf16 *u = static_cast<__bf16 *>(up);
#define PCT_FZ(REL_, SC_) \
  hipLaunchKernelGGL((dmh_fused_kernel<REL_, SC_>), grid, block, 0, stream, feat, ws, Q, H, stride, u, amask)
  if (rel_coord) {
    if (sc == 2) PCT_FZ(true, 2);
    else if (sc == 4) PCT_FZ(true, 4);
    else PCT_FZ(true, 8);
  } else {
    if (sc == 2) PCT_FZ(false, 2);
    else if (sc == 4) PCT_FZ(false, 4);
    else PCT_FZ(false, 8);
  }
#undef PCT_FZ
  return (int)hipGetLastError();
}

}  // namespace pct
