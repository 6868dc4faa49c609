// MSDeformAttn forward, pyramid-column kernel for 16-BIT values (fp16 / bf16; fp32 sampling locations and weights, fp32
// accumulation) with 4 or 8 sampling points per level -- BASELINE.json configs[4] (1024^2, 5 levels, 8 points, fp16) and
// the 16-bit variants of the other geometries.  New capability: the reference op is fp32 / fp64 only (cu:69,139).
//
// Same semantics as msda_forward.hip (reference: ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 + :38-89) and the same
// structure as msda_forward_col.hip (read that file's header first): work item = (image, pyramid column, head), per-level
// bounding-box windows staged by LDS-DMA with a zero apron, phases when the windows do not fit together, fixed level
// order, items handed out per XCD two ahead so that the next item's sampling locations are in flight during the gather.
// What differs:
//   * a head-pixel is 32 bytes (16 channels x 2 B): two ds_read_b128 per corner, each converted and accumulated in fp32
//     (fp16: v_fma_mix_f32; bf16: a shift / mask per element); the pool holds twice the pixels per KB;
//   * 8 points per level would need 80 registers of locations per lane: with P = 8 a (query, head) pair is shared by TWO
//     adjacent lanes, lane h taking points 4h .. 4h + 3 of every level (its own 32 bytes of every level's 64-byte location
//     group and 16 bytes of the weight group, loaded directly: the two lanes together read whole 64-byte groups), the
//     soft-max statistics and the final sums cross the pair by DPP, and lane h stores channels 8h .. 8h + 7;
//   * records are loaded lane by lane (no quad transposes): a lane's 32 / 64 contiguous bytes per level.
#include <math.h>
#include <stdlib.h>

#include <utility>

#include "msda_col_common.hpp"

namespace pct {

// Gather order of the levels (step -> level): finest, coarsest, then the rest (as in msda_forward_col.hip: the coarsest
// level is the one that most often still fits beside the finest in the pool, which saves a phase).
template <int L>
__host__ __device__ constexpr int col16_level_of_step(const int ll)
{
  return ll == 0 ? L - 1 : (ll == 1 ? 0 : L - ll);
}

template <typename T, int L, int P, bool FUSED, int BLOCK>
__global__ __launch_bounds__(BLOCK, 3) void msda_forward_col16_kernel(
    const typename Traits<T>::store_t *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const float *__restrict__ loc, const float *__restrict__ attn, const int N,
    const int S, const int M, const int pool_px, typename Traits<T>::store_t *__restrict__ out,
    const float *__restrict__ ref, const long long ref_batch_stride, unsigned *__restrict__ queue)
{
  using ST = typename Traits<T>::store_t;
  constexpr int D = 16, PXB = 32, NW = BLOCK / 64;
  constexpr int HALVES = P / 4;                   // lanes per (query, head)
  constexpr int PL = 4;                           // points per lane and level
  constexpr int QPW = BLOCK / HALVES;             // queries per item
  typedef ST st8 __attribute__((ext_vector_type(8)));
  static_assert(sizeof(ST) == 2 && (P == 4 || P == 8) && L >= 1 && L <= 5 && NW <= 16, "unsupported geometry");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char *pool = smem_raw;                                              // level windows, 32 B per head-pixel
  unsigned *bb = reinterpret_cast<unsigned *>(smem_raw + (size_t)pool_px * PXB);   // [L][NW][2] per-wave boxes {min lo, ~max hi}
  unsigned *next_idx = bb + NW * L * 2;                                        // [0] next item, [1] counter value fetched

  int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int MD = M * D;

  // ---- level geometry (uniform) -------------------------------------------------------------------------------------
  int Hs[L], Ws[L], St[L];
  float fH[L], fW[L];
  col_f32x2 fWH[L], invWH[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
    St[l] = (int)starts[l];
    fH[l] = uni((float)Hs[l]);
    fW[l] = uni((float)Ws[l]);
    fWH[l] = uni_pair((float)Ws[l], (float)Hs[l]);
    invWH[l] = uni_pair(1.0f / (float)Ws[l], 1.0f / (float)Hs[l]);
  }

  // ---- column grid: CX x CY cells such that no column holds more than QPW queries (as msda_forward_col.hip) ----------
  int CX, CY;
  {
    int Hf = Hs[0], Wf = Ws[0];
#pragma unroll
    for (int l = 1; l < L; ++l)
      if (Hs[l] * Ws[l] > Hf * Wf) { Hf = Hs[l]; Wf = Ws[l]; }
    const float area = (float)QPW * (float)(Hf * Wf) / (float)S;
    const int side = (int)sqrtf(area);
    const int nxt = min(Wf, max(8, (side + 4) & ~7));
    CX = (Wf + nxt - 1) / nxt;
    const int nx0 = (Wf + CX - 1) / CX;
    const int nyt = max(1, (int)(area / (float)nx0));
    CY = min(Hf, (Hf + nyt - 1) / nyt);
    for (int guard = 0; guard < 4096; ++guard) {
      int maxq = 0;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        int mx = 0, my = 0;
        mx = col_max_cell(Ws[l], CX);
        my = col_max_cell(Hs[l], CY);
        maxq += mx * my;
      }
      if (maxq <= QPW) break;
      if (CY < Hf) ++CY;
      else if (CX < Wf) ++CX;
      else break;
    }
  }
  // ---- this lane's SLOT in a column and the cell tables in LDS (as msda_forward_col.hip: a (query, head) pair's place --
  // level, x, y inside the level's largest cell -- is fixed for the launch, the per-item decode is two table look-ups and
  // a multiply-add instead of a walk over the levels with four scalar divisions each) ---------------------------------
  int lane_c0 = 0;
  unsigned lane_slot = 0x000FFFFFu;                                            // x | y << 10 | level << 20; x = y = 1023: no slot
  {
    int r = tid / HALVES;
    bool placed = false;
#pragma unroll
    for (int ll = 0; ll < L; ++ll) {
      const int l = L - 1 - ll;
      const int mx = col_max_cell(Ws[l], CX), my = col_max_cell(Hs[l], CY);
      const int cnt = mx * my;
      if (!placed && r < cnt) {
        const int ly = r / max(mx, 1), lx = r - ly * max(mx, 1);
        lane_slot = (unsigned)lx | ((unsigned)ly << 10) | ((unsigned)l << 20);
        lane_c0 = St[l] + ly * Ws[l] + lx;
        placed = true;
      }
      r -= placed ? 0 : cnt;
    }
  }
  const int tab_px = ((L * (CX + CY) + L) * 4 + PXB - 1) / PXB;
  const int pool_eff = pool_px - tab_px;
  unsigned *tab = reinterpret_cast<unsigned *>(pool + (size_t)max(pool_eff, 0) * PXB);
  const bool use_tab = pool_eff * 2 >= pool_px;            // else: FLAT columns (absurdly elongated maps), see the decode
  if (use_tab) {
    for (int t = tid; t < L * CX; t += BLOCK) {
      const int l = t / CX, c = t - l * CX;
      const int a = col_lo(c, Ws[l], CX);
      tab[t] = (unsigned)a | ((unsigned)(col_lo(c + 1, Ws[l], CX) - a) << 16);
    }
    for (int t = tid; t < L * CY; t += BLOCK) {
      const int l = t / CY, c = t - l * CY;
      const int a = col_lo(c, Hs[l], CY);
      tab[L * CX + t] = (unsigned)a | ((unsigned)(col_lo(c + 1, Hs[l], CY) - a) << 16);
    }
    if (tid < L) tab[L * (CX + CY) + tid] = (unsigned)Ws[tid];
    __syncthreads();
  }
  const int pool_use = use_tab ? pool_eff : pool_px;                           // pixels the windows may take
  bool big_map = false;                                                        // a level too large for the 16-bit box corners
#pragma unroll
  for (int l = 0; l < L; ++l) big_map = big_map || Ws[l] > 65531 || Hs[l] > 65531;
  int ncol = CX * CY;
  if (!use_tab) {
    ncol = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) ncol += (Hs[l] * Ws[l] + QPW - 1) / QPW;
  }
  const int items = N * ncol * M;
  const UDiv dv_ncolM = make_udiv(ncol * M), dv_2ncol = make_udiv(2 * ncol), dv_CX = make_udiv(CX);

  // pixels 0 and 1 of the pool are zeros: gated-out samples read them
  if (tid < 4) reinterpret_cast<col_f32x4 *>(pool)[tid] = col_f32x4{0.f, 0.f, 0.f, 0.f};

  // Which of a pixel's two 16-byte pieces a lane reads first (the other second): the 16 lanes the LDS serves together
  // see each pixel residue (mod 8) twice -- the two halves of a (query, head) pair with P = 8, the two 8-query rows of
  // the lane group with P = 4 -- and the piece order keeps those two off each other's banks.
  const unsigned rho = HALVES == 2 ? (unsigned)(lane & 1) : (unsigned)(lane >> 4) & 1u;
  unsigned rot[2];
  rot[0] = rho << 4;
  rot[1] = (rho ^ 1u) << 4;

  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int chunk = (items + 7) / 8;
  const int item_end = min((xcd + 1) * chunk, items);
  const unsigned n_x = (unsigned)max(item_end - xcd * chunk, 0);

  // item -> (image, head) [uniform] and this lane's query (qv = q, or ~q of the query an idle lane shadows); order inside
  // an image: head pair, column, head in the pair (msda_forward_col.hip)
  auto decode = [&](const int it, int &b_, int &m_, int &qv_) {
    b_ = udiv_s(it, dv_ncolM);
    const int r_img = it - b_ * (ncol * M);
    int col;
    if (r_img < 2 * ncol * (M >> 1)) {
      const int pr = udiv_s(r_img, dv_2ncol);
      const int rr = r_img - pr * 2 * ncol;
      col = rr >> 1;
      m_ = 2 * pr + (rr & 1);
    } else {
      col = r_img - 2 * ncol * (M >> 1);
      m_ = M - 1;
    }
    const int cy = udiv_s(col, dv_CX), cx = col - cy * CX;
    int q;
    bool valid;
    if (use_tab) {
      const unsigned lx = lane_slot & 0x3FFu, ly = (lane_slot >> 10) & 0x3FFu, lv = lane_slot >> 20;
      const unsigned char *tb = reinterpret_cast<const unsigned char *>(tab);
      const unsigned xt = *reinterpret_cast<const unsigned *>(tb + (__umul24(lv, (unsigned)(CX * 4)) + (unsigned)(cx * 4)));
      const unsigned yt = *reinterpret_cast<const unsigned *>(tb + (__umul24(lv, (unsigned)(CY * 4)) + (unsigned)((L * CX + cy) * 4)));
      const unsigned lw = *reinterpret_cast<const unsigned *>(tb + (lv * 4u + (unsigned)(L * (CX + CY) * 4)));
      valid = lx < (xt >> 16) && ly < (yt >> 16);
      q = lane_c0 + (int)__umul24(yt & 0xFFFFu, lw) + (int)(xt & 0xFFFFu);
    } else {
      int c = col;
      q = 0;
      valid = false;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const int cnt = Hs[l] * Ws[l], nch = (cnt + QPW - 1) / QPW;
        if (c >= 0 && c < nch) {
          q = St[l] + c * QPW + tid / HALVES;
          valid = c * QPW + tid / HALVES < cnt;
        }
        c = c >= nch ? c - nch : -1;
      }
    }
    // idle lanes shadow a query of the column (the wave's first busy lane's; none: query 0 -- correct, merely wide boxes)
    const unsigned long long vm = __ballot(valid);
    const int q_sh = vm ? __builtin_amdgcn_readlane(q, (int)__builtin_ctzll(vm)) : 0;
    qv_ = valid ? q : ~q_sh;
  };
  // this lane's 4 points of level l of its record: 32 contiguous bytes (per-image base uniform, 32-bit offsets)
  auto issue_loc_level = [&](auto lc, const int b_, const int m_, const int qv_, col_f32x4 (&raw)[L][2]) {
    constexpr int l = decltype(lc)::value;
    const float *base = loc + (long long)b_ * S * M * (L * P * 2);
    const unsigned r = (unsigned)((qv_ < 0 ? ~qv_ : qv_) * M + m_);
    const unsigned o = r * (unsigned)(L * P * 2) + (unsigned)(l * P * 2) + (unsigned)((tid % HALVES) * 8);
    raw[l][0] = *reinterpret_cast<const col_f32x4 *>(base + (size_t)o);
    raw[l][1] = *reinterpret_cast<const col_f32x4 *>(base + (size_t)(o + 4u));
  };

  int item = xcd * chunk + slot0;
  bool have = item < item_end;
  int b = 0, m = 0, qv = 0;
  col_f32x4 raw[L][2];
  if (have) {
    decode(item, b, m, qv);
    [&]<int... Ls>(std::integer_sequence<int, Ls...>) {
      (issue_loc_level(std::integral_constant<int, Ls>{}, b, m, qv, raw), ...);
    }(std::make_integer_sequence<int, L>{});
    if (queue && tid == 0) next_idx[1] = atomicAdd(queue + xcd * WIN_QUEUE_STRIDE, 1u);
  }

  while (have) {
    asm volatile("" : "+v"(tid), "+v"(lane_slot), "+v"(lane_c0));     // (opaque per iteration: see msda_forward_col.hip)
    const int hh = tid % HALVES;
    const bool valid = qv >= 0;
    const int q = valid ? qv : ~qv;
    const long long rec_img = (long long)b * S;

    col_f32x2 rr[L];
    if constexpr (FUSED) {
      const float *rrow = ref + b * ref_batch_stride;
      const unsigned o = (unsigned)q * (unsigned)(L * 2);
#pragma unroll
      for (int l = 0; l < L; ++l) rr[l] = *reinterpret_cast<const col_f32x2 *>(rrow + (size_t)(o + 2u * l));
    }

    // ---- the lane's sampling locations -> pixel coordinates (w_im, h_im) = loc * (W, H) - 0.5 -------------------------
    col_f32x2 lxy[L][PL];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int k = 0; k < PL; ++k) {
        col_f32x2 v = {raw[l][k >> 1][(k & 1) * 2], raw[l][k >> 1][(k & 1) * 2 + 1]};
        if constexpr (FUSED) v = __builtin_elementwise_fma(v, invWH[l], rr[l]);
        lxy[l][k] = __builtin_elementwise_fma(v, fWH[l], col_f32x2{-0.5f, -0.5f});
      }
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int k = 0; k < PL; ++k) asm volatile("" : "+v"(lxy[l][k]));
    __builtin_amdgcn_sched_barrier(0);

    // ---- publish the next item's index, fetch the one after it (msda_forward_col.hip: queue protocol, two ahead) -----
    unsigned f_new = 0u;
    bool fetched = false;
    if (tid == 0) {
      unsigned nxt = (unsigned)(item - xcd * chunk + nslots);
      if (queue) {
        const unsigned f_next = next_idx[1];
        if (f_next + 1u >= n_x) __hip_atomic_store(queue + xcd * WIN_QUEUE_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nxt = (unsigned)nslots + f_next;
        if (nxt < n_x) {
          const unsigned one = 1u, zero = 0u;
          const unsigned *qp = queue + xcd * WIN_QUEUE_STRIDE;
          asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0"
                       : "=v"(f_new) : "v"(zero), "v"(one), "s"(qp) : "memory");
          fetched = true;
        }
      }
      next_idx[0] = nxt;
    }

    // ---- pre-pass: per-level bounding box of the lane's four first corners (biased by +1) ----------------------------
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float mnx = __builtin_amdgcn_fmed3f(fminf(fminf(lxy[l][0][0], lxy[l][1][0]), fminf(lxy[l][2][0], lxy[l][3][0])),
                                                -1.f, uni(fW[l] - 0.5f));
      const float mxx = __builtin_amdgcn_fmed3f(fmaxf(fmaxf(lxy[l][0][0], lxy[l][1][0]), fmaxf(lxy[l][2][0], lxy[l][3][0])),
                                                -1.f, uni(fW[l] - 0.5f));
      const float mny = __builtin_amdgcn_fmed3f(fminf(fminf(lxy[l][0][1], lxy[l][1][1]), fminf(lxy[l][2][1], lxy[l][3][1])),
                                                -1.f, uni(fH[l] - 0.5f));
      const float mxy = __builtin_amdgcn_fmed3f(fmaxf(fmaxf(lxy[l][0][1], lxy[l][1][1]), fmaxf(lxy[l][2][1], lxy[l][3][1])),
                                                -1.f, uni(fH[l] - 0.5f));
      const unsigned lo = (unsigned)(cvt_flr(mnx) + 1) | ((unsigned)(cvt_flr(mny) + 1) << 16);
      const unsigned hi = (unsigned)(cvt_flr(mxx) + 2) | ((unsigned)(cvt_flr(mxy) + 2) << 16);
      const unsigned red = wave_reduce_box(lo, hi);                 // lane 31: min lo, lane 63: ~max hi
      if ((lane & 31) == 31) bb[(l * NW + wave) * 2 + (lane >> 5)] = red;
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- this lane's weights (FUSED: logits): its 4 points of every level, 16 bytes per level ------------------------
    float wts[L][PL];
    {
      const float *wbase_img = attn + rec_img * M * (L * P);
      const unsigned r = (unsigned)(q * M + m);
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const col_f32x4 t = *reinterpret_cast<const col_f32x4 *>(
            wbase_img + (size_t)(r * (unsigned)(L * P) + (unsigned)(l * P) + (unsigned)(hh * 4)));
#pragma unroll
        for (int k = 0; k < PL; ++k) wts[l][k] = t[k];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                                          // (A) boxes visible; pool free

    int item_n, b_n = 0, m_n = 0, qv_n = 0;
    {
      const unsigned nxt = __builtin_amdgcn_readfirstlane(next_idx[0]);
      item_n = nxt < n_x ? xcd * chunk + (int)nxt : item_end;
    }
    const bool have_n = item_n < item_end;

    // ---- windows and phases (uniform) --------------------------------------------------------------------------------
    int wx0[L], wy0[L], wwid[L], whgt[L], wsize[L], wbase[L], phase_of[L];
    bool starts_phase[L];
    {
#pragma unroll
      for (int l = 0; l < L; ++l) {
        unsigned lo, hi;
        block_box<NW>(bb + l * NW * 2, lo, hi);
        const int x0 = (int)(lo & 0xFFFFu) - 1, y0 = (int)(lo >> 16) - 1;
        const int x1 = (int)(hi & 0xFFFFu) - 1, y1 = (int)(hi >> 16) - 1;
        const bool empty = x0 > x1 || y0 > y1;
        wx0[l] = x0;
        wy0[l] = y0;
        wwid[l] = empty ? 1 : x1 - x0 + 1;
        whgt[l] = empty ? 0 : y1 - y0 + 1;
        wsize[l] = wwid[l] * whgt[l];
      }
      int ph = 0, used = 0;
      bool fresh = true;
#pragma unroll
      for (int ll = 0; ll < L; ++ll) {
        const int l = col16_level_of_step<L>(ll);
        starts_phase[l] = false;
        if (wsize[l] > pool_use - 2 || big_map) {     // (or too large for the 16-bit box corners)
          phase_of[l] = -1;
          wbase[l] = 0;
          continue;
        }
        if (used + wsize[l] > pool_use - 2) {
          ++ph;
          used = 0;
          fresh = true;
        }
        phase_of[l] = ph;
        starts_phase[l] = fresh;
        fresh = false;
        wbase[l] = used + 2;
        used += wsize[l];
      }
    }

    int woff[L];                                  // pixel (x0, y0) sits at pool index woff + y0 * width + x0
#pragma unroll
    for (int l = 0; l < L; ++l) woff[l] = wbase[l] - wy0[l] * wwid[l] - wx0[l];
    // (void * on purpose: with a _Float16 / __bf16 pointer here the host pass silently drops the kernel stubs)
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(const_cast<ST *>(value + rec_img * MD)), 0,
                                                        (int)((unsigned)S * (unsigned)MD * 2u), 0x00020000);
    float acc[2][8];                              // [piece slot j][channel]: slot j holds piece j ^ rho
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
    auto pin_acc = [&]() {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(acc[j][e]));
    };
    auto fma_piece = [&](const int j, const col_f32x4 raw16, const float w) {
      const st8 v = __builtin_bit_cast(st8, raw16);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[j][e] = fmaf(w, Traits<T>::to_acc(v[e]), acc[j][e]);
    };
    struct Geo {
      col_f32x2 g12, g34;
      int x0, y0;
      bool gate;
    };
    auto geometry = [&](auto lc, auto kc) {
      constexpr int l = decltype(lc)::value;
      constexpr int k = decltype(kc)::value;
      col_f32x2 pix = lxy[l][k];
      asm volatile("" : "+v"(pix));
      Geo g;
      g.gate = pix[1] > -1 && pix[0] > -1 && pix[1] < fH[l] && pix[0] < fW[l];    // false for NaN (cuh:290-296)
      pix[0] = g.gate ? pix[0] : 0.f;
      pix[1] = g.gate ? pix[1] : 0.f;
      const float wgt = g.gate ? wts[l][k] : 0.f;
      g.x0 = cvt_flr(pix[0]);
      g.y0 = cvt_flr(pix[1]);
      const float lw = __builtin_amdgcn_fractf(pix[0]), lh = __builtin_amdgcn_fractf(pix[1]);
      const col_f32x2 ax = {1.f - lw, lw}, ay = {1.f - lh, lh};
      const col_f32x2 t = ax * col_f32x2{wgt, wgt};
      g.g12 = t * col_f32x2{ay[0], ay[0]};
      g.g34 = t * col_f32x2{ay[1], ay[1]};
      return g;
    };
    auto gather_level_lds = [&](auto lc) {
      constexpr int l = decltype(lc)::value;
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
        ([&] {
          const Geo g = geometry(lc, std::integral_constant<int, Ks>{});
          const unsigned a = g.gate ? (unsigned)(__mul24(g.y0, wwid[l]) + g.x0 + woff[l]) << 5 : 0u;
          const unsigned rowb = g.gate ? (unsigned)wwid[l] << 5 : 0u;
          // one pixel row (two corners, 4 x 16 B) in flight at a time: with all four corners the 5-level, 8-point
          // instantiation spilled the next item's prefetched locations -- a wait for the load just issued, every item
          {
            col_f32x4 v1[2], v2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const unsigned char *pa = pool + (a + rot[j]);
              v1[j] = *reinterpret_cast<const col_f32x4 *>(pa);
              v2[j] = *reinterpret_cast<const col_f32x4 *>(pa + PXB);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              fma_piece(j, v1[j], g.g12[0]);
              fma_piece(j, v2[j], g.g12[1]);
            }
          }
          pin_acc();
          __builtin_amdgcn_sched_barrier(0);
          {
            col_f32x4 v3[2], v4[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const unsigned char *pb = pool + (a + rowb + rot[j]);
              v3[j] = *reinterpret_cast<const col_f32x4 *>(pb);
              v4[j] = *reinterpret_cast<const col_f32x4 *>(pb + PXB);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              fma_piece(j, v3[j], g.g34[0]);
              fma_piece(j, v4[j], g.g34[1]);
            }
          }
          pin_acc();
          __builtin_amdgcn_sched_barrier(0);
        }(), ...);
      }(std::make_integer_sequence<int, PL>{});
    };
    // A level whose box exceeds the pool, from global memory.  As in msda_forward_col.hip the fetches are shared so that
    // a wave instruction touches whole pixels: here a head-pixel is 32 bytes, so the PAIR of lanes (2i, 2i + 1) works on
    // one member's sample at a time -- corner offsets and weights broadcast by DPP, lane c fetches piece c of every corner
    // (32 tag look-ups per instruction instead of 64) and accumulates "piece c of member s's sum"; after the level the
    // pair exchanges the halves.  (With 8 points the two members are the two halves of one (query, head); with 4 points
    // two neighbouring queries.)
    auto gather_level_global = [&](auto lc) {
      constexpr int l = decltype(lc)::value;
      const int H = Hs[l], W = Ws[l];
      constexpr unsigned OOB = 0x80000000u;
      const unsigned MDb = (unsigned)MD * 2u;
      const unsigned pc = (unsigned)(lane & 1) << 4;                           // (an out-of-range offset stays out of range)
      float part[2][8];                                                        // [member s][channel of piece lane & 1]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 8; ++e) part[s2][e] = 0.f;
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
        ([&] {
          const Geo g = geometry(lc, std::integral_constant<int, Ks>{});
          const bool top = g.gate && g.y0 >= 0, bot = g.gate && g.y0 + 1 <= H - 1;
          const bool lft = g.x0 >= 0, rgt = g.x0 + 1 <= W - 1;
          const unsigned a = (unsigned)(St[l] + g.y0 * W + g.x0) * MDb + (unsigned)(m * D) * 2u;
          const unsigned o1 = (top && lft) ? a : OOB, o2 = (top && rgt) ? a + MDb : OOB;
          const unsigned o3 = (bot && lft) ? a + (unsigned)W * MDb : OOB, o4 = (bot && rgt) ? a + (unsigned)W * MDb + MDb : OOB;
          [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
            ([&] {
              constexpr int CT = BcastCtrl<2, Ss>::value;
              col_f32x4 v[4];
              v[0] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(dpp_u<CT>(o1) + pc), 0, 0));
              v[1] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(dpp_u<CT>(o2) + pc), 0, 0));
              v[2] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(dpp_u<CT>(o3) + pc), 0, 0));
              v[3] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(dpp_u<CT>(o4) + pc), 0, 0));
              const float w[4] = {dpp_f<CT>(g.g12[0]), dpp_f<CT>(g.g12[1]), dpp_f<CT>(g.g34[0]), dpp_f<CT>(g.g34[1])};
#pragma unroll
              for (int c4 = 0; c4 < 4; ++c4) {
                const st8 hv = __builtin_bit_cast(st8, v[c4]);
#pragma unroll
                for (int e = 0; e < 8; ++e) part[Ss][e] = fmaf(w[c4], Traits<T>::to_acc(hv[e]), part[Ss][e]);
              }
#pragma unroll
              for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(part[s2][e]));
              __builtin_amdgcn_sched_barrier(0);
            }(), ...);
          }(std::make_integer_sequence<int, 2>{});
        }(), ...);
      }(std::make_integer_sequence<int, PL>{});
      // lane i keeps piece i of its own sum (its part[i]) and receives piece i ^ 1 from its partner's part[i]
      const bool odd = lane & 1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float own = odd ? part[1][e] : part[0][e];
        const float t0 = dpp_f<0xB1>(part[0][e]), t1 = dpp_f<0xB1>(part[1][e]);  // quad_perm [1,0,3,2]: the partner's
        const float other = odd ? t1 : t0;
        if constexpr (HALVES == 2) {                                          // rho = lane & 1: slot 0 is piece i
          acc[0][e] += own;
          acc[1][e] += other;
        } else {                                                               // slot j holds piece j ^ rho
          const bool sw = (((unsigned)lane & 1u) ^ rho) != 0u;
          acc[0][e] += sw ? other : own;
          acc[1][e] += sw ? own : other;
        }
      }
      pin_acc();
      __builtin_amdgcn_sched_barrier(0);
    };

    // stage the windows of one phase by LDS-DMA, ROW-WISE as msda_forward_col.hip does: one wave instruction copies 64
    // consecutive 16-byte pieces = 32 pixels of ONE window row, the waves take the rows in turn; a lane's source offset is
    // a per-lane part that depends on its column only (once per level and 32-pixel column block) plus a uniform per-row
    // part in the instruction's scalar offset -- no vector arithmetic per copy (the piece-linear indexing this replaces
    // divided by the window width for every piece).  Columns / rows outside the map use an out-of-range offset: zeros.
    auto stage_phase = [&](const int phx) {
      constexpr int PPX = PXB / 16, CPX = 64 / PPX;
      constexpr unsigned OOB = 0x80000000u;
      const int ln = tid & 63;
      const int dx = ln / PPX, cc = ln & (PPX - 1);
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
      const unsigned MDb = (unsigned)MD * 2u;                                 // bytes per pixel, all heads
#pragma unroll
      for (int l = 0; l < L; ++l) {
        if (phase_of[l] == phx && wsize[l] > 0) {
          const unsigned lvl_off = (unsigned)St[l] * MDb + (unsigned)(m * D) * 2u;
          const unsigned row_bytes = (unsigned)Ws[l] * MDb;
          unsigned char *dst = pool + (size_t)wbase[l] * PXB;
          for (int c0 = 0; c0 < wwid[l]; c0 += CPX) {
            const int xw = c0 + dx, x = wx0[l] + xw;
            const unsigned voff = (unsigned)x < (unsigned)Ws[l] ? (unsigned)x * MDb + (unsigned)(cc * 16) : OOB;
            if (xw < wwid[l]) {
              for (int r = wv; r < whgt[l]; r += NW) {
                const int y = wy0[l] + r;
                const bool in_y = (unsigned)y < (unsigned)Hs[l];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rsrc, (__attribute__((address_space(3))) void *)(dst + (size_t)(r * wwid[l] + c0) * PXB), 16,
                    (int)(in_y ? voff : OOB), (int)(in_y ? lvl_off + (unsigned)y * row_bytes : 0u), 0, 0);
              }
            }
          }
        }
      }
    };
    auto front_end = [&]() {
      if constexpr (FUSED) {                      // soft-max over the record's L * P logits (both lanes of a pair)
        float mx = -INFINITY;
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
          for (int k = 0; k < PL; ++k) mx = fmaxf(mx, wts[l][k]);
        if constexpr (HALVES == 2) mx = fmaxf(mx, dpp_f<0xB1>(mx));
        float sum = 0.f;
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
          for (int k = 0; k < PL; ++k) {
            wts[l][k] = __expf(wts[l][k] - mx);
            sum += wts[l][k];
          }
        if constexpr (HALVES == 2) sum += dpp_f<0xB1>(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
          for (int k = 0; k < PL; ++k) wts[l][k] *= inv;
      }
    };

    if (have_n && !starts_phase[L - 1]) decode(item_n, b_n, m_n, qv_n);
    auto level_step = [&](auto llc) {
      constexpr int ll = decltype(llc)::value;
      constexpr int l = col16_level_of_step<L>(ll);
      if (starts_phase[l]) {
        if (ll > 0) __syncthreads();
        stage_phase(phase_of[l]);
        if constexpr (ll == 0) {
          if (have_n) decode(item_n, b_n, m_n, qv_n);                          // while the LDS-DMA pieces are in flight
        }
        __syncthreads();
      }
      if constexpr (ll == 0) {
        if (fetched) {
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(f_new)::"memory");
          next_idx[1] = f_new;
        }
        front_end();
      }
      __builtin_amdgcn_s_setprio(ll == 0 ? 3 : 2);          // (the gather before the other workgroups' bookkeeping: msda_forward_col.hip)
      if (phase_of[l] >= 0) gather_level_lds(std::integral_constant<int, l>{});
      else gather_level_global(std::integral_constant<int, l>{});
      if (have_n) issue_loc_level(std::integral_constant<int, ll>{}, b_n, m_n, qv_n, raw);   // level ll of the next item
    };
    [&]<int... LLs>(std::integer_sequence<int, LLs...>) {
      (level_step(std::integral_constant<int, LLs>{}), ...);
    }(std::make_integer_sequence<int, L>{});
    __builtin_amdgcn_s_setprio(0);

    // ---- store: slot 0 of lane h holds piece rho = h (P = 8) -- add the partner's slot 1 -- or both pieces (P = 4) ----
    if constexpr (HALVES == 2) {
      st8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = Traits<T>::from_acc(acc[0][e] + dpp_f<0xB1>(acc[1][e]));
      if (valid) *reinterpret_cast<st8 *>(out + (rec_img + q) * MD + m * D + (int)rho * 8) = o;
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        st8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = Traits<T>::from_acc(acc[j][e]);
        if (valid)
          *reinterpret_cast<st8 *>(reinterpret_cast<unsigned char *>(out + (rec_img + q) * MD + m * D) + rot[j]) = o;
      }
    }

    item = item_n;
    have = have_n;
    b = b_n;
    m = m_n;
    qv = qv_n;
  }
}

// ---- launcher: returns -100 when this geometry is not covered (caller uses another kernel) ----------------------------
template <typename T>
int launch_msda_forward_col16(const void *value, const int64_t *shapes, const int64_t *starts, const void *loc,
                              const void *attn, int N, int S, int M, int D, int L, int Lq, int P, void *out,
                              hipStream_t stream, const float *ref, long long ref_batch_stride)
{
  using ST = typename Traits<T>::store_t;
  if ((((uintptr_t)value | (uintptr_t)out | (uintptr_t)loc | (uintptr_t)attn) & 15u)) return -100;
  if (ref && (((uintptr_t)ref) & 7u)) return -100;
  if (D != 16 || (P != 4 && P != 8) || L < 3 || L > 5 || Lq != S || M < 1) return -100;
  if ((long long)N * ((long long)S + 4096) * M >= 0x7fffffffLL) return -100;
  if ((long long)S * M * L * P * 8 >= 0xffffffffLL) return -100;               // 32-bit byte offsets inside an image
  constexpr int BLOCK = 256;
  static const int pool_kb = [] { const char *e = getenv("PCT_COL_POOL_KB"); const int v = e ? atoi(e) : 0;
                                  return (v >= 16 && v <= 52) ? v : 50; }();
  const int pool_px = pool_kb * 1024 / 32;
  const size_t lds = (size_t)pool_px * 32 + ((size_t)(BLOCK / 64) * L * 2 + 4) * sizeof(unsigned);
  const dim3 grid(256 * 3), block(BLOCK);
  unsigned *queue = win_queue_slot(stream);
  const ST *v = static_cast<const ST *>(value);
  const float *lc = static_cast<const float *>(loc), *at = static_cast<const float *>(attn);
  ST *o = static_cast<ST *>(out);
#define PCT_COL16(L_, P_, FU_)                                                                                          \
  hipLaunchKernelGGL((msda_forward_col16_kernel<T, L_, P_, FU_, BLOCK>), grid, block, lds, stream, v, shapes, starts,  \
                     lc, at, N, S, M, pool_px, o, ref, ref_batch_stride, queue)
#define PCT_COL16_L(P_, FU_)                 \
  do {                                       \
    if (L == 3) PCT_COL16(3, P_, FU_);       \
    else if (L == 4) PCT_COL16(4, P_, FU_);  \
    else PCT_COL16(5, P_, FU_);              \
  } while (0)
  if (ref) {
    if (P == 4) PCT_COL16_L(4, true);
    else PCT_COL16_L(8, true);
  } else {
    if (P == 4) PCT_COL16_L(4, false);
    else PCT_COL16_L(8, false);
  }
#undef PCT_COL16_L
#undef PCT_COL16
  return (int)hipGetLastError();
}

template int launch_msda_forward_col16<half_bits>(const void *, const int64_t *, const int64_t *, const void *,
                                                  const void *, int, int, int, int, int, int, int, void *, hipStream_t,
                                                  const float *, long long);
template int launch_msda_forward_col16<bf16_bits>(const void *, const int64_t *, const int64_t *, const void *,
                                                  const void *, int, int, int, int, int, int, int, void *, hipStream_t,
                                                  const float *, long long);

}  // namespace pct
