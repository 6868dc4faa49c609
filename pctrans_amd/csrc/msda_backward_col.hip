// MSDeformAttn backward, "pyramid-column" kernel for MI355X (gfx950, wave64): fp32, D = 16, P = 4, Lq == S.
//
// Same semantics as msda_backward.hip (reference: ops/src/cuda/ms_deform_im2col_cuda.cuh:92-164 per sample, channel sums
// as in cuh:306-408; host side ops/src/cuda/ms_deform_attn_cuda.cu:88-158).
//
// Why a second windowed backward: msda_backward_win.hip (round 1) works on 8 x 16 query tiles of ONE level.  A tile of a
// coarse level looking at a fine level has a box the pool cannot hold -- those levels are retried slot by slot or fall to
// direct global atomics -- and knock-out builds showed that neither the LDS adds (31 %) nor the flush (7 %) but this
// skeleton is what the kernel spends its time in (2.6 of 3.8 ms at P2, batch 32).  Here the work item is the forward's
// (msda_forward_col.hip): (image, pyramid COLUMN, head) -- the queries of all levels over one cell of a CX x CY grid, one
// window per level serving the whole column -- and one lane = one (query, head) with all 16 channels:
//   * pass 1 ("dots"): the value windows are staged by LDS-DMA with the forward's zero apron (the gate of cuh:290-296 is
//     folded into the coordinates), every sample reads its four head-pixels (rotated 16-byte pieces, as the forward) and
//     forms the four dot products <grad_out, v_corner> IN THE LANE -- no cross-lane butterflies -- from which
//     grad_attn_weight and grad_sampling_loc follow (cuh:115-163 rearranged: sum_c w_c d_c, W a (hh (d2 - d1) + lh (d4 - d3)), ...);
//   * pass 2 ("scatter"): the SAME pool then holds the grad_value windows, channel-pair planar, as 64-bit integer
//     accumulators: a packed FMA with the 1.5 * 2^23 magic constant leaves round(w_c a g scale) of two channels in the
//     low mantissa bits of a register pair, and ONE ds_add_u64 adds both raw bit patterns (measured: 6.4 cycles per wave
//     instruction against 2 x 4.3 for ds_add_u32, and no per-value integer subtraction).  A 17th plane counts the
//     contributions n per texel; the flush recovers both 32-bit sums exactly from (T, n): low = T_lo - n K, carry =
//     (n K + low) >> 32, high = T_hi - carry - n K (K = 0x4B400000).  Integer sums: deterministic, order-independent;
//     scale = one power of two per item from max|grad_out| max|attn|, 2^-21 of that bound per contribution, 1024
//     contributions cannot overflow; non-finite bounds send the item down the direct float path;
//   * same-address LDS atomics are served a lane per clock (8 x 8 queries of the finest level share one texel of the
//     coarsest): step t of lane i adds channel pair (t + i) % 8 and the lane rows take a level's points in rotated order;
//   * one coalesced flush per window: one float atomic per (texel, channel) with n > 0 and a non-zero sum, a wave
//     instruction covering 4 pixels x 16 channels (consecutive dwords), four such blocks' LDS reads in flight together;
//   * the two passes share one pool (values, then accumulators), so a workgroup holds 1092 window pixels in 74 KB: the four
//     windows of a column with model-like offsets (~1 300 px) take two phases, the reference's init-like ones a single one;
//   * levels whose box exceeds the pool (and items whose bounds are not finite) are only FLAGGED here -- one byte per item --
//     and a second launch of the same template (DIRECT) walks the items, skips those with an empty mask and does the
//     flagged levels on the direct path (global gathers + float atomics, the reference's summation order).  Inlined into
//     the main kernel that cold path cost 120 spilled registers whose reloads sat in the per-item planning.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <utility>

#include "msda_col_common.hpp"

namespace pct {

#ifndef PCT_BCOL_KO
#define PCT_BCOL_KO 0         /* knock-outs (WRONG RESULTS, timing only): 1 = no scatter adds, 2 = no dots reads, 4 = no flush atomics,
                                 16 = no dots pass, 32 = no scatter pass, 64 = no count adds, 128 = no zeroing */
#endif

#ifndef PCT_BCOL_PRIO
#define PCT_BCOL_PRIO 0         /* wave priorities (s_setprio): 1 = dots 3 / scatter 2 / rest 0, 2 = dots 2 / scatter 3, 3 = flush raised too */
#endif

#ifndef PCT_BCOL_STAMP
#define PCT_BCOL_STAMP 0        /* diagnostic build: per-part cycle sums of wave 0 (s_memtime), tools/stamp_msda_bwd.py */
#endif
#if PCT_BCOL_KO && !defined(PCT_EXPERIMENT_BUILD)
#error "PCT_BCOL_KO knock-outs give wrong results: add -DPCT_EXPERIMENT_BUILD to build one"
#endif
#define PCT_BSTR2(x) #x
#define PCT_BSTR(x) PCT_BSTR2(x)
const char *msda_backward_col_build_flags()
{
  return "bcol: KO=" PCT_BSTR(PCT_BCOL_KO) " PRIO=" PCT_BSTR(PCT_BCOL_PRIO) " STAMP=" PCT_BSTR(PCT_BCOL_STAMP);
}

constexpr int BCOL_BLOCK = 256;
constexpr int BCOL_GPX = 1092;                                  // pixels per pool (= 4 mod 32: the 8 planes of one pixel start 8 banks apart)
constexpr int BCOL_CNT_BYTES = BCOL_GPX * 4;                    // contribution counts, one dword per pixel
constexpr int BCOL_PLANE_BYTES = BCOL_GPX * 8;                  // one channel-pair plane of 64-bit accumulators
constexpr int BCOL_POOL_BYTES = (BCOL_CNT_BYTES + 8 * BCOL_PLANE_BYTES + 63) & ~63;   // >= BCOL_GPX * 64 (the value windows)
constexpr int BCOL_TAB_BYTES = 4096;                            // cell tables
constexpr size_t BCOL_FLAG_BYTES = 1u << 20;                 // flag bytes per buffer (+ 64 bytes: the `any` / `readers` words)
constexpr size_t BCOL_FLAG_STRIDE = BCOL_FLAG_BYTES + 64;
static_assert(BCOL_POOL_BYTES >= BCOL_GPX * 64, "value windows must fit the pool");
static_assert(BCOL_CNT_BYTES + 7 * BCOL_PLANE_BYTES + 8 < 65536, "ds immediate offsets");
static_assert((BCOL_PLANE_BYTES / 4) % 64 == 8, "plane stride in banks");

template <int L>
__host__ __device__ constexpr int bcol_level_of_step(const int ll)   // finest, coarsest, then the middle levels (as the forward)
{
  return ll == 0 ? L - 1 : (ll == 1 ? 0 : L - ll);
}

template <int L, bool DIRECT>
__global__ __launch_bounds__(BCOL_BLOCK, 2) void msda_backward_col_kernel(
    const float *__restrict__ grad_out, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const float *__restrict__ loc, const float *__restrict__ attn, const int N,
    const int S, const int M, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn, unsigned *__restrict__ queue, unsigned char *__restrict__ flags, const int flag_cap,
    unsigned long long *__restrict__ stamps)
{
  unsigned long long t_prev = 0, t_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int part) {
    if constexpr (PCT_BCOL_STAMP && !DIRECT) {
      __builtin_amdgcn_sched_barrier(0);
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if (part >= 0) t_sum[part] += t - t_prev;
      t_prev = t;
    }
  };

  constexpr int P = 4, D = 16, PXB = 64, BLOCK = BCOL_BLOCK, NW = BLOCK / 64;
  static_assert(L >= 1 && L <= 5, "unsupported geometry");
  typedef unsigned col_u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) unsigned long long lds_u64;
  typedef __attribute__((address_space(3))) unsigned lds_u32;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char *pool = smem_raw;                                              // value windows (64 B per head-pixel) / accumulators
  constexpr int POOLB = DIRECT ? 0 : BCOL_POOL_BYTES;                          // (the DIRECT launch has no windows)
  unsigned *tab = reinterpret_cast<unsigned *>(smem_raw + POOLB);             // cell tables
  unsigned *bb = tab + BCOL_TAB_BYTES / 4;                                     // [L][NW][2] per-wave boxes {min lo, ~max hi}
  unsigned *mx = bb + NW * 5 * 2;                                              // [NW][2] max |grad_out| bits, max |attn| bits
  unsigned *next_idx = mx + NW * 2;                                            // the workgroup's next item

  int tid = threadIdx.x;
  const int wave = tid >> 6;
  int lane = tid & 63;
  const int MD = M * D;
  if constexpr (DIRECT) {
    unsigned *aw_ = reinterpret_cast<unsigned *>(flags + BCOL_FLAG_BYTES);
    unsigned any = 0u;
    if (tid == 0) {
      any = __hip_atomic_load(aw_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (the value must be HERE before this workgroup counts itself in: the last one to count clears the word)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(any)::"memory");
      const unsigned r = atomicAdd(aw_ + 1, 1u);
      if (r + 1u == gridDim.x) {
        __hip_atomic_store(aw_, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(aw_ + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      next_idx[0] = any;
    }
    __syncthreads();
    if (next_idx[0] == 0u) return;
  }

  // ---- level geometry (uniform) -------------------------------------------------------------------------------------
  int Hs[L], Ws[L], St[L];
  col_f32x2 fWH[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
    St[l] = (int)starts[l];
    fWH[l] = uni_pair((float)Ws[l], (float)Hs[l]);
  }

  // ---- column grid and this lane's slot: exactly the forward's (msda_forward_col.hip) ---------------------------------
  int CX, CY;
  {
    int Hf = Hs[0], Wf = Ws[0];
#pragma unroll
    for (int l = 1; l < L; ++l)
      if (Hs[l] * Ws[l] > Hf * Wf) { Hf = Hs[l]; Wf = Ws[l]; }
    const float area = (float)BLOCK * (float)(Hf * Wf) / (float)S;
    const int side = (int)sqrtf(area);
    const int nxt = min(Wf, max(8, (side + 4) & ~7));
    CX = (Wf + nxt - 1) / nxt;
    const int nx0 = (Wf + CX - 1) / CX;
    const int nyt = max(1, (int)(area / (float)nx0));
    CY = min(Hf, (Hf + nyt - 1) / nyt);
    for (int guard = 0; guard < 4096; ++guard) {
      int maxq = 0;
#pragma unroll
      for (int l = 0; l < L; ++l) maxq += col_max_cell(Ws[l], CX) * col_max_cell(Hs[l], CY);
      if (maxq <= BLOCK) break;
      if (CY < Hf) ++CY;
      else if (CX < Wf) ++CX;
      else break;
    }
  }
  int lane_c0 = 0;
  unsigned lane_slot = 0x000FFFFFu;                                            // x | y << 10 | level << 20; x = y = 1023: no slot
  {
    int r = tid;
    bool placed = false;
#pragma unroll
    for (int ll = 0; ll < L; ++ll) {
      const int l = L - 1 - ll;
      const int mxc = col_max_cell(Ws[l], CX), myc = col_max_cell(Hs[l], CY);
      const int cnt = mxc * myc;
      if (!placed && r < cnt) {
        const int ly = r / max(mxc, 1), lx = r - ly * max(mxc, 1);
        lane_slot = (unsigned)lx | ((unsigned)ly << 10) | ((unsigned)l << 20);
        lane_c0 = St[l] + ly * Ws[l] + lx;
        placed = true;
      }
      r -= placed ? 0 : cnt;
    }
  }
  const bool use_tab = (L * (CX + CY) + L) * 4 <= BCOL_TAB_BYTES;
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
  constexpr int pool_use = BCOL_GPX;
  bool big_map = false;                                                        // a level too large for the 16-bit box corners
#pragma unroll
  for (int l = 0; l < L; ++l) big_map = big_map || Ws[l] > 65531 || Hs[l] > 65531;
  int ncol = CX * CY;
  if (!use_tab) {
    ncol = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) ncol += (Hs[l] * Ws[l] + BLOCK - 1) / BLOCK;
  }
  const int items = N * ncol * M;
  const UDiv dv_ncolM = make_udiv(ncol * M), dv_2ncol = make_udiv(2 * ncol), dv_CX = make_udiv(CX);

  int qi = lane & 3;
  const bool qi0 = qi & 1, qi1 = qi & 2;
  const unsigned rho = (unsigned)(lane >> 3) & 3u;                             // rotation of a head-pixel's 16-byte pieces
  unsigned rot[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rot[j] = ((j + rho) & 3u) << 4;                  // byte offset of the piece register j reads
  // The scatter's own rotations.  Many queries of a column hit the SAME texel of a coarse level in the same instruction
  // (8 x 8 queries of the 128^2 level share one pixel of the 16^2 level: the 8 neighbours of a lane row and the wave's 4
  // rows), and same-address LDS atomics are served one lane per clock (measured: 21 cycles per ds_add_u64 on average
  // instead of 6.4).  So step t of lane i adds channel pair (t + i) % 8 -- eight row neighbours, eight different planes --
  // and the lane rows (lane / 16) take a level's four points in rotated order -- four rows, four different samples.
  const unsigned r8 = (unsigned)lane & 7u, srot = (unsigned)(lane >> 4) & 3u;
  unsigned poff[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) poff[t] = (unsigned)BCOL_CNT_BYTES + ((t + r8) & 7u) * (unsigned)BCOL_PLANE_BYTES;

  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int chunk = (items + 7) / 8;
  const int item_end = min((xcd + 1) * chunk, items);
  const unsigned n_x = (unsigned)max(item_end - xcd * chunk, 0);

  constexpr int NPL = L * 2, NGL = (NPL + 3) / 4;                              // locations: 16-byte pieces, 64-byte groups
  constexpr int NGW = (L + 3) / 4;                                             // weights: one 16-byte piece per level

  // item -> (image, head) [uniform] and this lane's query (qv = q, or ~q of the query an idle lane shadows); item order
  // inside an image: head pair, column, head in the pair (the forward's)
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
        const int cnt = Hs[l] * Ws[l], nch = (cnt + BLOCK - 1) / BLOCK;
        if (c >= 0 && c < nch) {
          q = St[l] + c * BLOCK + tid;
          valid = c * BLOCK + tid < cnt;
        }
        c = c >= nch ? c - nch : -1;
      }
    }
    const unsigned long long vm = __ballot(valid);
    const int q_sh = vm ? __builtin_amdgcn_readlane(q, (int)__builtin_ctzll(vm)) : 0;
    qv_ = valid ? q : ~q_sh;
  };
  auto rec_index = [&](const int qv_, const int m_) {
    return __umul24((unsigned)(qv_ ^ (qv_ >> 31)), (unsigned)M) + (unsigned)m_;
  };
  auto quad_offsets = [&](const unsigned own_bytes, unsigned (&off)[4]) {
    off[0] = dpp_u<0x00>(own_bytes) + ((unsigned)((0 - qi) & 3) << 4);
    off[1] = dpp_u<0x55>(own_bytes) + ((unsigned)((1 - qi) & 3) << 4);
    off[2] = dpp_u<0xAA>(own_bytes) + ((unsigned)((2 - qi) & 3) << 4);
    off[3] = dpp_u<0xFF>(own_bytes) + ((unsigned)((3 - qi) & 3) << 4);
  };
  constexpr int RSRC_FLAGS = 0x00020000;
  auto issue_loc = [&](const int b_, const int m_, const int qv_, col_f32x4 (&raw)[NGL][4]) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(loc + (long long)b_ * S * M * (L * P * 2)), 0,
                                                      (int)((unsigned)S * (unsigned)M * (unsigned)(L * P * 8)), RSRC_FLAGS);
    unsigned off[4];
    quad_offsets(rec_index(qv_, m_) * (unsigned)(L * P * 8), off);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int g = 0; g < NGL; ++g)
        raw[g][s4] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off[s4] + g * 64), 0, 0));
  };

  // ---- pieces shared by the main kernel and the DIRECT one ------------------------------------------------------------------
  // the record -> pixel coordinates (w_im, h_im) = loc * (W, H) - 0.5 (cuh:283-288) with the gate of cuh:290-296 folded in as
  // in the forward: a coordinate that fails its test (NaN included) moves to -2, one past the map to W; the staged window
  // carries a zero apron of two pixels, so such a sample reads zeros (all its gradients are exactly 0 for finite grad_out)
  // and adds into apron accumulators that are never flushed
  auto form_lxy = [&](col_f32x4 (&raw_)[NGL][4], col_f32x2 (&lxy_)[L][P]) {
#pragma unroll
    for (int g = 0; g < NGL; ++g) quad_transpose_in(raw_[g], qi0, qi1);
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int k = 0; k < P; ++k) {
        const col_f32x4 pc = raw_[(l * 2 + k / 2) / 4][(l * 2 + k / 2) & 3];
        const col_f32x2 v = {pc[(k & 1) * 2], pc[(k & 1) * 2 + 1]};
        const col_f32x2 px = __builtin_elementwise_fma(v, fWH[l], col_f32x2{-0.5f, -0.5f});
        lxy_[l][k][0] = fminf(px[0] > -1.f ? px[0] : -2.f, fWH[l][0]);
        lxy_[l][k][1] = fminf(px[1] > -1.f ? px[1] : -2.f, fWH[l][1]);
      }
  };
  // an item's weights and grad_output rows, quad-cooperatively
  auto load_wg = [&](const int b_, const unsigned own_, col_f32x4 (&wraw_)[NGW][4], col_f32x4 (&graw_)[4]) {
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(attn + (long long)b_ * S * M * (L * P)), 0,
                                                       (int)((unsigned)S * (unsigned)M * (unsigned)(L * P * 4)), RSRC_FLAGS);
    const auto rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(grad_out + (long long)b_ * S * MD), 0,
                                                       (int)((unsigned)S * (unsigned)MD * 4u), RSRC_FLAGS);
    unsigned off[4], ofg[4];
    quad_offsets(own_ * (unsigned)(L * P * 4), off);
    quad_offsets(own_ * (unsigned)(D * 4), ofg);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
      for (int g = 0; g < NGW; ++g)
        wraw_[g][s4] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)(off[s4] + g * 64), 0, 0));
      graw_[s4] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsg, (int)ofg[s4], 0, 0));
    }
  };
  // ... transposed; grad_output rotated like the head-pixel pieces (register j: piece (j + rho) % 4); an idle lane's are zero
  auto front = [&](col_f32x4 (&wraw_)[NGW][4], col_f32x4 (&graw_)[4], const bool idle_, float (&wts_)[L][P],
                   col_f32x2 (&go2_)[4][2]) {
#pragma unroll
    for (int g = 0; g < NGW; ++g) quad_transpose_in(wraw_[g], qi0, qi1);
    quad_transpose_in(graw_, qi0, qi1);
    rot_regs(graw_, rho & 1u, rho & 2u);
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int k = 0; k < P; ++k) wts_[l][k] = idle_ ? 0.f : wraw_[l / 4][l & 3][k];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) go2_[j][e >> 1][e & 1] = idle_ ? 0.f : graw_[j][e];
  };
  // grad_sampling_loc (128 B) and grad_attn_weight (64 B) records are written level by level as soon as a level's results
  // exist (48 result registers would otherwise live through the item); an idle lane's stores go out of the descriptors' range
  auto store_level = [&](const int b_, const unsigned own_, const int qv_, const int l, const float (&ga)[P],
                         const col_f32x2 (&gl)[P]) {
    const auto rsl = __builtin_amdgcn_make_buffer_rsrc(grad_loc + (long long)b_ * S * M * (L * P * 2), 0,
                                                       (int)((unsigned)S * (unsigned)M * (unsigned)(L * P * 8)), RSRC_FLAGS);
    const auto rsa = __builtin_amdgcn_make_buffer_rsrc(grad_attn + (long long)b_ * S * M * (L * P), 0,
                                                       (int)((unsigned)S * (unsigned)M * (unsigned)(L * P * 4)), RSRC_FLAGS);
    const unsigned dead = (unsigned)(qv_ >> 31) & 0x80000000u;
    const unsigned ol = (own_ * (unsigned)(L * P * 8)) | dead, oa = (own_ * (unsigned)(L * P * 4)) | dead;
    const col_f32x4 x0 = {gl[0][0], gl[0][1], gl[1][0], gl[1][1]};
    const col_f32x4 x1 = {gl[2][0], gl[2][1], gl[3][0], gl[3][1]};
    const col_f32x4 xa = {ga[0], ga[1], ga[2], ga[3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(col_u32x4, x0), rsl, (int)(ol + l * 32), 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(col_u32x4, x1), rsl, (int)(ol + l * 32 + 16), 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(col_u32x4, xa), rsa, (int)(oa + l * 16), 0, 0);
  };

  // More items than the flag buffer holds (the host cannot know: the level shapes live in device memory; it takes a pyramid
  // of thousands of tiny columns): the main launch does nothing and the DIRECT one does every level of every item.
  const bool overflow = items > flag_cap;
  // Two words behind the flag bytes: `any` (the main launch sets it when it leaves a level -- or, on overflow, everything --
  // to the DIRECT one) and `readers`.  Every workgroup of the DIRECT launch reads `any` first and then counts itself in;
  // the last one to do so clears both words for the buffer's next use.  With nothing flagged -- the normal case -- the
  // DIRECT launch ends right there (it used to derive the column grid and walk the flags first: 10.7 us per call, now 7.8).
  unsigned *anyw = reinterpret_cast<unsigned *>(flags + BCOL_FLAG_BYTES);
  if constexpr (!DIRECT) {
    if (overflow) {
      if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(anyw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  if constexpr (DIRECT) {
    // ---- the DIRECT launch: flagged levels only, global gathers + float atomics ----------------------------------------------
    // The QUAD works on one member's sample at a time (as the forward's global-memory levels): the member's geometry is
    // broadcast by DPP and lane c takes the 16-byte piece (s - c) % 4 of every corner -- the quad reads and atomically adds
    // whole 64-byte head-pixels, 16 tag look-ups per wave instruction instead of 64 (lane by lane with all 16 channels the
    // float atomics ran four times slower).  That piece of the member's grad_output row is exactly what the quad-cooperative
    // load delivers before its transposition, so grad_output is not transposed at all.  Per channel the reference's
    // summation order (cuh:104-163); the three channel sums of a sample are reduced across the quad afterwards.
    for (int it = blockIdx.x; it < items; it += (int)gridDim.x) {
      const unsigned mask = overflow ? (1u << L) - 1u : (unsigned)__builtin_amdgcn_readfirstlane((int)flags[it]);
      if (mask == 0u) continue;
      int b, m, qv;
      decode(it, b, m, qv);
      const unsigned own = rec_index(qv, m);
      const bool idle = qv < 0;
      col_f32x4 raw[NGL][4], wraw[NGW][4], gq[4];                              // gq[s]: piece (s - qi) % 4 of member s's grad_output
      issue_loc(b, m, qv, raw);
      load_wg(b, own, wraw, gq);
      col_f32x2 lxy[L][P];
      float wts[L][P];
      form_lxy(raw, lxy);
#pragma unroll
      for (int g = 0; g < NGW; ++g) quad_transpose_in(wraw[g], qi0, qi1);
#pragma unroll
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < P; ++k) wts[l][k] = wraw[l / 4][l & 3][k];
      const float *vimg = value + (long long)b * S * MD + m * D;
      float *gimg = grad_value + (long long)b * S * MD + m * D;
      [&]<int... Ls>(std::integer_sequence<int, Ls...>) {
        ([&] {
          constexpr int l = Ls;
          if (!((mask >> l) & 1u)) return;
          const int H = Hs[l], W = Ws[l];
          float ga[P];
          col_f32x2 gl[P];
          [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
            ([&] {
              constexpr int k = Ks;
              asm volatile("" : "+v"(lxy[l][k]));
              const col_f32x2 pix = lxy[l][k];
              const int x0 = cvt_flr(pix[0]), y0 = cvt_flr(pix[1]);                // inside [-2, W] x [-2, H]
              const float o_lw = __builtin_amdgcn_fractf(pix[0]), o_lh = __builtin_amdgcn_fractf(pix[1]);
              const bool top = (unsigned)y0 < (unsigned)H, bot = (unsigned)(y0 + 1) < (unsigned)H;
              const bool lft = (unsigned)x0 < (unsigned)W, rgt = (unsigned)(x0 + 1) < (unsigned)W;
              // valid corners as bits; none for an idle lane or a gated-out sample (the reference skips it: exact zeros)
              const unsigned o_ok = idle ? 0u : ((top && lft ? 1u : 0u) | (top && rgt ? 2u : 0u) | (bot && lft ? 4u : 0u) | (bot && rgt ? 8u : 0u));
              const int o_e0 = (St[l] + y0 * W + x0) * MD;                          // (launcher: S * M * D < 2^31)
              const float o_aw = wts[l][k];
              float pa[4], pw[4], ph[4];                                           // [member]: my 4 channels' share of its sums
              [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
                ([&] {
                  constexpr int s4 = Ss;
                  constexpr int CT = BcastCtrl<4, s4>::value;
                  const unsigned ok = dpp_u<CT>(o_ok);
                  const int e0 = dpp_i<CT>(o_e0);
                  const float lw = dpp_f<CT>(o_lw), lh = dpp_f<CT>(o_lh), aw = dpp_f<CT>(o_aw);
                  float s_a = 0.f, s_w = 0.f, s_h = 0.f;
                  if (ok != 0u) {
                    const float hw = 1.f - lw, hh = 1.f - lh;
                    const float wc[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                    const int pc = ((s4 - qi) & 3) * 4;
                    const int eo[4] = {e0 + pc, e0 + MD + pc, e0 + W * MD + pc, e0 + W * MD + MD + pc};
                    col_f32x4 v[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                      v[c] = ((ok >> c) & 1u) ? *reinterpret_cast<const col_f32x4 *>(vimg + eo[c]) : col_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                      const float g = gq[s4][e];
                      const float tgv = g * aw;                                    // top_grad_value (cuh:112)
                      float gh = -(hw * v[0][e]);
                      gh = fmaf(-lw, v[1][e], gh);
                      gh = fmaf(hw, v[2][e], gh);
                      gh = fmaf(lw, v[3][e], gh);
                      float gw = -(hh * v[0][e]);
                      gw = fmaf(hh, v[1][e], gw);
                      gw = fmaf(-lh, v[2][e], gw);
                      gw = fmaf(lh, v[3][e], gw);
                      float val = wc[0] * v[0][e];
                      val = fmaf(wc[1], v[1][e], val);
                      val = fmaf(wc[2], v[2][e], val);
                      val = fmaf(wc[3], v[3][e], val);
                      s_a = fmaf(g, val, s_a);
                      s_w = fmaf(gw, tgv, s_w);
                      s_h = fmaf(gh, tgv, s_h);
#pragma unroll
                      for (int c = 0; c < 4; ++c)
                        if ((ok >> c) & 1u) unsafeAtomicAdd(gimg + eo[c] + e, wc[c] * tgv);
                    }
                  }
                  pa[s4] = s_a;
                  pw[s4] = s_w;
                  ph[s4] = s_h;
                  __builtin_amdgcn_sched_barrier(0);
                }(), ...);
              }(std::make_integer_sequence<int, 4>{});
              // lane i <- sum over the quad's lanes of their share of member i's sums (two exchange steps)
              auto quad_reduce = [&](const float (&t)[4]) {
                const float k0 = qi0 ? t[1] : t[0], g0 = qi0 ? t[0] : t[1];       // keep members of my parity, give the others
                const float k1 = qi0 ? t[3] : t[2], g1 = qi0 ? t[2] : t[3];
                const float u0 = k0 + dpp_f<0xB1>(g0), u1 = k1 + dpp_f<0xB1>(g1);  // quad_perm [1,0,3,2]
                const float kk = qi1 ? u1 : u0, gg = qi1 ? u0 : u1;
                return kk + dpp_f<0x4E>(gg);                                       // quad_perm [2,3,0,1]
              };
              const float r_a = quad_reduce(pa), r_w = quad_reduce(pw), r_h = quad_reduce(ph);
              ga[k] = o_ok ? r_a : 0.f;
              gl[k][0] = o_ok ? r_w * fWH[l][0] : 0.f;
              gl[k][1] = o_ok ? r_h * fWH[l][1] : 0.f;
            }(), ...);
          }(std::make_integer_sequence<int, P>{});
          store_level(b, own, qv, l, ga, gl);
        }(), ...);
      }(std::make_integer_sequence<int, L>{});
    }
    return;
  }

  int item = xcd * chunk + slot0;
  bool have = item < item_end;
  int b = 0, m = 0, qv = 0;
  col_f32x4 raw[NGL][4];
  if (have) {
    decode(item, b, m, qv);
    issue_loc(b, m, qv, raw);
  }

  stamp(-1);
  while (have) {
    asm volatile("" : "+v"(tid), "+v"(qi), "+v"(lane_slot), "+v"(lane_c0));
    unsigned rfetch = 0u;
    if (queue && tid == 0) rfetch = atomicAdd(queue + xcd * WIN_QUEUE_STRIDE, 1u);
    const unsigned own = rec_index(qv, m);                                    // this lane's record: query * M + head
    const bool idle = qv < 0;

    col_f32x2 lxy[L][P];
    form_lxy(raw, lxy);

    // ---- pre-pass: per-level bounding box (the forward's) ---------------------------------------------------------------
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float mnx = fminf(fminf(lxy[l][0][0], lxy[l][1][0]), fminf(lxy[l][2][0], lxy[l][3][0]));
      const float mxx = fmaxf(fmaxf(lxy[l][0][0], lxy[l][1][0]), fmaxf(lxy[l][2][0], lxy[l][3][0]));
      const float mny = fminf(fminf(lxy[l][0][1], lxy[l][1][1]), fminf(lxy[l][2][1], lxy[l][3][1]));
      const float mxy = fmaxf(fmaxf(lxy[l][0][1], lxy[l][1][1]), fmaxf(lxy[l][2][1], lxy[l][3][1]));
      const unsigned lo = (unsigned)(cvt_flr(mnx) + 2) | ((unsigned)(cvt_flr(mny) + 2) << 16);
      const unsigned hi = (unsigned)(cvt_flr(mxx) + 3) | ((unsigned)(cvt_flr(mxy) + 3) << 16);
      const unsigned red = wave_reduce_box(lo, hi);                 // lane 31: min lo, lane 63: ~max hi
      if ((lane & 31) == 31) bb[(l * NW + wave) * 2 + (lane >> 5)] = red;
    }
    // ---- this item's weights and grad_output rows (quad-cooperative, looked at after the staging is issued) ----------------
    col_f32x4 wraw[NGW][4], graw[4];
    load_wg(b, own, wraw, graw);
    if (tid == 0) {
      unsigned fetched = (unsigned)(item - xcd * chunk + nslots);             // static stride when there is no queue
      if (queue) {
        if (rfetch + 1u >= n_x) atomicExch(queue + xcd * WIN_QUEUE_STRIDE, 0u);                   // that was the launch's last fetch
        fetched = (unsigned)nslots + rfetch;
      }
      next_idx[0] = fetched;
    }
    stamp(0);
    __syncthreads();                                                          // (A) boxes visible; pool free (previous flush read)
    stamp(1);
    int item_n, b_n = 0, m_n = 0, qv_n = 0;
    {
      const unsigned nxt = __builtin_amdgcn_readfirstlane(next_idx[0]);
      item_n = nxt < n_x ? xcd * chunk + (int)nxt : item_end;
    }
    const bool have_n = item_n < item_end;

    // ---- windows and phases (uniform; the forward's planning) ---------------------------------------------------------------
    int wx0[L], wy0[L], wwid[L], whgt[L], wsize[L], wbase[L], phase_of[L], woff[L];
    int nph = 0;
    {
#pragma unroll
      for (int l = 0; l < L; ++l) {
        unsigned lo, hi;
        block_box<NW>(bb + l * NW * 2, lo, hi);
        const unsigned lx = lo & 0xFFFFu, ly = lo >> 16;
        wx0[l] = (int)lx - 2;
        wy0[l] = (int)ly - 2;
        wwid[l] = (int)((hi & 0xFFFFu) - lx) + 1;
        whgt[l] = (int)((hi >> 16) - ly) + 1;
        wsize[l] = wwid[l] * whgt[l];
      }
      int ph = 0, used = 0;
      bool any = false;
#pragma unroll
      for (int ll = 0; ll < L; ++ll) {
        const int l = bcol_level_of_step<L>(ll);
        if (wsize[l] > pool_use || big_map) {
          phase_of[l] = -1;
          wbase[l] = 0;
          continue;
        }
        if (used + wsize[l] > pool_use) {
          ++ph;
          used = 0;
        }
        phase_of[l] = ph;
        wbase[l] = used;
        used += wsize[l];
        any = true;
      }
      nph = any ? ph + 1 : 0;
#pragma unroll
      for (int l = 0; l < L; ++l) woff[l] = wbase[l] - wy0[l] * wwid[l] - wx0[l];
    }
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value + (long long)b * S * MD), 0,
                                                        (int)((unsigned)S * (unsigned)MD * 4u), RSRC_FLAGS);
    float *gimg = grad_value + (long long)b * S * MD + m * D;                 // this image, this head

    // stage the value windows of one phase by LDS-DMA (the forward's stage_phase)
    auto stage_phase = [&](const int phx) {
      constexpr int PPX = PXB / 16, CPX = 64 / PPX;
      constexpr unsigned OOB = 0x80000000u;
      const int ln = tid & 63;
      const int dx = ln / PPX, cc = ln & (PPX - 1);
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
      const unsigned MDb = (unsigned)MD * 4u;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        if (phase_of[l] == phx && wsize[l] > 0) {
          const unsigned lvl_off = (unsigned)St[l] * MDb + (unsigned)(m * D) * 4u;
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

    if (nph > 0) stage_phase(0);
    if (have_n) decode(item_n, b_n, m_n, qv_n);                                // while the LDS-DMA pieces are in flight

    // ---- weights, grad_output (rotated like the head-pixel pieces), the item's magnitude bounds -------------------------------
    float wts[L][P];
    col_f32x2 go2[4][2];                                                       // register j: channels of piece (j + rho) % 4
    {
      front(wraw, graw, idle, wts, go2);
      unsigned gmax = 0u, amax = 0u;                                           // |x| as bits: Inf / NaN compare largest
#pragma unroll
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < P; ++k) amax = max(amax, __float_as_uint(wts[l][k]) & 0x7fffffffu);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          gmax = max(gmax, __float_as_uint(go2[j][e][0]) & 0x7fffffffu);
          gmax = max(gmax, __float_as_uint(go2[j][e][1]) & 0x7fffffffu);
        }
      gmax = wave_reduce_umax(gmax);
      amax = wave_reduce_umax(amax);
      if (lane == 0) {
        mx[wave * 2] = gmax;
        mx[wave * 2 + 1] = amax;
      }
    }
    stamp(2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                                          // (B1) phase 0 staged; bounds visible
    stamp(3);
    bool fixed_ok;
    float scale, inv_scale;
    {
      unsigned gb = 0u, ab = 0u;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        gb = max(gb, mx[w * 2]);
        ab = max(ab, mx[w * 2 + 1]);
      }
      gb = __builtin_amdgcn_readfirstlane(gb);
      ab = __builtin_amdgcn_readfirstlane(ab);
      const float C = __builtin_bit_cast(float, gb) * __builtin_bit_cast(float, ab);
      int e;
      (void)frexpf(C, &e);                                                    // C < 2^e
      fixed_ok = gb < 0x7f800000u && ab < 0x7f800000u && C < 1e30f;
      // a window dword receives at most BLOCK * P = 1024 contributions w_corner * attn * grad_out, each below C < 2^e:
      // rounded to multiples of 2^(e - 21) every contribution is below 2^21 in magnitude and the int32 sums stay inside
      // (-2^31, 2^31) (a contribution also has to stay below 2^22 for the magic-constant rounding)
      const int shift = max(min(21 - e, 100), -100);
      scale = uni(ldexpf(1.f, shift));
      inv_scale = uni(ldexpf(1.f, -shift));
    }
    // ---- pass 1 of a level from its LDS window: four dot products per sample ---------------------------------------------
    auto dots_lds = [&](auto lc) {
      constexpr int l = decltype(lc)::value;
      float ga[P];
      col_f32x2 gl[P];
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
        ([&] {
          constexpr int k = Ks;
          // (opaque: the compiler otherwise forms every sample's geometry at the top of the item and keeps it alive)
          asm volatile("" : "+v"(lxy[l][k]));
          const col_f32x2 pix = lxy[l][k];
          const int x0 = cvt_flr(pix[0]), y0 = cvt_flr(pix[1]);
          const float lw = __builtin_amdgcn_fractf(pix[0]), lh = __builtin_amdgcn_fractf(pix[1]);
          const float hw = 1.f - lw, hh = 1.f - lh;
          const unsigned a = (unsigned)(__mul24(y0, wwid[l]) + x0 + woff[l]) << 6;
          const unsigned rowb = (unsigned)wwid[l] << 6;
          float d[4];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            col_f32x4 va[4], vb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned char *pa = pool + (a + (r ? rowb : 0u) + rot[j]);
              if (PCT_BCOL_KO & 2) {
                va[j] = col_f32x4{(float)a, 1.f, 2.f, 3.f};
                vb[j] = va[j];
              } else {
                va[j] = *reinterpret_cast<const col_f32x4 *>(pa);
                vb[j] = *reinterpret_cast<const col_f32x4 *>(pa + PXB);
              }
            }
            col_f32x2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                sa = __builtin_elementwise_fma(go2[j][e], col_f32x2{va[j][2 * e], va[j][2 * e + 1]}, sa);
                sb = __builtin_elementwise_fma(go2[j][e], col_f32x2{vb[j][2 * e], vb[j][2 * e + 1]}, sb);
              }
            d[2 * r] = sa[0] + sa[1];
            d[2 * r + 1] = sb[0] + sb[1];
            __builtin_amdgcn_sched_barrier(0);
          }
          // cuh:115-163 with the channel sums taken first: d_c = <grad_out, v_c>
          ga[k] = hh * (hw * d[0] + lw * d[1]) + lh * (hw * d[2] + lw * d[3]);
          // (opaque: the products aw * (W, H) do not depend on the phase, and hoisted out of the phase loop they were 32
          // registers, spilled and reloaded once per sample)
          float aw = wts[l][k];
          asm volatile("" : "+v"(aw));
          gl[k][0] = aw * fWH[l][0] * (hh * (d[1] - d[0]) + lh * (d[3] - d[2]));
          gl[k][1] = aw * fWH[l][1] * (hw * (d[2] - d[0]) + lw * (d[3] - d[1]));
        }(), ...);
      }(std::make_integer_sequence<int, P>{});
      store_level(b, own, qv, l, ga, gl);
    };
    // ---- pass 2: round(w_c * attn * grad_out * scale) of two channels per 64-bit add, one count per corner ---------------
    auto scatter_impl = [&](auto lc, const col_f32x2 (&gor)[8]) {
      constexpr int l = decltype(lc)::value;
      const col_f32x2 magic2 = {12582912.f, 12582912.f};                      // 1.5 * 2^23
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
        ([&] {
          constexpr int k = Ks;
          // this lane row's point of step k (see srot above): (k + srot) % 4
          const unsigned idx = ((unsigned)k + srot) & 3u;
          const bool i0 = idx & 1u, i1 = idx & 2u;
          auto sel4 = [&](const float a0, const float a1, const float a2, const float a3) {
            const float lo = i0 ? a1 : a0, hi = i0 ? a3 : a2;
            return i1 ? hi : lo;
          };
          col_f32x2 pix = {sel4(lxy[l][0][0], lxy[l][1][0], lxy[l][2][0], lxy[l][3][0]),
                           sel4(lxy[l][0][1], lxy[l][1][1], lxy[l][2][1], lxy[l][3][1])};
          float aw = sel4(wts[l][0], wts[l][1], wts[l][2], wts[l][3]);
          asm volatile("" : "+v"(pix), "+v"(aw));
          const int x0 = cvt_flr(pix[0]), y0 = cvt_flr(pix[1]);
          const float lw = __builtin_amdgcn_fractf(pix[0]), lh = __builtin_amdgcn_fractf(pix[1]);
          const float as = aw * scale;
          const col_f32x2 t2 = col_f32x2{1.f - lw, lw} * col_f32x2{as, as};
          const col_f32x2 g12 = t2 * col_f32x2{1.f - lh, 1.f - lh}, g34 = t2 * col_f32x2{lh, lh};
          const unsigned pidx = (unsigned)(__mul24(y0, wwid[l]) + x0 + woff[l]);
          if (!(PCT_BCOL_KO & (1 | 64))) {
            lds_u32 *cp = (lds_u32 *)(pool + pidx * 4u);
            lds_u32 *cq = (lds_u32 *)(pool + (pidx + (unsigned)wwid[l]) * 4u);
            __hip_atomic_fetch_add(cp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(cp + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(cq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(cq + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const unsigned pr = (pidx + (r ? (unsigned)wwid[l] : 0u)) * 8u;
            const col_f32x2 wa = r ? col_f32x2{g34[0], g34[0]} : col_f32x2{g12[0], g12[0]};
            const col_f32x2 wb = r ? col_f32x2{g34[1], g34[1]} : col_f32x2{g12[1], g12[1]};
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const col_f32x2 xa = __builtin_elementwise_fma(wa, gor[t], magic2);
              const col_f32x2 xb = __builtin_elementwise_fma(wb, gor[t], magic2);
              if (!(PCT_BCOL_KO & 1)) {
                lds_u64 *q = (lds_u64 *)(pool + (pr + poff[t]));
                __hip_atomic_fetch_add(q, __builtin_bit_cast(unsigned long long, xa), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(q + 1, __builtin_bit_cast(unsigned long long, xb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              } else {
                asm volatile("" ::"v"(xa), "v"(xb));
              }
              // (without the fence the compiler forms the 64 packed products of a sample -- of several samples -- first
              // and spills them on the way to the adds)
              if (t & 1) __builtin_amdgcn_sched_barrier(0);
            }
          }
        }(), ...);
      }(std::make_integer_sequence<int, P>{});
    };
    // ---- flush of one window: (T, n) -> the two 32-bit sums -> two float atomics per (texel, channel pair) with n > 0 -------
    auto flush_window = [&](const int l) {
      // (opaque per call: the per-lane pointers and plane offsets below do not depend on the phase, and hoisted out of the
      // phase loop -- four levels of them -- they were registers taken from the dots and scatter passes)
      asm volatile("" : "+v"(lane));
      // a wave takes every NW-th window row; one wave instruction covers 4 pixels x 16 channels: ONE atomic instruction whose
      // lanes sit on consecutive dwords of grad_value (whole 64-byte head-pixels; two lanes share an accumulator pair and take
      // its low / high field).  Rows and columns outside the map (the apron) and untouched texels (n = 0) are skipped.
      // Four column blocks at a time with all their LDS reads (counts and accumulators, unconditionally) in flight together:
      // one block per round trip -- count, branch, accumulator, decode, atomic -- took 800 cycles per block, a third of an item.
      const int H = Hs[l], W = Ws[l];
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int ch = lane & 15, dxl = lane >> 4;
      float *glev = gimg + (long long)St[l] * MD + ch;
      const unsigned plane_b = (unsigned)BCOL_CNT_BYTES + (unsigned)(ch >> 1) * (unsigned)BCOL_PLANE_BYTES;
      constexpr int UF = 4;
      for (int r = wv; r < whgt[l]; r += NW) {
        const int y = wy0[l] + r;
        if ((unsigned)y >= (unsigned)H) continue;
        const int rowi = wbase[l] + r * wwid[l];
        float *grow = glev + (long long)y * W * MD;
        for (int c0 = 0; c0 < wwid[l]; c0 += 4 * UF) {
          unsigned n[UF];
          unsigned long long T[UF];
          int xs[UF];
#pragma unroll
          for (int u = 0; u < UF; ++u) {
            const int xw = c0 + 4 * u + dxl, x = wx0[l] + xw;
            const bool okx = xw < wwid[l] && (unsigned)x < (unsigned)W;
            xs[u] = okx ? x : -1;
            const int idx = rowi + (okx ? xw : 0);                               // (a pixel of this row in any case)
            n[u] = *reinterpret_cast<const unsigned *>(pool + (size_t)idx * 4);
            T[u] = *reinterpret_cast<const unsigned long long *>(pool + plane_b + (size_t)idx * 8);
          }
#pragma unroll
          for (int u = 0; u < UF; ++u) {
            if (xs[u] >= 0 && n[u] != 0u) {
              constexpr unsigned K = 0x4B400000u;
              const unsigned nK = n[u] * K;
              const int lo = (int)((unsigned)T[u] - nK);
              const unsigned long long U = (unsigned long long)n[u] * K + (unsigned long long)(long long)lo;
              const int hi = (int)((unsigned)(T[u] >> 32) - (unsigned)(U >> 32) - nK);
              const int mine = (ch & 1) ? hi : lo;
              if (!(PCT_BCOL_KO & 4)) {
                if (mine != 0) unsafeAtomicAdd(grow + (long long)xs[u] * MD, (float)mine * inv_scale);
              }
            }
          }
        }
      }
    };
#pragma unroll 1
    for (int ph = 0; ph < nph; ++ph) {
      if (ph > 0) {
        __syncthreads();                                                      // the previous phase's flush has read the pool
        stage_phase(ph);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp(9);
      }
      if (fixed_ok) {
        if (PCT_BCOL_PRIO) __builtin_amdgcn_s_setprio(PCT_BCOL_PRIO == 2 ? 2 : 3);
        [&]<int... Ls>(std::integer_sequence<int, Ls...>) {
          ([&] { if (!(PCT_BCOL_KO & 16) && phase_of[Ls] == ph) dots_lds(std::integral_constant<int, Ls>{}); }(), ...);
        }(std::make_integer_sequence<int, L>{});
        if (PCT_BCOL_PRIO) __builtin_amdgcn_s_setprio(0);
        stamp(4);
        __syncthreads();                                                      // (B2) every wave is done with the values
        if (ph == nph - 1 && have_n) issue_loc(b_n, m_n, qv_n, raw);           // in flight until the next iteration
        int used = 0;
#pragma unroll
        for (int l = 0; l < L; ++l) used += phase_of[l] == ph ? wsize[l] : 0;
        if (!(PCT_BCOL_KO & 128)) {
          const col_f32x4 z = {0.f, 0.f, 0.f, 0.f};
          for (int i = tid; i * 4 < used; i += BLOCK) *reinterpret_cast<col_f32x4 *>(pool + (size_t)i * 16) = z;   // counts
#pragma unroll
          for (int p = 0; p < 8; ++p)
            for (int i = tid; i * 2 < used; i += BLOCK)
              *reinterpret_cast<col_f32x4 *>(pool + BCOL_CNT_BYTES + (size_t)p * BCOL_PLANE_BYTES + (size_t)i * 16) = z;
        }
        __syncthreads();                                                      // (B3) accumulators zeroed
        col_f32x2 gor[8];                                                     // register t: channel pair (t + r8) % 8
      // go2 holds pair (u + 2 * rho) % 8 in slot u = 2 * j + e; the scatter wants pair (t + r8) % 8 in register t: rotate by
        // the remaining amount d = (r8 - 2 * rho) % 8 (three layers of selects, once per item)
        {
          const unsigned d = (r8 - 2u * rho) & 7u;
          col_f32x2 a[8], c[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] = go2[u >> 1][u & 1];
#pragma unroll
          for (int u = 0; u < 8; ++u) c[u] = (d & 1u) ? a[(u + 1) & 7] : a[u];
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] = (d & 2u) ? c[(u + 2) & 7] : c[u];
#pragma unroll
          for (int u = 0; u < 8; ++u) gor[u] = (d & 4u) ? a[(u + 4) & 7] : a[u];
        }
        auto scatter_lds = [&](auto lc) { scatter_impl(lc, gor); };
        if (PCT_BCOL_PRIO) __builtin_amdgcn_s_setprio(PCT_BCOL_PRIO == 2 ? 3 : 2);
        stamp(5);
        [&]<int... Ls>(std::integer_sequence<int, Ls...>) {
          ([&] { if (!(PCT_BCOL_KO & 32) && phase_of[Ls] == ph) scatter_lds(std::integral_constant<int, Ls>{}); }(), ...);
        }(std::make_integer_sequence<int, L>{});
        if (PCT_BCOL_PRIO) __builtin_amdgcn_s_setprio(PCT_BCOL_PRIO == 3 ? 1 : 0);
        stamp(6);
        __syncthreads();                                                      // (B4) every add of this phase is in the pool
        stamp(7);
#pragma unroll
        for (int l = 0; l < L; ++l)
          if (phase_of[l] == ph) flush_window(l);
        if (PCT_BCOL_PRIO == 3) __builtin_amdgcn_s_setprio(0);
        stamp(8);
      }
    }
    if ((nph == 0 || !fixed_ok) && have_n) issue_loc(b_n, m_n, qv_n, raw);
    // levels this launch did not do: a level whose box exceeds the pool; every level of an item whose bounds are not finite
    if (tid == 0) {
      unsigned mask = fixed_ok ? 0u : (1u << L) - 1u;
#pragma unroll
      for (int l = 0; l < L; ++l) mask |= phase_of[l] < 0 ? 1u << l : 0u;
      flags[item] = (unsigned char)mask;
      if (mask) __hip_atomic_store(anyw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    item = item_n;
    have = have_n;
    b = b_n;
    m = m_n;
    qv = qv_n;
    stamp(10);
  }
  if constexpr (PCT_BCOL_STAMP && !DIRECT) {
    if (tid == 0 && stamps)
      for (int i = 0; i < 16; ++i) stamps[(size_t)blockIdx.x * 16 + i] = t_sum[i];
  }
}

// ---- kernel choice for the backward (diagnostics; see include/pctrans_hip.h) ---------------------------------------------
static std::atomic<int> g_bwd_choice{-1};                                    // -1: follow PCT_MSDA_BWD_KERNEL
static std::atomic<int> g_bwd_last{0};
void set_msda_bwd_kernel_choice(int c) { g_bwd_choice.store((c >= 0 && c <= 3) ? c : -1); }
int msda_bwd_last_kernel() { return g_bwd_last.load(); }
void note_msda_bwd_kernel(int k) { g_bwd_last.store(k); }
int msda_bwd_kernel_choice()
{
  const int c = g_bwd_choice.load();
  if (c >= 0) return c;
  static const int env = [] {
    const char *e = getenv("PCT_MSDA_BWD_KERNEL");
    return !e ? 0 : (!strcmp(e, "win") ? 1 : (!strcmp(e, "generic") ? 2 : (!strcmp(e, "col") ? 3 : 0)));
  }();
  return env;
}

// One byte per work item for the main launch to tell the DIRECT one which levels are left (see the header).  Buffers of
// BCOL_FLAG_BYTES each, allocated once per device: a ring for eager launches (launches in flight together on different
// streams must not share one) and a pool handed out once each to launches recorded into a HIP graph (the pointer is baked
// in).  nullptr: no buffer to be had (first use under stream capture, pool exhausted) -> the caller falls back.  A launch
// with more items than a buffer holds is dealt with inside the kernels (`overflow`).
// Eager ring: a slot is handed to launch k and again to launch k + RING.  Launches on ONE stream are ordered, so the re-use is
// safe there by construction; across streams it is only safe once the slot's previous launch has COMPLETED -- every slot
// carries the stream it was last used on and an event recorded behind that launch (bcol_flag_release), and a slot whose
// previous launch, on another stream, is still running is not handed out: the caller falls back to the windowed kernel
// (visible through pct_msda_last_bwd_kernel()).  `prepare` = allocate only (pct_prepare_device: outside any capture, once per
// device, so that the first backward never allocates or synchronises).
namespace {
constexpr int BCOL_MAX_DEV = 64, BCOL_RING = 8, BCOL_CAPTURE_POOL = 56;
struct BcolFlagPool {
  unsigned char *base = nullptr;
  unsigned seq = 0, cap_used = 0;
  hipStream_t last_stream[BCOL_RING] = {};
  hipEvent_t done[BCOL_RING] = {};
  bool used[BCOL_RING] = {};
};
std::mutex g_bcol_mu;
BcolFlagPool g_bcol_pool[BCOL_MAX_DEV];
}  // namespace

static unsigned char *bcol_flag_buffer(hipStream_t stream, int *slot_out, const bool prepare = false)
{
  *slot_out = -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BCOL_MAX_DEV) return nullptr;
  bool capturing = false;
  if (!prepare) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    capturing = cap != hipStreamCaptureStatusNone;
  }
  std::lock_guard<std::mutex> lock(g_bcol_mu);
  BcolFlagPool &pl = g_bcol_pool[dev];
  if (!pl.base) {
    if (capturing) return nullptr;                                              // (an allocation would invalidate the capture)
    void *p = nullptr;
    const size_t bytes = (size_t)(BCOL_RING + BCOL_CAPTURE_POOL) * BCOL_FLAG_STRIDE;
    // (the memset runs on the null stream; launches may come from non-blocking streams: wait for it once)
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError();
      if (p) (void)hipFree(p);
      return nullptr;
    }
    for (int i = 0; i < BCOL_RING; ++i)
      if (hipEventCreateWithFlags(&pl.done[i], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        for (int j = 0; j < i; ++j) (void)hipEventDestroy(pl.done[j]);
        (void)hipFree(p);
        return nullptr;
      }
    pl.base = static_cast<unsigned char *>(p);
  }
  if (prepare) return pl.base;
  if (capturing) {
    if (pl.cap_used >= (unsigned)BCOL_CAPTURE_POOL) return nullptr;
    return pl.base + (size_t)(BCOL_RING + pl.cap_used++) * BCOL_FLAG_STRIDE;
  }
  const int slot = (int)(pl.seq % BCOL_RING);
  if (pl.used[slot] && pl.last_stream[slot] != stream && hipEventQuery(pl.done[slot]) != hipSuccess) {
    (void)hipGetLastError();                                                    // still running on another stream: not shared
    return nullptr;
  }
  ++pl.seq;
  *slot_out = slot;
  return pl.base + (size_t)slot * BCOL_FLAG_STRIDE;
}
// behind the launches that use an eager slot: remember the stream and mark the slot's completion
static void bcol_flag_release(const int slot, hipStream_t stream)
{
  if (slot < 0) return;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BCOL_MAX_DEV) return;
  std::lock_guard<std::mutex> lock(g_bcol_mu);
  BcolFlagPool &pl = g_bcol_pool[dev];
  pl.last_stream[slot] = stream;
  pl.used[slot] = true;
  if (hipEventRecord(pl.done[slot], stream) != hipSuccess) (void)hipGetLastError();
}
int prepare_msda_backward_col_device()
{
  int slot;
  return bcol_flag_buffer(nullptr, &slot, true) ? 0 : (int)hipErrorOutOfMemory;
}

unsigned long long *win_stamp_buffer();                                      // msda_forward_win.hip (diagnostic)

// returns -100 when this geometry is not covered (caller uses another kernel)
int launch_msda_backward_col(const float *value, const int64_t *shapes, const int64_t *starts, const float *loc,
                             const float *attn, const float *grad_out, int N, int S, int M, int D, int L, int Lq,
                             int P, float *grad_value, float *grad_loc, float *grad_attn, bool forced, hipStream_t stream)
{
  if ((((uintptr_t)value | (uintptr_t)grad_out | (uintptr_t)loc | (uintptr_t)attn | (uintptr_t)grad_loc |
        (uintptr_t)grad_attn) & 15u) || ((uintptr_t)grad_value & 3u))
    return -100;
  if (D != 16 || P != 4 || L < 3 || L > 5 || Lq != S || M < 1) return -100;
  if ((long long)N * ((long long)S + 4096) * M >= 0x7fffffffLL) return -100;   // item / record arithmetic headroom
  if ((long long)S * M * L * P * 8 >= 0x7fffffffLL) return -100;               // 32-bit byte offsets inside an image, all tensors
  if ((long long)S * M * D * 4 >= 0x7fffffffLL) return -100;                   // (0x80000000 is the out-of-range sentinel)
  if (S >= (1 << 24) || M >= (1 << 16)) return -100;                           // 24-bit multiplies in the record index
  // (measured against the windowed kernel down to one 256^2-image pyramid, 43 k (query, head) pairs: 0.042 vs 0.055 ms;
  // profiles/r03_bwd_small_sizes.txt)
  if (!forced && (long long)N * S * M < 32768) return -100;
  int flag_slot = -1;
  unsigned char *flags = bcol_flag_buffer(stream, &flag_slot);
  if (!flags) return -100;
  // (PCT_BCOL_FLAG_CAP: a smaller capacity, so that a test can reach the overflow route with an oracle-sized case)
  static const int cap_env = [] { const char *e = getenv("PCT_BCOL_FLAG_CAP"); return e ? atoi(e) : 0; }();
  const int flag_cap = (cap_env > 0 && cap_env < (int)BCOL_FLAG_BYTES) ? cap_env : (int)BCOL_FLAG_BYTES;
  const size_t lds = (size_t)BCOL_POOL_BYTES + BCOL_TAB_BYTES + ((size_t)(BCOL_BLOCK / 64) * (5 * 2 + 2) + 4) * sizeof(unsigned);
  const size_t lds_direct = (size_t)BCOL_TAB_BYTES + 256;                      // (tables, boxes' slots, queue words)
  const dim3 grid(256 * 2), block(BCOL_BLOCK);
  unsigned *queue = win_queue_slot(stream);                                    // nullptr: static item stride
#define PCT_BCOL(L_)                                                                                                       \
  do {                                                                                                                     \
    const hipError_t attr_rc = func_attr_per_device(reinterpret_cast<const void *>(&msda_backward_col_kernel<L_, false>));  \
    const hipError_t attr_rc2 = func_attr_per_device(reinterpret_cast<const void *>(&msda_backward_col_kernel<L_, true>));  \
    if (attr_rc != hipSuccess) return (int)attr_rc;                                                                        \
    if (attr_rc2 != hipSuccess) return (int)attr_rc2;                                                                      \
    hipLaunchKernelGGL((msda_backward_col_kernel<L_, false>), grid, block, lds, stream, grad_out, value, shapes, starts,   \
                       loc, attn, N, S, M, grad_value, grad_loc, grad_attn, queue, flags, flag_cap, PCT_BCOL_STAMP ? win_stamp_buffer() : nullptr);                                 \
    hipLaunchKernelGGL((msda_backward_col_kernel<L_, true>), grid, block, lds_direct, stream, grad_out, value, shapes,     \
                       starts, loc, attn, N, S, M, grad_value, grad_loc, grad_attn, nullptr, flags, flag_cap, nullptr);                       \
  } while (0)
  if (L == 3) PCT_BCOL(3);
  else if (L == 4) PCT_BCOL(4);
  else PCT_BCOL(5);
#undef PCT_BCOL
  const int rc = (int)hipGetLastError();
  bcol_flag_release(flag_slot, stream);
  return rc;
}

}  // namespace pct
