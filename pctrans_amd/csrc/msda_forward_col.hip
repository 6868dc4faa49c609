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

#include "msda_col_common.hpp"

namespace pct {

#ifndef PCT_COL_STORE_NT
#define PCT_COL_STORE_NT 1
#endif
#ifndef PCT_COL_LOC_NT
#define PCT_COL_LOC_NT 0      /* experiment: nt on the location records only (whole lines, one reader) */
#endif
#ifndef PCT_COL_KO_NOCONF
#define PCT_COL_KO_NOCONF 0   /* knock-out (WRONG RESULTS, timing only): LDS gather addresses forced conflict-free */
#endif
#ifndef PCT_COL_STREAM_NT
#define PCT_COL_STREAM_NT 0   /* measured: nt on the record loads re-fetches the half lines two heads / two load groups share
                                (I: 7.9 -> 9.7 GB read per launch, 1.95 -> 2.14 ms); kept as a build switch */
#endif

// ---- quad-cooperative record access ------------------------------------------------------------------------------------
// A lane's records (128 B of sampling locations, 64 B of weights, 64 B of output per (query, head)) are 512 B - 1 KB apart
// from its neighbours': loaded lane by lane, every 16-byte access of a wave instruction touches a different 128-byte line
// and the texture-addresser serialises them (64 tag look-ups per instruction; measured: 16 K of the 77 K cycles an
// item took, and everything behind it in the queue waits).  So the 4 lanes of a quad fetch 64 CONSECUTIVE bytes of ONE
// record per instruction (16 look-ups), taking the quad's four records in turn, and the 4 x 4 block of 16-byte pieces is
// transposed in registers: instruction s gives lane i piece (s - i) % 4 of record s; the lane rotates its four registers
// by its own index (two layers of v_cndmask) and a quad_perm rotation by k delivers piece k of its own record.
template <int K>
__device__ __forceinline__ col_f32x4 quad_rot(const col_f32x4 v)      // lane c receives lane (c - K) % 4's value
{
  constexpr int ctrl = ((0 - K) & 3) | (((1 - K) & 3) << 2) | (((2 - K) & 3) << 4) | (((3 - K) & 3) << 6);
  if constexpr (K == 0) return v;
  return col_f32x4{dpp_f<ctrl>(v[0]), dpp_f<ctrl>(v[1]), dpp_f<ctrl>(v[2]), dpp_f<ctrl>(v[3])};
}
// x[k] <- x[(a + k) % 4] with the per-lane amount a = a0 + 2 * a1
__device__ __forceinline__ void rot_regs(col_f32x4 (&x)[4], const bool a0, const bool a1)
{
  col_f32x4 t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) t[k][e] = a0 ? x[(k + 1) & 3][e] : x[k][e];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) x[k][e] = a1 ? t[(k + 2) & 3][e] : t[k][e];
}
// in: x[s] = what this lane loaded for record s of its quad (piece (s - i) % 4); out: x[k] = piece k of its own record
__device__ __forceinline__ void quad_transpose_in(col_f32x4 (&x)[4], const bool i0, const bool i1)
{
  rot_regs(x, i0, i1);
  x[1] = quad_rot<1>(x[1]);
  x[2] = quad_rot<2>(x[2]);
  x[3] = quad_rot<3>(x[3]);
}

template <int L, bool FUSED, int BLOCK, bool STAMP = false>
__global__ __launch_bounds__(BLOCK, BLOCK == 1024 ? 4 : (BLOCK == 512 ? 4 : 3)) void msda_forward_col_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const float *__restrict__ loc, const float *__restrict__ attn, const int N, const int S, const int M,
    const int pool_px, float *__restrict__ out, const float *__restrict__ ref, const long long ref_batch_stride,
    unsigned *__restrict__ queue, unsigned long long *__restrict__ stamps = nullptr)
{
  constexpr int P = 4, D = 16, PXB = 64, NW = BLOCK / 64;
  // locations, weights and outputs are touched exactly once: non-temporal, so that they do not push the value lines (which
  // neighbouring columns and the sibling head re-use) out of the XCD's L2
  constexpr bool STREAM_NT = PCT_COL_STREAM_NT;
  // diagnostic build only (STAMP): per-phase cycle sums of wave 0, written to a buffer nothing else reads
  unsigned long long t_prev = 0, t_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int phase) {
    if constexpr (STAMP) {
      __builtin_amdgcn_sched_barrier(0);
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if (phase >= 0) t_sum[phase] += t - t_prev;
      t_prev = t;
    }
  };
  static_assert(L >= 1 && L <= 5 && NW <= 16, "unsupported geometry");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char *pool = smem_raw;                                              // level windows, 64 B per head-pixel
  unsigned *bb = reinterpret_cast<unsigned *>(smem_raw + (size_t)pool_px * PXB);   // [L][NW][2] per-wave boxes {min lo, ~max hi}
  unsigned *next_idx = bb + NW * L * 2;                                        // the workgroup's next item

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
    invWH[l] = uni_pair(1.0f / (float)Ws[l], 1.0f / (float)Hs[l]);           // FUSED: offset / (W, H) as offset * (1 / W, 1 / H)
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
  const UDiv dv_ncolM = make_udiv(ncol * M), dv_2ncol = make_udiv(2 * ncol), dv_CX = make_udiv(CX);
  const UDiv dv_2CX = make_udiv(2 * CX), dv_2CY = make_udiv(2 * CY);
  auto col_lo_f = [&](const int c, const int W, const int C, const UDiv dv_2C) {   // == col_lo(c, W, C), scalar unit
    return udiv_s(2 * c * W + C - 1, dv_2C);
  };

  // pixels 0 and 1 of the pool are zeros: gated-out samples read them (weight 0 times a guaranteed-finite value)
  if (tid < 8) reinterpret_cast<col_f32x4 *>(pool)[tid] = col_f32x4{0.f, 0.f, 0.f, 0.f};

  // this lane's place in its quad (quad-cooperative record access, above)
  int qi = lane & 3;
  const bool qi0 = qi & 1, qi1 = qi & 2;
  const bool qn1 = ((4 - qi) & 3) & 2;                                         // bit 1 of (-qi) % 4 (bit 0 is qi0)

  // per-lane rotation of the four 16-byte pieces of a head-pixel (see the header)
  const unsigned rho = (unsigned)(lane >> 3) & 3u;
  unsigned rot[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rot[j] = ((j + rho) & 3u) << 4;

  // ---- persistent, XCD-chunked walk over the items, software-pipelined across items --------------------------------
  // Queue protocol as in msda_win_common.hpp (first item static, every further one nslots + the XCD's counter, the fetch
  // that returns n_x - 1 resets the counter), run TWO items ahead: the index of item i + 1 is published before barrier
  // (A) of item i, so right behind that barrier every lane knows its next query and ISSUES THE NEXT ITEM'S SAMPLING-
  // LOCATION LOADS; they stay in flight through staging and gather of item i (32 registers) and item i + 1 starts on
  // data that has already landed.  (Measured before this change: a workgroup spent a third of an item waiting for these
  // loads.)  A fetch is made only while the next item is valid, so fetches == items processed still holds.
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int chunk = (items + 7) / 8;
  const int item_end = min((xcd + 1) * chunk, items);
  const unsigned n_x = (unsigned)max(item_end - xcd * chunk, 0);               // items of this XCD

  constexpr int NPL = L * 2, NGL = (NPL + 3) / 4;                              // locations: 16-byte pieces, 64-byte groups
  constexpr int NGW = (L + 3) / 4;                                             // weights: one 16-byte piece per level

  // item -> (image, head) [uniform] and this lane's query (qv = q, or ~q of the query an idle lane shadows)
  auto decode = [&](const int it, int &b_, int &m_, int &qv_) {
    // Item order inside an image: head PAIR outermost, then column, then the head inside the pair.  The two heads of a
    // pair share every 128-byte line of the value tensor (64 B each) and run in adjacent workgroups; the columns of one
    // pair follow each other, so an XCD works through one head pair's value maps (2.8 MB of lines at the north-star
    // shape, inside its 4 MB L2) with all the neighbouring columns -- whose windows overlap -- in flight together.
    // (With the 8 heads of a column adjacent instead, an XCD held 12 columns x 8 heads of windows at once, more than its
    // L2: measured 1.41x / 1.92x the algorithmic bytes on the memory side for distributions I / M.)
    b_ = udiv_s(it, dv_ncolM);
    const int r_img = it - b_ * (ncol * M);
    int col;
    if (r_img < 2 * ncol * (M >> 1)) {
      const int pr = udiv_s(r_img, dv_2ncol);
      const int rr = r_img - pr * 2 * ncol;
      col = rr >> 1;
      m_ = 2 * pr + (rr & 1);
    } else {                                                                   // odd head count: the last head alone
      col = r_img - 2 * ncol * (M >> 1);
      m_ = M - 1;
    }
    const int cy = udiv_s(col, dv_CX), cx = col - cy * CX;
    int q = 0, r = tid, q_first = 0;
    bool found = false, have_first = false;
#pragma unroll
    for (int ll = 0; ll < L; ++ll) {
      const int l = L - 1 - ll;                                               // finest level (last in PCTrans) first
      const int xa = col_lo_f(cx, Ws[l], CX, dv_2CX), nx = col_lo_f(cx + 1, Ws[l], CX, dv_2CX) - xa;
      const int ya = col_lo_f(cy, Hs[l], CY, dv_2CY), ny = col_lo_f(cy + 1, Hs[l], CY, dv_2CY) - ya;
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
    qv_ = found ? q : ~q_first;            // idle lanes shadow a query of the column: they cannot move its boxes
  };
  // the four queries of this lane's quad (record s of the quad belongs to lane 4 * (lane / 4) + s)
  auto quad_queries = [&](const int qv_, int (&qs)[4]) {
    qs[0] = dpp_i<0x00>(qv_);
    qs[1] = dpp_i<0x55>(qv_);
    qs[2] = dpp_i<0xAA>(qv_);
    qs[3] = dpp_i<0xFF>(qv_);
  };
  // group g (64 bytes) of the location records of the quad's four queries.  Per-image base pointers are uniform and the
  // offsets 32-bit (launcher: a per-image tensor stays below 4 GiB), so the loads take the scalar-base + vector-offset
  // form instead of 64-bit vector address arithmetic.
  auto issue_loc_group = [&](auto gc, const int b_, const int m_, const int qv_, col_f32x4 (&raw)[NGL][4]) {
    constexpr int g = decltype(gc)::value;
    int qs[4];
    quad_queries(qv_, qs);
    const float *base = loc + (long long)b_ * S * M * (L * P * 2);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      int pi = g * 4 + ((s4 - qi) & 3);
      if (NPL % 4 != 0 && pi >= NPL) pi = g * 4;                               // (a partial last group: harmless repeat)
      const unsigned r = (unsigned)((qs[s4] < 0 ? ~qs[s4] : qs[s4]) * M + m_);
      raw[g][s4] = (STREAM_NT || PCT_COL_LOC_NT) ? __builtin_nontemporal_load(reinterpret_cast<const col_f32x4 *>(
                                   base + (size_t)(r * (unsigned)(L * P * 2) + (unsigned)(pi * 4))))
                             : *reinterpret_cast<const col_f32x4 *>(base + (size_t)(r * (unsigned)(L * P * 2) + (unsigned)(pi * 4)));
    }
  };
  auto issue_loc = [&](const int b_, const int m_, const int qv_, col_f32x4 (&raw)[NGL][4]) {
    [&]<int... Gs>(std::integer_sequence<int, Gs...>) {
      (issue_loc_group(std::integral_constant<int, Gs>{}, b_, m_, qv_, raw), ...);
    }(std::make_integer_sequence<int, NGL>{});
  };
  // FUSED: the query's reference points (one (x, y) per level), fetched with the record
  auto issue_ref = [&](const int b_, const int qv_, col_f32x2 (&rr)[L]) {
    if constexpr (FUSED) {
      const float *rrow = ref + b_ * ref_batch_stride;
      const unsigned o = (unsigned)(qv_ < 0 ? ~qv_ : qv_) * (unsigned)(L * 2);
#pragma unroll
      for (int l = 0; l < L; ++l) rr[l] = *reinterpret_cast<const col_f32x2 *>(rrow + (size_t)(o + 2u * l));
    }
  };

  int item = xcd * chunk + slot0;
  bool have = item < item_end;
  int b = 0, m = 0, qv = 0;
  col_f32x4 raw[NGL][4];
  // (thread 0) the counter value fetched last lives in LDS word next_idx[1], not in a register: a register alive across
  // the whole item was spilled by the compiler, and spilling an atomic's result means waiting for it on the spot
  if (have) {
    decode(item, b, m, qv);
    issue_loc(b, m, qv, raw);
    if (queue && tid == 0) next_idx[1] = atomicAdd(queue + xcd, 1u);
  }

  stamp(-1);
  while (have) {
    // (opaque per iteration: the compiler otherwise hoists every per-lane expression of tid / qi out of the item loop --
    // a dozen 64-bit piece offsets, float copies of tid, ... -- and spills them)
    asm volatile("" : "+v"(tid), "+v"(qi));
    const long long rec_img = (long long)b * S;
    int qs[4];
    quad_queries(qv, qs);

    // FUSED: the reference points sit in L2 (shared by the heads and the batch): fetched here, not a whole item ahead
    // (eight more registers alive across the gather spilled); the transposition below runs while they arrive
    col_f32x2 rr[L];
    issue_ref(b, qv, rr);
    // ---- the record: sampling locations (FUSED: reference point + offset / (W, H)), loaded one item ago ----------------
    col_f32x2 lxy[L][P];
    {
#pragma unroll
      for (int g = 0; g < NGL; ++g) quad_transpose_in(raw[g], qi0, qi1);
#pragma unroll
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < P; ++k) {
          const col_f32x4 pc = raw[(l * 2 + k / 2) / 4][(l * 2 + k / 2) & 3];
          col_f32x2 v = {pc[(k & 1) * 2], pc[(k & 1) * 2 + 1]};
          if constexpr (FUSED) v = __builtin_elementwise_fma(v, invWH[l], rr[l]);
          // from here on the PIXEL coordinates (w_im, h_im) = loc * (W, H) - 0.5 (cuh:283-288), formed once for the
          // boxes and the gather (packed FMAs: x and y in one instruction)
          lxy[l][k] = __builtin_elementwise_fma(v, fWH[l], col_f32x2{-0.5f, -0.5f});
        }
    }

    // (everything that waits for the reference-point loads must be complete before the atomic below is issued)
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int k = 0; k < P; ++k) asm volatile("" : "+v"(lxy[l][k]));
    __builtin_amdgcn_sched_barrier(0);
    // ---- publish the next item's index, fetch the one after it (its value is parked in LDS behind the staging barrier).
    // Issued behind the loads the pre-pass waits for: the memory counter is in-order, a wait for those would include it. ---
    unsigned f_new = 0u;
    bool fetched = false;
    if (tid == 0) {
      unsigned nxt = (unsigned)(item - xcd * chunk + nslots);                 // static stride when there is no queue
      if (queue) {
        const unsigned f_next = next_idx[1];
        if (f_next + 1u >= n_x)                                               // that was the launch's last fetch
          __hip_atomic_store(queue + xcd, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nxt = (unsigned)nslots + f_next;
        if (nxt < n_x) {
          // (inline asm: the compiler waits for a returning atomic at the end of this divergent block -- a full round
          // trip to L2 in front of the pre-pass; the wait now sits where the value is parked, behind the pre-pass)
          const unsigned one = 1u, zero = 0u;
          const unsigned *qp = queue + xcd;
          // (s_nop: the base may just have been restored by v_readlane; a vector-memory instruction reading a scalar
          // register a vector instruction wrote needs 5 wait states, and the hazard recogniser does not see into inline asm)
          asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0"
                       : "=v"(f_new) : "v"(zero), "v"(one), "s"(qp) : "memory");
          fetched = true;
        }
      }
      next_idx[0] = nxt;
    }

    // ---- pre-pass: per-level bounding box (first corners, biased by +1; the box covers x0 .. x0 + 1) ----------------
#pragma unroll
    for (int l = 0; l < L; ++l) {
      // min / max over the lane's four samples on the raw pixel coordinates, then ONE clamp per lane and level (a clamp
      // is monotone, so it commutes with min / max; v_min / v_max return the other operand for a NaN, so a NaN sample --
      // gated out anyway -- drops out).  A sample is gated in iff -1 < w_im < W: clamping to [-1, W - 0.5] maps a
      // gated-out coordinate onto one a gated-in sample could have, so it can only widen the box towards the map
      // border, never past the 1-pixel apron.  The gather floors the very same registers.
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
    // ---- this item's weights (FUSED: logits): fetched now, looked at after the staging barrier.  (Behind the pre-pass:
    // anything that waits on the memory counter there -- it is in-order -- would otherwise wait for these loads too.) ---------
    col_f32x4 wraw[NGW][4];
    {
      const float *wbase_img = attn + rec_img * M * (L * P);
#pragma unroll
      for (int g = 0; g < NGW; ++g)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          int pi = g * 4 + ((s4 - qi) & 3);
          if (L % 4 != 0 && pi >= L) pi = g * 4;
          const unsigned r = (unsigned)((qs[s4] < 0 ? ~qs[s4] : qs[s4]) * M + m);
          wraw[g][s4] = STREAM_NT ? __builtin_nontemporal_load(reinterpret_cast<const col_f32x4 *>(
                                        wbase_img + (size_t)(r * (unsigned)(L * P) + (unsigned)(pi * 4))))
                                  : *reinterpret_cast<const col_f32x4 *>(wbase_img + (size_t)(r * (unsigned)(L * P) + (unsigned)(pi * 4)));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // boxes in LDS before the barrier
    stamp(0);
    __syncthreads();                                                          // (A) boxes visible; pool free
    stamp(1);

    // ---- the next item: its query per lane, its location loads issued (in flight until the next iteration) --------------
    int item_n, b_n = 0, m_n = 0, qv_n = 0;
    {
      const unsigned nxt = __builtin_amdgcn_readfirstlane(next_idx[0]);
      item_n = nxt < n_x ? xcd * chunk + (int)nxt : item_end;                // never out of the chunk
    }
    const bool have_n = item_n < item_end;             // (decoded once this item's first staging is issued, below)

    // ---- windows and phases (uniform) --------------------------------------------------------------------------------
    int wx0[L], wy0[L], wwid[L], whgt[L], wsize[L], wbase[L], phase_of[L];
    bool starts_phase[L];
    {
#pragma unroll
      for (int l = 0; l < L; ++l) {
        unsigned lo, hi;
        block_box<NW>(bb + l * NW * 2, lo, hi);
        const int x0 = (int)(lo & 0xFFFFu) - 1, y0 = (int)(lo >> 16) - 1;    // un-bias: origin may be -1 (apron)
        const int x1 = (int)(hi & 0xFFFFu) - 1, y1 = (int)(hi >> 16) - 1;
        const bool empty = x0 > x1 || y0 > y1;
        wx0[l] = x0;
        wy0[l] = y0;
        wwid[l] = empty ? 1 : x1 - x0 + 1;
        whgt[l] = empty ? 0 : y1 - y0 + 1;
        wsize[l] = wwid[l] * whgt[l];
      }
      // levels are gathered in a fixed order, finest (last) first; a level that does not fit beside the ones already
      // planned opens a new PHASE: the pool is re-staged right before it (two barriers)
      int ph = 0, used = 0;
      bool fresh = true;
#pragma unroll
      for (int ll = 0; ll < L; ++ll) {
        const int l = L - 1 - ll;
        starts_phase[l] = false;
        if (wsize[l] > pool_px - 2) {                                          // never fits: gathered from global memory
          phase_of[l] = -1;
          wbase[l] = 0;
          continue;
        }
        if (used + wsize[l] > pool_px - 2) {
          ++ph;
          used = 0;
          fresh = true;
        }
        phase_of[l] = ph;
        starts_phase[l] = fresh;
        fresh = false;
        wbase[l] = used + 2;                                                   // pixels 0, 1 are the zero pixels
        used += wsize[l];
      }
    }

    int woff[L];
#pragma unroll
    for (int l = 0; l < L; ++l) woff[l] = wbase[l] - wy0[l] * wwid[l] - wx0[l];
    float wts[L][P];
    stamp(2);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value + (long long)b * S * MD), 0,
                                                        (int)((unsigned)S * (unsigned)MD * 4u), 0x00020000);

    col_f32x2 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = col_f32x2{0.f, 0.f};

    // (the empty asm pins the accumulators: without it the compiler sinks a level's 128 packed FMAs below the LDS reads of
    // all four samples and spills the 256 data registers in between)
    auto pin_acc = [&]() {
      asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]),
                        "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
    };
    // two corners of one pixel row: corner-major, so consecutive packed FMAs go to different accumulators (8 chains)
    auto fma_row = [&](const col_f32x4 (&va)[4], const col_f32x4 (&vb)[4], const float wa, const float wb) {
      const col_f32x2 wwa = {wa, wa}, wwb = {wb, wb};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          acc[j][e] = __builtin_elementwise_fma(wwa, col_f32x2{va[j][2 * e], va[j][2 * e + 1]}, acc[j][e]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          acc[j][e] = __builtin_elementwise_fma(wwb, col_f32x2{vb[j][2 * e], vb[j][2 * e + 1]}, acc[j][e]);
    };
    // a sample's geometry: gate (cuh:290-296), first corner, the four bilinear * attention weights
    struct Geo {
      col_f32x2 g12, g34;                 // (hh * hw, hh * lw) * w and (lh * hw, lh * lw) * w
      int x0, y0;
      bool gate;
    };
    auto geometry = [&](auto lc, auto kc) {
      constexpr int l = decltype(lc)::value;
      constexpr int k = decltype(kc)::value;
      // (opaque: without it the compiler hoists every sample's geometry to the top of the item and keeps it alive)
      col_f32x2 pix = lxy[l][k];
      asm volatile("" : "+v"(pix));
      Geo g;
      g.gate = pix[1] > -1 && pix[0] > -1 && pix[1] < fH[l] && pix[0] < fW[l];    // false for NaN
      // a gated-out sample (possibly Inf / NaN coordinates) becomes pixel (0, 0) with weight 0, reading the zero pixels
      pix[0] = g.gate ? pix[0] : 0.f;
      pix[1] = g.gate ? pix[1] : 0.f;
      const float wgt = g.gate ? wts[l][k] : 0.f;
      g.x0 = cvt_flr(pix[0]);
      g.y0 = cvt_flr(pix[1]);
      const float lw = __builtin_amdgcn_fractf(pix[0]), lh = __builtin_amdgcn_fractf(pix[1]);
      const col_f32x2 ax = {1.f - lw, lw}, ay = {1.f - lh, lh};                // (hw, lw), (hh, lh): pairs as the packed ops take them
      const col_f32x2 t = ax * col_f32x2{wgt, wgt};                            // (hw, lw) * w
      g.g12 = t * col_f32x2{ay[0], ay[0]};
      g.g34 = t * col_f32x2{ay[1], ay[1]};
      return g;
    };

    // One level from its LDS window, one pixel ROW of a sample at a time (two corners = 8 x 16 B per lane in flight): the
    // other waves of the SIMD cover the LDS latency.  (A rolling pipeline with the next row's reads in flight during the
    // FMAs needs 32 more data registers than the 168 of three workgroups per CU leave: it spilled and was slower.)
    // All four corners lie inside the staged window (out-of-map ones are zeros); a gated-out sample reads the two zero
    // pixels (offset 0, row step 0).
    auto gather_level_lds = [&](auto lc) {
      constexpr int l = decltype(lc)::value;
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
        ([&] {
          const Geo g = geometry(lc, std::integral_constant<int, Ks>{});
          // pixel (x0, y0) sits at pool index woff + y0 * width + x0 (woff folds the window origin and base: one scalar)
#if PCT_COL_KO_NOCONF
          const unsigned a = g.gate ? (unsigned)(((__mul24(g.y0, wwid[l]) + g.x0 + woff[l]) & ~3) | (lane & 3)) << 6 : 0u;
#else
          const unsigned a = g.gate ? (unsigned)(__mul24(g.y0, wwid[l]) + g.x0 + woff[l]) << 6 : 0u;
#endif
          const unsigned rowb = g.gate ? (unsigned)wwid[l] << 6 : 0u;
          {
            col_f32x4 va[4], vb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned char *pa = pool + (a + rot[j]);
              va[j] = *reinterpret_cast<const col_f32x4 *>(pa);
              vb[j] = *reinterpret_cast<const col_f32x4 *>(pa + PXB);
            }
            fma_row(va, vb, g.g12[0], g.g12[1]);
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
            fma_row(va, vb, g.g34[0], g.g34[1]);
          }
          pin_acc();
          __builtin_amdgcn_sched_barrier(0);
        }(), ...);
      }(std::make_integer_sequence<int, P>{});
    };
    // ... or from global memory (a level whose box exceeds the pool), through the buffer descriptor: out-of-map corners
    // get an out-of-range offset and read zeros.  A head-pixel is 64 contiguous bytes; fetched lane by lane, every
    // 16-byte access of a wave instruction touches a different 128-byte line and the texture addresser serialises the 64
    // tag look-ups (measured: uniformly random locations ran at 0.064 of the roofline, half the windowed kernel's rate).
    // So the QUAD works on one member's sample at a time: the member's four corner offsets and weights are broadcast
    // (DPP), lane c fetches piece (s - c) % 4 of every corner of member s's sample -- the quad reads whole 64-byte
    // pixels, 16 look-ups per instruction -- and accumulates "its piece of member s's sum".  After the level the 4 x 4
    // block of pieces is transposed back (as the record loads are) and added to the lane's own accumulators.
    // Measured at the north-star shape, uniformly random locations: 13.9 -> 5.3 ms; distributions I / M pay < 1 % (same
    // box A/B).  (As a real, non-inlined function the kernel ran at half speed on EVERY input: scratch set-up and
    // call-clobbered registers.  With two or four members' fetches in flight the hot path picked up spills whose reloads
    // wait on the memory counter behind the prefetched records: +5 % on I / M, and 6.1 ms on U.  One member at a time is
    // spill-free at 4 levels and the fastest on U.)
    auto gather_level_global = [&](auto lc) {
      constexpr int l = decltype(lc)::value;
      const int H = Hs[l], W = Ws[l];
      constexpr unsigned OOB = 0x80000000u;                                    // S * M * D * 4 < 2^31 (C ABI check)
      const unsigned MDb = (unsigned)MD * 4u;
      col_f32x2 part[4][2];                                                    // [member s]: piece (s - qi) % 4 of s's sum
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) part[s4][0] = part[s4][1] = col_f32x2{0.f, 0.f};
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
        ([&] {
          const Geo g = geometry(lc, std::integral_constant<int, Ks>{});
          const bool top = g.gate && g.y0 >= 0, bot = g.gate && g.y0 + 1 <= H - 1;
          const bool lft = g.x0 >= 0, rgt = g.x0 + 1 <= W - 1;
          const unsigned a = (unsigned)(St[l] + g.y0 * W + g.x0) * MDb + (unsigned)(m * D) * 4u;
          const unsigned o1 = (top && lft) ? a : OOB, o2 = (top && rgt) ? a + MDb : OOB;
          const unsigned o3 = (bot && lft) ? a + (unsigned)W * MDb : OOB, o4 = (bot && rgt) ? a + (unsigned)W * MDb + MDb : OOB;
          [&]<int... Ss>(std::integer_sequence<int, Ss...>) {                    // one member at a time (16 registers of data)
            ([&] {
              constexpr int CT = BcastCtrl<4, Ss>::value;
              const unsigned pc = (unsigned)((Ss - qi) & 3) << 4;              // (an out-of-range offset stays out of range)
              col_f32x4 v[4];
              v[0] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(dpp_u<CT>(o1) + pc), 0, 0));
              v[1] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(dpp_u<CT>(o2) + pc), 0, 0));
              v[2] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(dpp_u<CT>(o3) + pc), 0, 0));
              v[3] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(dpp_u<CT>(o4) + pc), 0, 0));
              const float w[4] = {dpp_f<CT>(g.g12[0]), dpp_f<CT>(g.g12[1]), dpp_f<CT>(g.g34[0]), dpp_f<CT>(g.g34[1])};
#pragma unroll
              for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                  part[Ss][e] = __builtin_elementwise_fma(col_f32x2{w[c4], w[c4]},
                                                          col_f32x2{v[c4][2 * e], v[c4][2 * e + 1]}, part[Ss][e]);
              asm volatile("" : "+v"(part[0][0]), "+v"(part[0][1]), "+v"(part[1][0]), "+v"(part[1][1]), "+v"(part[2][0]),
                                "+v"(part[2][1]), "+v"(part[3][0]), "+v"(part[3][1]));
              __builtin_amdgcn_sched_barrier(0);
            }(), ...);
          }(std::make_integer_sequence<int, 4>{});
        }(), ...);
      }(std::make_integer_sequence<int, P>{});
      col_f32x4 x[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) x[s4] = col_f32x4{part[s4][0][0], part[s4][0][1], part[s4][1][0], part[s4][1][1]};
      quad_transpose_in(x, qi0, qi1);                                          // x[k] = piece k of this lane's own sum
      rot_regs(x, rho & 1u, rho & 2u);                                         // x[j] = piece (j + rho) % 4: the slot order
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        acc[j4][0] += col_f32x2{x[j4][0], x[j4][1]};
        acc[j4][1] += col_f32x2{x[j4][2], x[j4][3]};
      }
      pin_acc();
      __builtin_amdgcn_sched_barrier(0);
    };

    // stage the windows of one phase: LDS-DMA (buffer_load_dwordx4 ... lds), no VGPR round trip, every piece in flight at
    // once; LDS address of a piece = wave-uniform base + lane * 16; the global source is a per-lane 32-bit offset into this
    // image's buffer descriptor
    auto stage_phase = [&](const int phx) {
      // One wave instruction copies 64 consecutive 16-byte pieces = CPX pixels of ONE window row; the waves take the rows
      // in turn.  A lane's source offset splits into a per-lane part that depends on its column only (computed once per
      // level and 64-piece column block) and a per-row part that is uniform and travels in the instruction's scalar
      // offset: no vector arithmetic per copy.  (Piece-linear indexing cost ~18 vector instructions per copy: a third of
      // the kernel's vector work when the windows are wide.)  Columns outside the map keep an out-of-range offset, rows
      // outside the map use it for every lane: the descriptor's bounds check then delivers zeros -- the apron.
      constexpr int PPX = PXB / 16, CPX = 64 / PPX;
      constexpr unsigned OOB = 0x80000000u;
      const int ln = tid & 63;
      const int dx = ln / PPX, cc = ln & (PPX - 1);
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
      const unsigned MDb = (unsigned)MD * 4u;                                 // bytes per pixel, all heads
#pragma unroll
      for (int l = 0; l < L; ++l) {
        if (phase_of[l] == phx && wsize[l] > 0) {
          const unsigned lvl_off = (unsigned)St[l] * MDb + (unsigned)(m * D) * 4u;   // bytes, this level and head
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
    auto front_end = [&]() {                                                  // the weights, once they are needed
#pragma unroll
      for (int g = 0; g < NGW; ++g) quad_transpose_in(wraw[g], qi0, qi1);
#pragma unroll
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < P; ++k) wts[l][k] = wraw[l / 4][l & 3][k];
      if constexpr (FUSED) {                                                  // softmax over the record's L * P logits
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
    };

    auto level_step = [&](auto llc) {
      constexpr int ll = decltype(llc)::value;
      constexpr int l = L - 1 - ll;
      if (starts_phase[l]) {
        if (ll > 0) __syncthreads();                                          // every wave is done with the pool
        stage_phase(phase_of[l]);
        if constexpr (ll == 0) {
          if (have_n) decode(item_n, b_n, m_n, qv_n);                          // while the LDS-DMA pieces are in flight
        }
        if (ll == 0) stamp(3);
        __syncthreads();                                                      // windows staged (vmcnt(0) + barrier)
        if (ll == 0) stamp(4);
      }
      if constexpr (ll == 0) {
        // (thread 0) park the counter value fetched at the top of the item: the weights below need the memory counter at
        // zero anyway, so this wait is free -- anywhere earlier it would stall on the loads issued since
        if (fetched) {
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(f_new)::"memory");
          next_idx[1] = f_new;
        }
        front_end();
      }
      if (phase_of[l] >= 0) gather_level_lds(std::integral_constant<int, l>{});
      else gather_level_global(std::integral_constant<int, l>{});             // box larger than the pool: global memory
      // one group of the next item's location record per level: its registers are the ones this level's points freed
      if constexpr (ll < NGL) {
        if (have_n) issue_loc_group(std::integral_constant<int, ll>{}, b_n, m_n, qv_n, raw);
      }

    };
    if (!starts_phase[L - 1]) {                                               // (the finest level is gathered from global)
      if (have_n) decode(item_n, b_n, m_n, qv_n);
      stamp(3);
      stamp(4);
    }
    [&]<int... LLs>(std::integer_sequence<int, LLs...>) {
      (level_step(std::integral_constant<int, LLs>{}), ...);
    }(std::make_integer_sequence<int, L>{});
    static_assert(NGL <= L, "one location group per level");

    stamp(5);
    {
      // store, quad-cooperatively: register j of lane c holds piece (j + rho) % 4 of ITS record (rho is the same in the
      // whole quad).  A quad_perm rotation by k hands lane i the register k of lane (i - k) % 4; rotating the four received
      // registers by -i puts the one that came from lane s into slot s: piece (i - s + rho) % 4 of record s.  Store
      // instruction s then writes the 64 consecutive bytes of record s from the quad's four lanes.
      col_f32x4 u[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = col_f32x4{acc[j][0][0], acc[j][0][1], acc[j][1][0], acc[j][1][1]};
      u[1] = quad_rot<1>(u[1]);
      u[2] = quad_rot<2>(u[2]);
      u[3] = quad_rot<3>(u[3]);
      // slot s <- received register (i - s) % 4: reverse the order (compile time), then rotate by (-i) % 4
      col_f32x4 w4[4] = {u[0], u[3], u[2], u[1]};
      rot_regs(w4, qi0, qn1);
      int qo[4];
      quad_queries(qv, qo);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (qo[s4] >= 0) {
          float *op = out + rec_img * M * D +
                      (size_t)((unsigned)(qo[s4] * M + m) * (unsigned)D + (unsigned)((((qi - s4) + (int)rho) & 3) * 4));
          if (STREAM_NT || PCT_COL_STORE_NT) __builtin_nontemporal_store(w4[s4], reinterpret_cast<col_f32x4 *>(op));
          else *reinterpret_cast<col_f32x4 *>(op) = w4[s4];
        }
      }
    }

    item = item_n;
    have = have_n;
    b = b_n;
    m = m_n;
    qv = qv_n;
    stamp(6);
  }
  if constexpr (STAMP) {
    if (tid == 0 && stamps)
      for (int i = 0; i < 8; ++i) stamps[(size_t)blockIdx.x * 8 + i] = t_sum[i];
  }
}

unsigned long long *win_stamp_buffer();                                      // msda_forward_win.hip (diagnostic)

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
  if ((long long)S * M * L * P * 8 >= 0xffffffffLL) return -100;               // 32-bit byte offsets inside an image
  // threads per workgroup x workgroups per CU: 1024 x 1 (~150 KB pool, 128 registers), 512 x 2 (~75 KB each, 128 registers),
  // 384 x 2 (~75 KB, 168 registers), 256 x 3 (~50 KB, 168 registers).  With more than one workgroup per CU the memory
  // phases of one (records, staging) overlap the gather of the others.
  static const int block_env = [] { const char *e = getenv("PCT_COL_BLOCK"); const int v = e ? atoi(e) : 0;
                                    return (v == 256 || v == 384 || v == 512 || v == 768 || v == 1024) ? v : 256; }();
  static const int pool_env = [] { const char *e = getenv("PCT_COL_POOL_KB"); return e ? atoi(e) : 0; }();
  const int BLOCKV = block_env;
  const bool one_wg = BLOCKV == 1024 || BLOCKV == 768;
  const int pool_max = one_wg ? 158 : 78;
  const int pool_kb = (pool_env >= 16 && pool_env <= pool_max) ? pool_env : (one_wg ? 150 : (BLOCKV == 256 ? 50 : 74));
  const int wg_per_cu = one_wg ? 1 : ((BLOCKV == 256 && pool_kb <= 52) ? 3 : 2);
  const int pool_px = pool_kb * 1024 / 64;
  const size_t lds = (size_t)pool_px * 64 + ((size_t)(BLOCKV / 64) * L * 2 + 4) * sizeof(unsigned);   // pool, boxes, queue words
  const dim3 grid(256 * wg_per_cu), block(BLOCKV);
  unsigned *queue = win_queue_slot(stream);                                    // nullptr: static item stride
  const float *v = static_cast<const float *>(value);
  const float *lc = static_cast<const float *>(loc), *at = static_cast<const float *>(attn);
  float *o = static_cast<float *>(out);
#define PCT_COL_K(L_, FU_, B_, ST_)                                                                                     \
  do {                                                                                                                  \
    static const hipError_t attr_rc = hipFuncSetAttribute(                                                              \
        reinterpret_cast<const void *>(&msda_forward_col_kernel<L_, FU_, B_, ST_>),                                     \
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                                        \
    if (attr_rc != hipSuccess) return (int)attr_rc;                                                                     \
    hipLaunchKernelGGL((msda_forward_col_kernel<L_, FU_, B_, ST_>), grid, block, lds, stream, v, shapes, starts, lc,    \
                       at, N, S, M, pool_px, o, ref, ref_batch_stride, queue, ST_ ? win_stamp_buffer() : nullptr);      \
  } while (0)
#define PCT_COL_B(L_, FU_, ST_)                                \
  do {                                                         \
    if (BLOCKV == 1024) PCT_COL_K(L_, FU_, 1024, ST_);         \
    else if (BLOCKV == 768) PCT_COL_K(L_, FU_, 768, ST_);      \
    else if (BLOCKV == 512) PCT_COL_K(L_, FU_, 512, ST_);      \
    else if (BLOCKV == 384) PCT_COL_K(L_, FU_, 384, ST_);      \
    else PCT_COL_K(L_, FU_, 256, ST_);                         \
  } while (0)
#define PCT_COL(L_, FU_) PCT_COL_B(L_, FU_, false)
  if (win_stamp_buffer() && L == 4) {   // diagnostic: per-phase cycle stamps (tools/stamp_msda.py)
    if (ref) PCT_COL_B(4, true, true);
    else PCT_COL_B(4, false, true);
    return (int)hipGetLastError();
  }
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
#undef PCT_COL_B
#undef PCT_COL_K
  return (int)hipGetLastError();
}

}  // namespace pct
