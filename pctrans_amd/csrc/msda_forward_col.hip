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
#ifndef PCT_COL_W_NT
#define PCT_COL_W_NT 0
#endif
#ifndef PCT_COL_KO_NOLOC
#define PCT_COL_KO_NOLOC 0    /* knock-out (WRONG RESULTS, timing only): 1 = the location records are not loaded, 2 = nor the weights */
#endif
#ifndef PCT_COL_KO_NOSTORE
#define PCT_COL_KO_NOSTORE 0  /* knock-out (WRONG RESULTS, timing only): no output stores */
#endif
#ifndef PCT_COL_W_LATE
#define PCT_COL_W_LATE 0      /* experiment: the odd head of a pair fetches its weights later (no effect on traffic or time) */
#endif
#ifndef PCT_COL_ITEM_ORDER
#define PCT_COL_ITEM_ORDER 0
#endif
#ifndef PCT_COL_EARLY
#define PCT_COL_EARLY 1       /* the second phase's windows are staged into the finest level's region as soon as that level is gathered */
#endif
#ifndef PCT_COL_EARLY_WAIT
#define PCT_COL_EARLY_WAIT 1  /* the early-staged second phase is waited for with a counted vmcnt (the next item's record stays in flight) */
#endif
#ifndef PCT_COL_PRIO
#define PCT_COL_PRIO 8        /* wave priorities (s_setprio): 0 = none, 8 = by part of an item as set below; 1..7: A/B variants.  A wave
                                 in its gather is served before the waves of the other workgroups that are decoding, planning or
                                 staging -- the gather is the LDS-latency chain of an item, everything else fills its gaps --
                                 and the first (finest, largest) level before the later ones.  Same-box A/B, P2 batch 128:
                                 I 1.70 -> 1.65 ms, M 2.24 -> 2.16 ms; other splits (gather 3 / 3, 1 / 1, planning raised) were
                                 within 1 % of each other on I and 2 % worse on M */
#endif
#ifndef PCT_COL_PRIO_TOP      /* wave priorities (s_setprio, 0..3) by part of an item, PCT_COL_PRIO == 8: records + pre-pass, */
#define PCT_COL_PRIO_TOP 0    /* planning + staging + decode, first gathered level, other levels                             */
#define PCT_COL_PRIO_PLAN 0
#define PCT_COL_PRIO_L0 3
#define PCT_COL_PRIO_REST 2
#endif
#ifndef PCT_COL_PRIO_FRONT
#define PCT_COL_PRIO_FRONT 0
#endif
#ifndef PCT_COL_KO_NOSTAGE
#define PCT_COL_KO_NOSTAGE 0  /* knock-out (WRONG RESULTS, timing only): no window staging (LDS-DMA) at all */
#endif
#ifndef PCT_COL_KO_NOGATHER
#define PCT_COL_KO_NOGATHER 0 /* knock-out (WRONG RESULTS, timing only): 1 = no LDS reads + FMAs in the gather, 2 = no gather at all */
#endif
#ifndef PCT_COL_LATIN
#define PCT_COL_LATIN 2       /* LDS gather with a per-lane corner order that is bank-conflict free for ANY locations (see
                                 gather_level_lds_latin): 0 = off, 1 = every level, 2 = levels whose window is much larger than the
                                 column's cell (wide offset distributions).  Same-box A/B, P2 batch 128 (profiles/
                                 r04_forward_structural_ab.txt): 2 against 0: M 2.168 -> 2.085 ms as an op, 2.290 -> 2.209 ms in the
                                 bench step; I 1.649 -> 1.700 (the two code paths cost the init-like case 3 spilled registers); 1: M 2.038,
                                 I 1.779 */
#endif
#ifndef PCT_COL_LATIN_HALO
#define PCT_COL_LATIN_HALO 16
#endif
#ifndef PCT_COL_KO_ALIAS
#define PCT_COL_KO_ALIAS 0    /* knock-out (WRONG RESULTS, timing only): every image reads and writes image 0's tensors -- the same instruction
                                 stream with (almost) no HBM traffic: what the kernel costs when memory bandwidth is free */
#endif
#ifndef PCT_COL_KO_NOCONF
#define PCT_COL_KO_NOCONF 0   /* knock-out (WRONG RESULTS, timing only): LDS gather addresses forced conflict-free */
#endif
#ifndef PCT_COL_ORDER
#define PCT_COL_ORDER 1       /* gather order of the levels: 0 = finest .. coarsest, 1 = finest, coarsest, then the rest */
#endif
#ifndef PCT_COL_LOC_AT
#define PCT_COL_LOC_AT 0      /* the gather step behind which the next item's location record is fetched, all groups at once (the
                                 two 64-byte halves of a 128-byte line back to back: -2 % on I and M against one group
                                 per gathered level, same-box A/B) */
#endif
#ifndef PCT_COL_STREAM_NT
#define PCT_COL_STREAM_NT 0   /* measured: nt on the record loads re-fetches the half lines two heads / two load groups share
                                (I: 7.9 -> 9.7 GB read per launch, 1.95 -> 2.14 ms); kept as a build switch */
#endif

// The knock-out switches produce WRONG RESULTS by design (timing experiments, tools/variant.sh): a library built with one
// of them must say so -- refused unless the build declares itself an experiment, and reported by pct_build_info().
#if (PCT_COL_KO_NOLOC || PCT_COL_KO_NOSTORE || PCT_COL_KO_NOSTAGE || PCT_COL_KO_NOGATHER || PCT_COL_KO_NOCONF || PCT_COL_KO_ALIAS) && \
    !defined(PCT_EXPERIMENT_BUILD)
#error "PCT_COL_KO_* knock-outs give wrong results: add -DPCT_EXPERIMENT_BUILD (tools/variant.sh does) to build one"
#endif
#define PCT_STR2(x) #x
#define PCT_STR(x) PCT_STR2(x)
const char *msda_forward_col_build_flags()
{
  return "col: KO=" PCT_STR(PCT_COL_KO_NOLOC) PCT_STR(PCT_COL_KO_NOSTORE) PCT_STR(PCT_COL_KO_NOSTAGE) PCT_STR(PCT_COL_KO_NOGATHER)
         PCT_STR(PCT_COL_KO_NOCONF) PCT_STR(PCT_COL_KO_ALIAS) " STORE_NT=" PCT_STR(PCT_COL_STORE_NT) " LOC_NT=" PCT_STR(PCT_COL_LOC_NT) " W_NT=" PCT_STR(PCT_COL_W_NT)
         " STREAM_NT=" PCT_STR(PCT_COL_STREAM_NT) " W_LATE=" PCT_STR(PCT_COL_W_LATE) " ITEM_ORDER=" PCT_STR(PCT_COL_ITEM_ORDER)
         " EARLY=" PCT_STR(PCT_COL_EARLY) " EARLY_WAIT=" PCT_STR(PCT_COL_EARLY_WAIT) " PRIO=" PCT_STR(PCT_COL_PRIO) "/" PCT_STR(PCT_COL_PRIO_TOP)
         PCT_STR(PCT_COL_PRIO_PLAN) PCT_STR(PCT_COL_PRIO_L0) PCT_STR(PCT_COL_PRIO_REST) PCT_STR(PCT_COL_PRIO_FRONT) " ORDER=" PCT_STR(PCT_COL_ORDER)
         " LOC_AT=" PCT_STR(PCT_COL_LOC_AT) " LATIN=" PCT_STR(PCT_COL_LATIN) "/" PCT_STR(PCT_COL_LATIN_HALO);
}

// Gather order of the levels (step ll -> level).  The finest level (the last one in PCTrans' pyramids) comes first; its
// window is the largest, and the level that most often still fits beside it in the pool is the COARSEST one, so that
// one goes second and the two middle levels share the next phase: with model-like offsets at the north-star shape
// {finest, coarsest} + {the two middle levels} is two phases where finest-to-coarsest order needed three for every
// fifth item (each phase costs a re-staging of the pool and two barriers).
template <int L>
__host__ __device__ constexpr int col_level_of_step(const int ll)
{
  return PCT_COL_ORDER == 0 ? L - 1 - ll : (ll == 0 ? L - 1 : (ll == 1 ? 0 : L - ll));
}

// PP ("piece planes", an MI355X-native operand layout for the callers that control the producers -- the encoder's own
// projection GEMM writes it, pct_ms_deform_attn_forward_planes_f32): value, locations / offsets and weights / logits are
// stored per image as [heads * pieces][queries][4 floats], piece = four consecutive floats of a (query, head) record.  A
// lane then loads piece i of ITS OWN record at (plane * S + query) * 16: the 64 lanes of a wave read the pieces of
// consecutive queries, i.e. whole 128-byte lines, 2 - 8 tag look-ups per instruction -- and the quad-cooperative access with
// its register transposes (176 vector instructions per wave and item), the half lines a record shares with its sibling
// head's and the 64-byte-in-512 pixel slices of the window staging are all gone.  Same arithmetic, bit-identical output.
template <int L, bool FUSED, int BLOCK, bool STAMP = false, bool PP = false>
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
  unsigned long long t_prev = 0, t_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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
  col_f32x2 fWH[L], invWH[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
    St[l] = (int)starts[l];
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
        mx = col_max_cell(Ws[l], CX);
        my = col_max_cell(Hs[l], CY);
        maxq += mx * my;
      }
      if (maxq <= BLOCK) break;
      if (CY < Hf) ++CY;
      else if (CX < Wf) ++CX;
      else break;                                                              // (L pixels per column: cannot exceed BLOCK)
    }
  }

  // ---- this lane's SLOT in a column (fixed for the whole launch).  A column's queries are dealt to the lanes level by
  // level, finest (last) first, each level taking a block of mx * my lanes where mx x my is its LARGEST cell over the
  // grid -- the grid search above made these blocks fit the workgroup -- so a lane's (level, x, y inside the cell) never
  // changes from item to item and the per-item decode is two table look-ups and a multiply-add (it used to walk the
  // levels: ~100 vector and ~250 scalar instructions per item).  A lane whose slot lies outside a smaller border cell
  // idles for that item.
  int lane_c0 = 0;                                                             // level start + (y * W + x) inside the cell
  unsigned lane_slot = 0x000FFFFFu;                                            // x | y << 10 | level << 20; x = y = 1023: no slot
  {
    int r = tid;
    bool placed = false;
#pragma unroll
    for (int ll = 0; ll < L; ++ll) {
      const int l = L - 1 - ll;
      int mx = 0, my = 0;
      mx = col_max_cell(Ws[l], CX);
      my = col_max_cell(Hs[l], CY);
      const int cnt = mx * my;                                                 // (<= BLOCK <= 1024: x, y fit 10 bits)
      if (!placed && r < cnt) {
        const int ly = r / max(mx, 1), lx = r - ly * max(mx, 1);
        lane_slot = (unsigned)lx | ((unsigned)ly << 10) | ((unsigned)l << 20);
        lane_c0 = St[l] + ly * Ws[l] + lx;
        placed = true;
      }
      r -= placed ? 0 : cnt;
    }
  }
  // cell tables in LDS, carved off the end of the pool: xtab[l][cx] = first x | width << 16, ytab[l][cy] likewise, then
  // the level widths.  A grid whose tables would take more than half the pool (absurdly elongated maps: thousands of
  // columns along one axis) is not used at all: the launch then runs on FLAT columns, see the decode.
  const int tab_px = ((L * (CX + CY) + L) * 4 + PXB - 1) / PXB;
  const int pool_eff = pool_px - tab_px;
  unsigned *tab = reinterpret_cast<unsigned *>(pool + (size_t)max(pool_eff, 0) * PXB);
  const bool use_tab = pool_eff * 2 >= pool_px;
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
    for (int l = 0; l < L; ++l) ncol += (Hs[l] * Ws[l] + BLOCK - 1) / BLOCK;
  }
  const int items = N * ncol * M;
  const UDiv dv_ncolM = make_udiv(ncol * M), dv_2ncol = make_udiv(2 * ncol), dv_CX = make_udiv(CX);

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
    if (PCT_COL_KO_ALIAS) b_ = 0;
    int col;
    if (PCT_COL_ITEM_ORDER == 1 && M == 8) {                                   // (experiment) column outermost, the 8 heads adjacent
      col = r_img >> 3;
      m_ = r_img & 7;
    } else if (PCT_COL_ITEM_ORDER == 2 && M == 8) {                            // (experiment) two pairs, then column, then 4 heads
      const int pg = udiv_s(r_img, make_udiv(4 * ncol));
      const int rr = r_img - pg * 4 * ncol;
      col = rr >> 2;
      m_ = 4 * pg + (rr & 3);
    } else if (r_img < 2 * ncol * (M >> 1)) {
      const int pr = udiv_s(r_img, dv_2ncol);
      const int rr = r_img - pr * 2 * ncol;
      col = rr >> 1;
      m_ = 2 * pr + (rr & 1);
    } else {                                                                   // odd head count: the last head alone
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
      // flat columns: BLOCK consecutive queries of one level (no tables, no divisions; the boxes of such a strip are wide,
      // so its levels mostly take the global-memory path -- correct, not fast)
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
    // idle lanes shadow a query of the column (they cannot move its boxes): the wave's first busy lane's; a wave without
    // any busy lane shadows query 0 (correct, merely wide boxes; it takes a pyramid whose levels shrink the wrong way)
    const unsigned long long vm = __ballot(valid);
    const int q_sh = vm ? __builtin_amdgcn_readlane(q, (int)__builtin_ctzll(vm)) : 0;
    qv_ = valid ? q : ~q_sh;
  };
  // ---- quad-cooperative record access (see the top of the file) through per-image buffer descriptors: a scalar base, one
  // 32-bit vector offset per record and the piece as an immediate -- no 64-bit vector address arithmetic.  A lane forms
  // the byte offset of ITS OWN record once; the quad's other three get it by DPP (record s of the quad belongs to lane
  // 4 * (lane / 4) + s).  Offsets past the end of a descriptor read zeros / drop the store: an idle lane's record is
  // simply moved out of range for the store.  (launcher: S < 2^24, per-image tensors below 2 GiB)
  typedef unsigned col_u32x4 __attribute__((ext_vector_type(4)));
  auto rec_index = [&](const int qv_, const int m_) {                         // query * M + head of the lane's own record
    return __umul24((unsigned)(qv_ ^ (qv_ >> 31)), (unsigned)M) + (unsigned)m_;
  };
  auto quad_offsets = [&](const unsigned own_bytes, unsigned (&off)[4]) {     // + the piece this lane fetches of record s
    off[0] = dpp_u<0x00>(own_bytes) + ((unsigned)((0 - qi) & 3) << 4);
    off[1] = dpp_u<0x55>(own_bytes) + ((unsigned)((1 - qi) & 3) << 4);
    off[2] = dpp_u<0xAA>(own_bytes) + ((unsigned)((2 - qi) & 3) << 4);
    off[3] = dpp_u<0xFF>(own_bytes) + ((unsigned)((3 - qi) & 3) << 4);
  };
  constexpr int RSRC_FLAGS = 0x00020000;
  // the location records of the quad's four queries, all groups (64 bytes each) at once: the two halves of a 128-byte line
  // are requested back to back
  auto issue_loc = [&](const int b_, const int m_, const int qv_, col_f32x4 (&raw)[NGL][4]) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(loc + (long long)b_ * S * M * (L * P * 2)), 0,
                                                      (int)((unsigned)S * (unsigned)M * (unsigned)(L * P * 8)), RSRC_FLAGS);
    if constexpr (PP) {                           // piece i of (query, head m_): plane m_ * NPL + i, the plane in the scalar offset
      const unsigned qo = (unsigned)(qv_ ^ (qv_ >> 31)) << 4;
      const unsigned s16 = (unsigned)__builtin_amdgcn_readfirstlane(S) * 16u;
      const unsigned p0 = (unsigned)__builtin_amdgcn_readfirstlane(m_) * (unsigned)NPL * s16;       // (scalar unit, not 12 vector multiplies)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int g = 0; g < NGL; ++g)
          raw[g][s4] = g * 4 + s4 < NPL
                           ? __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                 rs, (int)qo, (int)(p0 + (unsigned)(g * 4 + s4) * s16), PCT_COL_LOC_NT ? 2 : 0))
                           : col_f32x4{0.f, 0.f, 0.f, 0.f};
      return;
    }
    unsigned off[4];
    quad_offsets(rec_index(qv_, m_) * (unsigned)(L * P * 8), off);
    if (PCT_COL_KO_NOLOC == 1 || PCT_COL_KO_NOLOC == 2) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int g = 0; g < NGL; ++g) raw[g][s4] = col_f32x4{(float)off[s4], 0.3f, 0.4f, 0.6f};
      return;
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int g = 0; g < NGL; ++g)                // (a partial last group reads into the next record: never looked at)
        raw[g][s4] = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off[s4] + g * 64), 0, PCT_COL_LOC_NT ? 2 : 0));
  };
  // FUSED: the query's reference points (one (x, y) per level), fetched with the record
  auto issue_ref = [&](const int b_, const int qv_, col_f32x2 (&rr)[L]) {
    if constexpr (FUSED) {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ref + b_ * ref_batch_stride), 0,
                                                        (int)((unsigned)S * (unsigned)(L * 8)), RSRC_FLAGS);
      const unsigned o = (unsigned)(qv_ ^ (qv_ >> 31)) * (unsigned)(L * 8);
      if constexpr (L % 2 == 0) {                   // 16-byte loads (launcher: ref is 16-byte aligned when L is even)
#pragma unroll
        for (int l = 0; l < L; l += 2) {
          const col_f32x4 r4 = __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(o + 8u * l), 0, 0));
          rr[l] = col_f32x2{r4[0], r4[1]};
          rr[l + 1] = col_f32x2{r4[2], r4[3]};
        }
      } else {
#pragma unroll
        for (int l = 0; l < L; ++l)
          rr[l] = __builtin_bit_cast(col_f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(o + 8u * l), 0, 0));
      }
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
    if (queue && tid == 0) next_idx[1] = atomicAdd(queue + xcd * WIN_QUEUE_STRIDE, 1u);
  }

  stamp(-1);
  while (have) {
    // (opaque per iteration: the compiler otherwise hoists every per-lane expression of tid / qi out of the item loop --
    // a dozen 64-bit piece offsets, float copies of tid, ... -- and spills them)
    asm volatile("" : "+v"(tid), "+v"(qi), "+v"(lane_slot), "+v"(lane_c0));
    const unsigned own = rec_index(qv, m);                                    // this lane's record: query * M + head

    // FUSED: the reference points sit in L2 (shared by the heads and the batch): fetched here, not a whole item ahead
    // (eight more registers alive across the gather spilled); the transposition below runs while they arrive
    col_f32x2 rr[L];
    issue_ref(b, qv, rr);
    // ---- the record: sampling locations (FUSED: reference point + offset / (W, H)), loaded one item ago ----------------
    col_f32x2 lxy[L][P];
    {
      if constexpr (!PP) {
#pragma unroll
        for (int g = 0; g < NGL; ++g) quad_transpose_in(raw[g], qi0, qi1);
      }
#pragma unroll
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < P; ++k) {
          const col_f32x4 pc = raw[(l * 2 + k / 2) / 4][(l * 2 + k / 2) & 3];
          col_f32x2 v = {pc[(k & 1) * 2], pc[(k & 1) * 2 + 1]};
          if constexpr (FUSED) v = __builtin_elementwise_fma(v, invWH[l], rr[l]);
          // from here on the PIXEL coordinates (w_im, h_im) = loc * (W, H) - 0.5 (cuh:283-288), formed once for the
          // boxes and the gather (packed FMAs: x and y in one instruction)
          const col_f32x2 px = __builtin_elementwise_fma(v, fWH[l], col_f32x2{-0.5f, -0.5f});
          // The reference gates a sample on -1 < w_im < W and -1 < h_im < H (cuh:290-296).  Here the gate is folded into
          // the coordinate itself: a coordinate that fails its test (NaN included: the comparison is false) moves to -2,
          // one that passes W moves to W.  The staged window carries a ZERO APRON of two pixels on each side (columns
          // -2, -1 and W, W + 1), so a sample moved there reads four apron pixels -- its contribution is exactly 0 and no
          // map pixel is touched, which is what the reference's `if` gives -- and every other sample keeps its own
          // coordinates.  No per-sample gate, select or predicate is left in the gather (it cost 13 vector instructions
          // per sample).
          lxy[l][k][0] = fminf(px[0] > -1.f ? px[0] : -2.f, fWH[l][0]);
          lxy[l][k][1] = fminf(px[1] > -1.f ? px[1] : -2.f, fWH[l][1]);
        }
    }

    // (everything that waits for the reference-point loads must be complete before the atomic below is issued)
    if constexpr (FUSED) {
#pragma unroll
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < P; ++k) asm volatile("" : "+v"(lxy[l][k]));
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp(7);                                                                 // records waited for, transposed, lxy formed
    // ---- publish the next item's index, fetch the one after it (its value is parked in LDS behind the staging barrier).
    // Issued behind the loads the pre-pass waits for: the memory counter is in-order, a wait for those would include it. ---
    unsigned f_new = 0u;
    bool fetched = false;
    if (tid == 0) {
      unsigned nxt = (unsigned)(item - xcd * chunk + nslots);                 // static stride when there is no queue
      if (queue) {
        const unsigned f_next = next_idx[1];
        if (f_next + 1u >= n_x)                                               // that was the launch's last fetch
          __hip_atomic_store(queue + xcd * WIN_QUEUE_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nxt = (unsigned)nslots + f_next;
        if (nxt < n_x) {
          // (inline asm: the compiler waits for a returning atomic at the end of this divergent block -- a full round
          // trip to L2 in front of the pre-pass; the wait now sits where the value is parked, behind the pre-pass)
          const unsigned one = 1u, zero = 0u;
          const unsigned *qp = queue + xcd * WIN_QUEUE_STRIDE;
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
      // min / max over the lane's four samples (already inside [-2, W] x [-2, H], see above); the gather floors the very
      // same registers.  Packed as u16 pairs biased by +2 (low corner) / +3 (high corner + 1: the box covers x0 .. x0 + 1).
      const float mnx = fminf(fminf(lxy[l][0][0], lxy[l][1][0]), fminf(lxy[l][2][0], lxy[l][3][0]));
      const float mxx = fmaxf(fmaxf(lxy[l][0][0], lxy[l][1][0]), fmaxf(lxy[l][2][0], lxy[l][3][0]));
      const float mny = fminf(fminf(lxy[l][0][1], lxy[l][1][1]), fminf(lxy[l][2][1], lxy[l][3][1]));
      const float mxy = fmaxf(fmaxf(lxy[l][0][1], lxy[l][1][1]), fmaxf(lxy[l][2][1], lxy[l][3][1]));
      const unsigned lo = (unsigned)(cvt_flr(mnx) + 2) | ((unsigned)(cvt_flr(mny) + 2) << 16);
      const unsigned hi = (unsigned)(cvt_flr(mxx) + 3) | ((unsigned)(cvt_flr(mxy) + 3) << 16);
      const unsigned red = wave_reduce_box(lo, hi);                 // lane 31: min lo, lane 63: ~max hi
      if ((lane & 31) == 31) bb[(l * NW + wave) * 2 + (lane >> 5)] = red;
    }
    // ---- this item's weights (FUSED: logits): fetched now, looked at after the staging barrier.  (Behind the pre-pass:
    // anything that waits on the memory counter there -- it is in-order -- would otherwise wait for these loads too.) ---------
    col_f32x4 wraw[NGW][4];
    auto load_weights = [&]() {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(attn + (long long)b * S * M * (L * P)), 0,
                                                        (int)((unsigned)S * (unsigned)M * (unsigned)(L * P * 4)), RSRC_FLAGS);
      if constexpr (PP) {                         // piece = level: the level's P = 4 weights of (query, head)
        const unsigned qo = (unsigned)(qv ^ (qv >> 31)) << 4;
        const unsigned s16 = (unsigned)__builtin_amdgcn_readfirstlane(S) * 16u;
        const unsigned p0 = (unsigned)__builtin_amdgcn_readfirstlane(m) * (unsigned)L * s16;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int g = 0; g < NGW; ++g)
            wraw[g][s4] = g * 4 + s4 < L
                              ? __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    rs, (int)qo, (int)(p0 + (unsigned)(g * 4 + s4) * s16), PCT_COL_W_NT ? 2 : 0))
                              : col_f32x4{0.f, 0.f, 0.f, 0.f};
        return;
      }
      unsigned off[4];
      quad_offsets(own * (unsigned)(L * P * 4), off);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int g = 0; g < NGW; ++g)
          wraw[g][s4] = (PCT_COL_KO_NOLOC == 2 || PCT_COL_KO_NOLOC == 3) ? col_f32x4{(float)off[s4], 0.1f, 0.2f, 0.3f} :
              __builtin_bit_cast(col_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off[s4] + g * 64), 0, PCT_COL_W_NT ? 2 : 0));
    };
    // A weight record is 64 bytes: HALF a 128-byte line, the other half being the sibling head's, which the neighbouring
    // workgroup fetches at about the same moment -- and two misses on one line in flight together are two fills on the
    // memory side (measured: the weights came in 1.75x (I) / 2.08x (M) over).  So the odd head of a pair asks later, once
    // the even head's fill has landed in L2.
    const bool w_late = PCT_COL_W_LATE && (m & 1);
    if (!w_late) load_weights();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // boxes in LDS before the barrier
    stamp(0);
    __syncthreads();                                                          // (A) boxes visible; pool free
    stamp(1);
    if (PCT_COL_PRIO == 3 || PCT_COL_PRIO == 4) __builtin_amdgcn_s_setprio(3);
    if (PCT_COL_PRIO == 8) __builtin_amdgcn_s_setprio(PCT_COL_PRIO_PLAN);

    // ---- the next item: its query per lane, its location loads issued (in flight until the next iteration) --------------
    int item_n, b_n = 0, m_n = 0, qv_n = 0;
    {
      const unsigned nxt = __builtin_amdgcn_readfirstlane(next_idx[0]);
      item_n = nxt < n_x ? xcd * chunk + (int)nxt : item_end;                // never out of the chunk
    }
    const bool have_n = item_n < item_end;             // (decoded once this item's first staging is issued, below)

    // ---- windows and phases (uniform) --------------------------------------------------------------------------------
    int wx0[L], wy0[L], wwid[L], whgt[L], wsize[L], wbase[L], phase_of[L];
    bool latin[L];
#pragma unroll
    for (int l = 0; l < L; ++l) latin[l] = false;
    bool starts_phase[L];
    {
#pragma unroll
      for (int l = 0; l < L; ++l) {
        unsigned lo, hi;
        block_box<NW>(bb + l * NW * 2, lo, hi);
        // un-bias (the origin may be -2: apron).  hi >= lo + 1 in both halves -- a box always holds a sample's two
        // columns and two rows -- so there is no empty case
        const unsigned lx = lo & 0xFFFFu, ly = lo >> 16;
        wx0[l] = (int)lx - 2;
        wy0[l] = (int)ly - 2;
        wwid[l] = (int)((hi & 0xFFFFu) - lx) + 1;
        // (conflict-free corner order, below: the four corners of a sample must fall into four different bank classes --
        // pool index mod 4 -- which they do when the window is 2 mod 4 pixels wide; the extra columns are staged like any other)
        // PCT_COL_LATIN == 2: only where the box is much larger than the column's cell on this level, i.e. where the lanes'
        // offsets differ by several pixels and their pixel classes collide (model-like offsets: halo ~14 + 14 pixels; the
        // init-like ones, whose classes are consecutive by construction, stay at 4 - 10 and keep the cheaper fixed order)
        if (PCT_COL_LATIN == 1) latin[l] = true;
        if (PCT_COL_LATIN == 2)
          latin[l] = (wwid[l] - col_max_cell(Ws[l], CX)) + ((int)((hi >> 16) - ly) + 1 - col_max_cell(Hs[l], CY)) >= PCT_COL_LATIN_HALO;
        if (latin[l]) wwid[l] += (2 - wwid[l]) & 3;
        whgt[l] = (int)((hi >> 16) - ly) + 1;
        wsize[l] = wwid[l] * whgt[l];
      }
      // levels are gathered in a fixed order (col_level_of_step); a level that does not fit beside the ones already
      // planned opens a new PHASE: the pool is re-staged right before it (two barriers)
      int ph = 0, used = 0;
      bool fresh = true;
#pragma unroll
      for (int ll = 0; ll < L; ++ll) {
        const int l = col_level_of_step<L>(ll);
        starts_phase[l] = false;
        // (never fits -- or a map too large for the 16-bit box corners --: gathered from global memory, which does not look at the box)
        if (wsize[l] > pool_use || big_map) {
          phase_of[l] = -1;
          wbase[l] = 0;
          continue;
        }
        if (used + wsize[l] > pool_use) {
          ++ph;
          used = 0;
          fresh = true;
        }
        phase_of[l] = ph;
        starts_phase[l] = fresh;
        fresh = false;
        wbase[l] = used;
        used += wsize[l];
      }
    }

    // EARLY second phase.  With model-like offsets the four windows of a column need two phases ({finest, coarsest} and the
    // two middle levels).  The finest level is gathered first and sits at the bottom of the pool, the second phase's windows
    // are planned from the bottom too: when they fit inside the finest window's region, they are staged right behind the
    // finest level's gather (one barrier: every wave is done with that region) and FLY while the coarsest level is gathered
    // from its own region above -- instead of the workgroup sitting through that flight between two barriers later.
    bool early_ok = false;
    if constexpr (PCT_COL_EARLY && L >= 3) {
      int used1 = 0;
      bool has1 = false;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        used1 += phase_of[l] == 1 ? wsize[l] : 0;
        has1 = has1 || phase_of[l] == 1;
      }
      early_ok = has1 && phase_of[L - 1] == 0 && phase_of[col_level_of_step<L>(1)] == 0 && wbase[L - 1] == 0 &&
                 used1 <= wsize[L - 1];
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
    // a sample's geometry: first corner and the four bilinear * attention weights (the gate of cuh:290-296 is already in
    // the coordinates, see where lxy is formed)
    struct Geo {
      col_f32x2 g12, g34;                 // (hh * hw, hh * lw) * w and (lh * hw, lh * lw) * w
      int x0, y0;
    };
    auto geometry = [&](auto lc, auto kc) {
      constexpr int l = decltype(lc)::value;
      constexpr int k = decltype(kc)::value;
      // (opaque: without it the compiler hoists every sample's geometry to the top of the item and keeps it alive)
      asm volatile("" : "+v"(lxy[l][k]));                                      // (in place: the coordinates are not needed again)
      const col_f32x2 pix = lxy[l][k];
      Geo g;
      g.x0 = cvt_flr(pix[0]);
      g.y0 = cvt_flr(pix[1]);
      const float lw = __builtin_amdgcn_fractf(pix[0]), lh = __builtin_amdgcn_fractf(pix[1]);
      const col_f32x2 ax = {1.f - lw, lw}, ay = {1.f - lh, lh};                // (hw, lw), (hh, lh): pairs as the packed ops take them
      const col_f32x2 t = ax * col_f32x2{wts[l][k], wts[l][k]};                // (hw, lw) * w
      g.g12 = t * col_f32x2{ay[0], ay[0]};
      g.g34 = t * col_f32x2{ay[1], ay[1]};
      return g;
    };

    // One level from its LDS window, one pixel ROW of a sample at a time (two corners = 8 x 16 B per lane in flight): the
    // other waves of the SIMD cover the LDS latency.  (A rolling pipeline with the next row's reads in flight during the
    // FMAs needs 32 more data registers than the 168 of three workgroups per CU leave: it spilled and was slower.)
    // All four corners lie inside the staged window (out-of-map ones are apron zeros).
    auto gather_level_lds = [&](auto lc) {
      constexpr int l = decltype(lc)::value;
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
        ([&] {
          const Geo g = geometry(lc, std::integral_constant<int, Ks>{});
          // pixel (x0, y0) sits at pool index woff + y0 * width + x0 (woff folds the window origin and base: one scalar)
#if PCT_COL_KO_NOCONF
          const unsigned a = (unsigned)(((__mul24(g.y0, wwid[l]) + g.x0 + woff[l]) & ~3) | (lane & 3)) << 6;
#else
          const unsigned a = (unsigned)(__mul24(g.y0, wwid[l]) + g.x0 + woff[l]) << 6;
#endif
          const unsigned rowb = (unsigned)wwid[l] << 6;
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
    // The same level with a CONFLICT-FREE corner order.  The LDS serves a ds_read_b128 in groups of 16 lanes, four lanes of
    // each 8-lane block (one QUAD) per group; the piece rotation above keeps the quads of a group on different bank
    // quarters, so what is left to collide are the four lanes of a quad whose pixels fall into the same class (pool index
    // mod 4: pixels are 64 B = 16 banks apart) -- with model-like offsets 59 % of all LDS cycles of the kernel.  A sample's
    // four corners are the pool indices i, i + 1, i + width, i + width + 1: with the window width = 2 (mod 4) they are one
    // pixel of EACH class.  So lane q of the quad starts with the corner of class q and walks the classes upwards: at every
    // corner step the quad's four lanes read four different classes, whatever the locations.  Price: the corner order
    // differs per lane and sample -- offsets and weights are selected by two bits of (q - i) -- ~27 vector instructions
    // per sample more than the fixed order.  The sums are accumulated in a different order per lane than with the fixed
    // order (results agree to rounding, not bit for bit).
    auto gather_level_lds_latin = [&](auto lc) {
      constexpr int l = decltype(lc)::value;
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
        ([&] {
          asm volatile("" : "+v"(lxy[l][Ks]));
          const col_f32x2 pix = lxy[l][Ks];
          const int x0 = cvt_flr(pix[0]), y0 = cvt_flr(pix[1]);
          const float lw = __builtin_amdgcn_fractf(pix[0]), lh = __builtin_amdgcn_fractf(pix[1]);
          const float hw = 1.f - lw, hh = 1.f - lh;
          const col_f32x2 tw = col_f32x2{hw, lw} * col_f32x2{wts[l][Ks], wts[l][Ks]};      // (hw, lw) * w, as the fixed order
          const unsigned pi = (unsigned)(__mul24(y0, wwid[l]) + x0 + woff[l]);
          const unsigned e = ((unsigned)qi - pi) & 3u;          // first corner d = dx + 2 dy of this lane; then d + 1, d + 2, d + 3 (mod 4)
          const bool e0 = e & 1u, e1 = e & 2u, f1 = (e + 1u) & 2u;                          // dy of the steps: e1, f1, !e1, !f1
          const float wa = e0 ? tw[1] : tw[0], wb = e0 ? tw[0] : tw[1];                    // dx of the steps: e0, !e0, e0, !e0
          const unsigned rowb = (unsigned)wwid[l] << 6;
          const unsigned base = pi << 6;
          const unsigned xa = e0 ? 64u : 0u, xb = xa ^ 64u;
          const unsigned a0 = base + xa + (e1 ? rowb : 0u), a1 = base + xb + (f1 ? rowb : 0u);
          const unsigned a2 = base + xa + (e1 ? 0u : rowb), a3 = base + xb + (f1 ? 0u : rowb);
          const float w0 = wa * (e1 ? lh : hh), w1 = wb * (f1 ? lh : hh), w2 = wa * (e1 ? hh : lh), w3 = wb * (f1 ? hh : lh);
          {
            col_f32x4 va[4], vb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              va[j] = *reinterpret_cast<const col_f32x4 *>(pool + (a0 + rot[j]));
              vb[j] = *reinterpret_cast<const col_f32x4 *>(pool + (a1 + rot[j]));
            }
            fma_row(va, vb, w0, w1);
          }
          pin_acc();
          __builtin_amdgcn_sched_barrier(0);
          {
            col_f32x4 va[4], vb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              va[j] = *reinterpret_cast<const col_f32x4 *>(pool + (a2 + rot[j]));
              vb[j] = *reinterpret_cast<const col_f32x4 *>(pool + (a3 + rot[j]));
            }
            fma_row(va, vb, w2, w3);
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
          const bool top = (unsigned)g.y0 < (unsigned)H, bot = (unsigned)(g.y0 + 1) < (unsigned)H;   // (y0, x0 in [-2, H] x [-2, W])
          const bool lft = (unsigned)g.x0 < (unsigned)W, rgt = (unsigned)(g.x0 + 1) < (unsigned)W;
          // (PP: piece c of pixel p of head m at ((m * 4 + c) * S + p) * 16: the pixel stride is 16 bytes, the piece a plane)
          const unsigned pxs = PP ? 16u : MDb;
          const unsigned a = PP ? (unsigned)(m * (D / 4) * S + St[l] + g.y0 * W + g.x0) * 16u
                                : (unsigned)(St[l] + g.y0 * W + g.x0) * MDb + (unsigned)(m * D) * 4u;
          const unsigned o1 = (top && lft) ? a : OOB, o2 = (top && rgt) ? a + pxs : OOB;
          const unsigned o3 = (bot && lft) ? a + (unsigned)W * pxs : OOB, o4 = (bot && rgt) ? a + (unsigned)W * pxs + pxs : OOB;
          [&]<int... Ss>(std::integer_sequence<int, Ss...>) {                    // one member at a time (16 registers of data)
            ([&] {
              constexpr int CT = BcastCtrl<4, Ss>::value;
              const unsigned pc = PP ? (unsigned)((Ss - qi) & 3) * ((unsigned)S * 16u)     // (an out-of-range offset stays out of range)
                                     : (unsigned)((Ss - qi) & 3) << 4;
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
          // (PP: piece cc of a pixel lives in plane m * 4 + cc, pixels 16 bytes apart: a copy instruction reads four runs of
          // 256 consecutive bytes instead of sixteen 64-byte slices 512 bytes apart)
          const unsigned lvl_off = PP ? ((unsigned)(m * (D / 4)) * (unsigned)S + (unsigned)St[l]) * 16u
                                      : (unsigned)St[l] * MDb + (unsigned)(m * D) * 4u;   // bytes, this level and head
          const unsigned row_bytes = (unsigned)Ws[l] * (PP ? 16u : MDb);
          unsigned char *dst = pool + (size_t)wbase[l] * PXB;
          for (int c0 = 0; c0 < wwid[l]; c0 += CPX) {
            const int xw = c0 + dx, x = wx0[l] + xw;
            const unsigned voff = (unsigned)x < (unsigned)Ws[l]
                                      ? (PP ? (unsigned)x * 16u + (unsigned)cc * ((unsigned)S * 16u) : (unsigned)x * MDb + (unsigned)(cc * 16))
                                      : OOB;
            if (xw < wwid[l]) {
              for (int r = wv; r < whgt[l]; r += NW) {
                const int y = wy0[l] + r;
                const bool in_y = (unsigned)y < (unsigned)Hs[l];
                if (PCT_COL_KO_NOSTAGE) continue;
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
      if constexpr (!PP) {
#pragma unroll
        for (int g = 0; g < NGW; ++g) quad_transpose_in(wraw[g], qi0, qi1);
      }
#pragma unroll
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < P; ++k) wts[l][k] = wraw[l / 4][l & 3][k];
      if constexpr (FUSED) {                                                  // softmax over the record's L * P logits
        // exp(x - mx) as exp2(x * log2(e) - mx * log2(e)): one packed FMA per logit pair instead of a subtraction and a
        // multiplication per logit; sums and the final scaling packed as well; 1 / sum as v_rcp_f32 (1 ulp) instead of
        // the ten-instruction IEEE division -- 50 vector instructions instead of 98 per (query, head)
        static_assert(P % 2 == 0, "logit pairs");
        float mx = -INFINITY;
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
          for (int k = 0; k < P; ++k) mx = fmaxf(mx, wts[l][k]);
        constexpr float LOG2E = 1.44269504088896340736f;
        const float nm = -mx * LOG2E;
        col_f32x2 sum2 = {0.f, 0.f};
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
          for (int k = 0; k < P; k += 2) {
            const col_f32x2 a = __builtin_elementwise_fma(col_f32x2{wts[l][k], wts[l][k + 1]}, col_f32x2{LOG2E, LOG2E},
                                                          col_f32x2{nm, nm});
            wts[l][k] = __builtin_amdgcn_exp2f(a[0]);
            wts[l][k + 1] = __builtin_amdgcn_exp2f(a[1]);
            sum2 += col_f32x2{wts[l][k], wts[l][k + 1]};
          }
        const float inv = __builtin_amdgcn_rcpf(sum2[0] + sum2[1]);
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
          for (int k = 0; k < P; k += 2) {
            const col_f32x2 w2 = col_f32x2{wts[l][k], wts[l][k + 1]} * col_f32x2{inv, inv};
            wts[l][k] = w2[0];
            wts[l][k + 1] = w2[1];
          }
      }
    };

    auto level_step = [&](auto llc) {
      constexpr int ll = decltype(llc)::value;
      constexpr int l = col_level_of_step<L>(ll);
      if (starts_phase[l]) {
        if (ll > 0 && early_ok && phase_of[l] == 1) {
          // staged early (below): landed + visible.  The memory counter is in order and the NEXT item's location record
          // was requested behind that staging: wait for everything BUT those youngest loads -- a plain barrier's
          // vmcnt(0) made every two-phase item sit through the best part of an HBM round trip here
#if PCT_COL_EARLY_WAIT
          if (have_n) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NGL * 4) : "memory");
          } else
#endif
            __syncthreads();
        } else {
          if (ll > 0) __syncthreads();                                        // every wave is done with the pool
          stage_phase(phase_of[l]);
          if (ll == 0) stamp(8);                                              // first phase's LDS-DMA issued
          if constexpr (ll == 0) {
            if (have_n) decode(item_n, b_n, m_n, qv_n);                        // while the LDS-DMA pieces are in flight
            if (w_late) load_weights();
          }
          if (ll == 0) stamp(3);
          __syncthreads();                                                    // windows staged (vmcnt(0) + barrier)
          if (ll == 0) stamp(4);
          if (ll == 0 && PCT_COL_PRIO == 4) __builtin_amdgcn_s_setprio(0);
          if (ll == 0 && PCT_COL_PRIO == 8 && PCT_COL_PRIO_FRONT) __builtin_amdgcn_s_setprio(PCT_COL_PRIO_L0);   // weights / soft-max too
        }
      }
      if constexpr (ll == 0) {
        // (thread 0) park the counter value fetched at the top of the item: the weights below need the memory counter at
        // zero anyway, so this wait is free -- anywhere earlier it would stall on the loads issued since
        if (fetched) {
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(f_new)::"memory");
          next_idx[1] = f_new;
        }
        front_end();
        stamp(9);                                                             // weights waited for, transposed (FUSED: soft-max)
      }
      if (PCT_COL_PRIO == 6) __builtin_amdgcn_s_setprio(ll + 1 < 3 ? ll + 1 : 3);     // (the further into the item, the more urgent)
      if (PCT_COL_PRIO == 7) __builtin_amdgcn_s_setprio(ll == 0 ? 3 : 2);
      if (PCT_COL_PRIO == 8) __builtin_amdgcn_s_setprio(ll == 0 ? PCT_COL_PRIO_L0 : PCT_COL_PRIO_REST);
      if (PCT_COL_KO_NOGATHER == 2) {
      } else if (phase_of[l] >= 0) {
        if constexpr (PCT_COL_LATIN == 1) gather_level_lds_latin(std::integral_constant<int, l>{});
        else if constexpr (PCT_COL_LATIN == 2) {
          if (latin[l]) gather_level_lds_latin(std::integral_constant<int, l>{});
          else gather_level_lds(std::integral_constant<int, l>{});
        } else
          gather_level_lds(std::integral_constant<int, l>{});
      }
      else gather_level_global(std::integral_constant<int, l>{});             // box larger than the pool: global memory
      // (marking the global-memory path unlikely makes hipcc outline it behind a real call: 544 bytes of scratch per lane, 10x slower)
      if (ll == 0) stamp(10);                                                 // first level gathered
      if constexpr (ll == 0 && PCT_COL_EARLY && L >= 3) {
        if (early_ok) {
          __syncthreads();                                                    // every wave is done with the finest window
          stage_phase(1);
        }
      }
      // one group of the next item's location record per level: its registers are the ones this level's points freed
      // the next item's location record (32 registers: the ones this level's points and the weights freed)
      if constexpr (ll == (PCT_COL_LOC_AT < L ? PCT_COL_LOC_AT : L - 1)) {
        if (have_n) issue_loc(b_n, m_n, qv_n, raw);
      }

    };
    if (!starts_phase[L - 1]) {                                               // (the finest level is gathered from global)
      if (have_n) decode(item_n, b_n, m_n, qv_n);
      if (w_late) load_weights();
      stamp(3);
      stamp(4);
    }
    if (PCT_COL_PRIO == 1) __builtin_amdgcn_s_setprio(3);
    if (PCT_COL_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    if (PCT_COL_PRIO == 5) __builtin_amdgcn_s_setprio(1);
    [&]<int... LLs>(std::integer_sequence<int, LLs...>) {
      (level_step(std::integral_constant<int, LLs>{}), ...);
    }(std::make_integer_sequence<int, L>{});
    if (PCT_COL_PRIO == 1 || PCT_COL_PRIO == 3 || (PCT_COL_PRIO >= 5 && PCT_COL_PRIO < 8)) __builtin_amdgcn_s_setprio(0);
    if (PCT_COL_PRIO == 8) __builtin_amdgcn_s_setprio(PCT_COL_PRIO_TOP);
    if (PCT_COL_PRIO == 2) __builtin_amdgcn_s_setprio(3);
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
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(out + (long long)b * S * MD, 0,
                                                        (int)((unsigned)S * (unsigned)MD * 4u), RSRC_FLAGS);
      // an idle lane's record goes out of range (the descriptor's bounds check drops the store)
      // (the record index is formed again rather than kept alive through the gather)
      const unsigned own_o = (rec_index(qv, m) * (unsigned)(D * 4)) | ((unsigned)(qv >> 31) & 0x80000000u);
      const unsigned oo[4] = {dpp_u<0x00>(own_o), dpp_u<0x55>(own_o), dpp_u<0xAA>(own_o), dpp_u<0xFF>(own_o)};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        if (!PCT_COL_KO_NOSTORE || w4[s4][0] == 1.2345e30f)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(col_u32x4, w4[s4]), rs,
                                               (int)(oo[s4] + ((unsigned)(((qi - s4) + (int)rho) & 3) << 4)), 0,
                                               (STREAM_NT || PCT_COL_STORE_NT) ? 2 : 0);
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
      for (int i = 0; i < 16; ++i) stamps[(size_t)blockIdx.x * 16 + i] = t_sum[i];
  }
}

unsigned long long *win_stamp_buffer();                                      // msda_forward_win.hip (diagnostic)

// ---- launcher: returns -100 when this geometry is not covered (caller uses another kernel) ----------------------------
// ref == nullptr: plain op; ref != nullptr: fused front-end (loc = raw offsets, attn = raw logits).
int launch_msda_forward_col(const void *value, const int64_t *shapes, const int64_t *starts, const void *loc,
                            const void *attn, int N, int S, int M, int D, int L, int Lq, int P, void *out,
                            hipStream_t stream, const float *ref, long long ref_batch_stride, bool planes)
{
  if ((((uintptr_t)value | (uintptr_t)out | (uintptr_t)loc | (uintptr_t)attn) & 15u)) return -100;
  if (ref && (((uintptr_t)ref) & ((L % 2 == 0) ? 15u : 7u))) return -100;
  if (ref && (L % 2 == 0) && (ref_batch_stride & 3)) return -100;
  if (D != 16 || P != 4 || L < 3 || L > 5 || Lq != S || M < 1) return -100;
  if ((long long)N * ((long long)S + 4096) * M >= 0x7fffffffLL) return -100;   // item / record arithmetic headroom
  if ((long long)S * M * L * P * 8 >= 0x7fffffffLL) return -100;               // 32-bit byte offsets inside an image, all tensors
  if ((long long)S * M * D * 4 >= 0x7fffffffLL) return -100;                   // (0x80000000 is the out-of-range sentinel)
  if (S >= (1 << 24) || M >= (1 << 16)) return -100;                           // 24-bit multiplies in the record index
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
  static const int grid_env = [] { const char *e = getenv("PCT_COL_GRID_WG"); return e ? atoi(e) : 0; }();   // (diagnostic: workgroups per CU launched)
  const dim3 grid(256 * ((grid_env >= 1 && grid_env <= wg_per_cu) ? grid_env : wg_per_cu)), block(BLOCKV);
  unsigned *queue = win_queue_slot(stream);                                    // nullptr: static item stride
  const float *v = static_cast<const float *>(value);
  const float *lc = static_cast<const float *>(loc), *at = static_cast<const float *>(attn);
  float *o = static_cast<float *>(out);
#define PCT_COL_K(L_, FU_, B_, ST_)                                                                                     \
  do {                                                                                                                  \
    const hipError_t attr_rc = func_attr_per_device(reinterpret_cast<const void *>(&msda_forward_col_kernel<L_, FU_, B_, ST_>)); \
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
  if (planes) {                         // piece-plane operands (see the kernel): 256-thread workgroups only
    if (BLOCKV != 256) return -100;
#define PCT_COL_PP(L_, FU_)                                                                                             \
  do {                                                                                                                  \
    const hipError_t attr_rc = func_attr_per_device(reinterpret_cast<const void *>(&msda_forward_col_kernel<L_, FU_, 256, false, true>)); \
    if (attr_rc != hipSuccess) return (int)attr_rc;                                                                     \
    hipLaunchKernelGGL((msda_forward_col_kernel<L_, FU_, 256, false, true>), grid, block, lds, stream, v, shapes, starts, lc, \
                       at, N, S, M, pool_px, o, ref, ref_batch_stride, queue, nullptr);                                 \
  } while (0)
    if (ref) {
      if (L == 3) PCT_COL_PP(3, true);
      else if (L == 4) PCT_COL_PP(4, true);
      else PCT_COL_PP(5, true);
    } else {
      if (L == 3) PCT_COL_PP(3, false);
      else if (L == 4) PCT_COL_PP(4, false);
      else PCT_COL_PP(5, false);
    }
#undef PCT_COL_PP
    return (int)hipGetLastError();
  }
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
