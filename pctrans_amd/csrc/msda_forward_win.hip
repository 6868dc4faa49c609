// MSDeformAttn forward, windowed-LDS gather ("v2") for MI355X (gfx950, wave64).
//
// Same semantics as msda_forward.hip (reference: ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 + :38-89).
//
// Why: the straightforward gather moves 4 corners x 64 B per sample = L*P*4*64 B (16-32 KB) per query through the
// vector-L1 / texture-address path -- ~13x the algorithmic bytes -- and that path, not HBM, bounds the v1 kernel at
// ~20 % of the HBM roofline.  LDS serves 256 B/clk/CU (ds_read_b128), 4x the L1 path, so:
//
//   * work item = (image b, query tile, head m); a tile is TQ = 256/QL queries (QL = lanes per head-pixel, 16 B each):
//     an 8x8 (QL=4) / 16x8 (QL=2) pixel block of one pyramid level when the queries are the pyramid's own pixels
//     (Lq == S, PCTrans' encoder self-attention), otherwise TQ consecutive queries.  Which queries share a tile only
//     affects speed, never results.
//   * each lane of a query's QL-lane group OWNS points p == lane (mod QL) of every level: it alone loads their
//     (x, y, weight) (8-B + 4-B loads, no LDS staging, no D-fold redundancy), derives the bilinear geometry once and
//     later broadcasts it to its siblings with DPP quad_perm moves (register crossbar, no LDS traffic);
//   * pre-pass: per level the bounding box of every in-map corner the tile touches is reduced with packed-u16
//     min/max over DPP + readlane + one LDS hop across the 4 waves; boxes are packed into an LDS pool
//     (64 B per head-pixel) while they fit, the box rows are staged with coalesced 16-B loads, and every sample of a
//     staged level is then gathered from LDS.  Out-of-map corners point at a zero pixel in LDS (no 0 * Inf hazards).
//     A level whose box does not fit (coarse-level tiles looking at a fine level, scattered locations) falls back to
//     the global gather for that level only, still with owner-computed geometry;
//   * persistent grid: WG g serves XCD (g % 8); each XCD walks one contiguous chunk of the (b, tile, m) items with the
//     8 heads of a tile adjacent, so the 512-B pixel lines a tile's heads share are fetched into one L2.
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <utility>

#include "msda_win_common.hpp"

namespace pct {

// NS = query slots per lane: one work item covers NS * (256 / QL) queries, so the per-item work (bounding boxes,
// window set-up, staging, barriers) is amortised over NS gathers.
// FUSED: `loc` holds raw sampling offsets, `attn` raw attention logits, `ref` the reference points; the kernel forms
// location = ref[q, l] + offset / (W_l, H_l) and weight = softmax over the record's L*P logits itself
// (ops/modules/ms_deform_attn.py:100-109), as msda_forward_dpp.hip does.
template <typename T, int D, int L, int P, int NS, bool FUSED, bool STAMP = false>
__global__ __launch_bounds__(WIN_BLOCK, (NS == 4 || P == 8) ? 3 : 4) void msda_forward_win_kernel(   // 50 KB LDS -> 3 WG/CU
    const typename Traits<T>::store_t *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const float *__restrict__ loc, const float *__restrict__ attn, const int N,
    const int S, const int M, const int Lq, const int pyramid, const int pool_px,
    typename Traits<T>::store_t *__restrict__ out, const float *__restrict__ ref, const long long ref_batch_stride,
    unsigned *__restrict__ queue, unsigned long long *__restrict__ stamps = nullptr)
{
  using ST = typename Traits<T>::store_t;
  constexpr int VEC = 16 / (int)sizeof(ST);       // channels per lane
  constexpr int QL = D / VEC;                     // lanes per (query, head) = per head-pixel
  constexpr int TQ = WIN_BLOCK / QL;              // queries per slot
  // pyramid-mode tile: TW pixels wide, SH rows per slot, NS slots stacked -> 8x8 (NS=1), 8x16 (NS=2), 16x16 (NS=4)
  // for QL = 4 (squarer tiles = smaller halo per query)
  constexpr int TW = (TQ / WIN_TH) * (NS >= 4 ? 2 : 1);
  constexpr int SH = TQ / TW;                     // rows per slot
  constexpr int THT = SH * NS;                    // tile height
  constexpr int PPL = P / QL;                     // points each lane owns per level
  constexpr int PXB = QL * 16;                    // bytes per head-pixel
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
  static_assert(D % VEC == 0 && (QL == 2 || QL == 4) && P % QL == 0 && L <= WIN_MAXL, "unsupported geometry");
  using v16 = vec_t<ST, VEC>;
  typedef float f32x2 __attribute__((ext_vector_type(2)));

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char *pool = smem_raw;                                   // the level windows (head-pixels of PXB bytes)
  unsigned *bb = reinterpret_cast<unsigned *>(smem_raw + (size_t)pool_px * PXB);   // [2][4 waves][L][2]
  unsigned *next_idx = bb + 2 * (WIN_BLOCK / 64) * WIN_MAXL * 2;                    // the workgroup's next item (index in the chunk)

  const int tid = threadIdx.x, wave = tid >> 6;
  const int c = tid & (QL - 1);                   // lane within the query's group
  const int j = tid / QL;                         // query position within a slot
  const int MD = M * D;

  // level geometry (uniform -> SGPRs) and the tile census
  int Hs[L], Ws[L], St[L], tiles_before[L + 1];
  tiles_before[0] = 0;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
    St[l] = (int)starts[l];
    tiles_before[l + 1] = tiles_before[l] + ((Hs[l] + THT - 1) / THT) * ((Ws[l] + TW - 1) / TW);
  }
  // FUSED: offsets are divided by (W_l, H_l); the quotient is formed as offset * (1 / W_l) with the reciprocal
  // computed once per level (<= 1 ulp from the true quotient of the reference, i.e. < 1e-6 pixel)
  float invW[L], invH[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    invW[l] = 1.0f / (float)Ws[l];
    invH[l] = 1.0f / (float)Hs[l];
  }
  const int T_img = pyramid ? tiles_before[L] : (Lq + TQ * NS - 1) / (TQ * NS);
  const int items = N * T_img * M;

  // pixels 0 and 1 of the pool are zeros: gated-out samples read them (weight 0 times a guaranteed-finite value)
  if (tid < 2 * QL) reinterpret_cast<vec_t<float, 4> *>(pool)[tid] = vec_t<float, 4>{0.f, 0.f, 0.f, 0.f};

  // persistent, XCD-chunked walk over the items
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int chunk = (items + 7) / 8;
  const int item_end = min((xcd + 1) * chunk, items);

  // Per item two barriers: (A) boxes visible + pool free, (B) windows staged.  The boxes of item i+1 are computed
  // right after the gather of item i, into the other half of `bb`, so the waves that finish their gather early spend
  // the wait on the next item's point loads instead of idling at a third barrier.
  int m = 0, b = 0;
  // this lane's NS queries: slot s is query q0 + s * dq of image b (dq uniform), valid iff bit s of okmask; the record
  // index ((b * Lq + q) * M + m) is formed where it is used (one 64-bit mad on a uniform base)
  int q0 = 0, dq = 0;
  unsigned okmask = 0;
  // the item's sampling locations (FUSED: already ref + offset / (W, H)), loaded ONCE by the box pre-pass and carried in
  // registers to the gather -- the gather used to read sampling_locations a second time, 29 us later and long out of L2
  // (1.43 GB per launch at batch 64, a sixth of the kernel's traffic).  Only while the item's locations fit 32 VGPRs
  // (NS * L * PPL pairs) under the 168-VGPR budget of the 16 x 16 tile; the other geometries reload them per slot as
  // before (CARRY = false: lxy dies with the pre-pass).
  constexpr bool CARRY = NS == 4 && P == 4 && NS * L * PPL * 2 <= 32;
  f32x2 lxy[NS][L][PPL];
  auto prepass = [&](const int item, unsigned *bbw) {
    m = item % M;
    const int bt = item / M;
    const int t = bt % T_img;
    b = bt / T_img;

    // ---- this lane's NS queries ----------------------------------------------------------------------------------
    int qdup[NS];                                 // pre-pass only: q, or a valid query of the tile when q is past the edge
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
      okmask = 0;
      dq = pyramid ? SH * Wq : TQ;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        int q;
        bool ok;
        if (pyramid) {
          const int qy = ty * THT + s * SH + j / TW, qx = tx * TW + (j % TW);
          ok = qy < Hq && qx < Wq;
          q = Sq + qy * Wq + qx;
          qdup[s] = Sq + min(qy, Hq - 1) * Wq + min(qx, Wq - 1);
        } else {
          q = (t * NS + s) * TQ + j;
          ok = q < Lq;
          qdup[s] = min(q, Lq - 1);
        }
        okmask |= ok ? 1u << s : 0u;
        if (s == 0) q0 = q;
      }
    }

    // ---- pre-pass: per-level bounding box (incl. 1-pixel apron) of every corner the tile touches ---------------
    {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        // a lane past the tile's edge repeats a valid query of the tile: it cannot move the box, and nothing below
        // has to exclude it
        const float *lrec = loc + (((long long)b * Lq + qdup[s]) * M + m) * (L * P * 2) + c * 2;
        const float *rrow = FUSED ? ref + b * ref_batch_stride + (long long)qdup[s] * (L * 2) : nullptr;
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
          for (int k = 0; k < PPL; ++k) {
            f32x2 v = *reinterpret_cast<const f32x2 *>(lrec + (l * P + k * QL) * 2);
            if constexpr (FUSED) {
              const f32x2 r = *reinterpret_cast<const f32x2 *>(rrow + 2 * l);
              v = f32x2{fmaf(v[0], invW[l], r[0]), fmaf(v[1], invH[l], r[1])};
            }
            lxy[s][l][k] = v;
          }
      }
#pragma unroll
      for (int l = 0; l < L; ++l) {
        // Box of the FIRST corner (x0, y0) = floor(w_im, h_im) over the lane's points, on floats: floor is monotone, so
        // min/max commute with it and one floor per lane and level replaces one per point.  A sample is gated in iff
        // -1 < w_im < W (and likewise h): clamping to [-1, W - 0.5] maps every gated-out coordinate onto a value a
        // gated-in sample could have, so such samples can only widen the box towards the map border, never past the
        // 1-pixel apron (NaN clamps to -1).  Same expression as the gather's, so both sides floor the same number.
        float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int k = 0; k < PPL; ++k) {
            const float h_im = lxy[s][l][k][1] * Hs[l] - 0.5f, w_im = lxy[s][l][k][0] * Ws[l] - 0.5f;
            const float wc = __builtin_amdgcn_fmed3f(w_im, -1.f, (float)Ws[l] - 0.5f);
            const float hc = __builtin_amdgcn_fmed3f(h_im, -1.f, (float)Hs[l] - 0.5f);
            mnx = fminf(mnx, wc);
            mxx = fmaxf(mxx, wc);
            mny = fminf(mny, hc);
            mxy = fmaxf(mxy, hc);
          }
        // biased by +1 to stay unsigned: x0 in [-1, W-1] -> [0, W]; the box covers x0 .. x0 + 1
        unsigned lo = (unsigned)((int)floorf(mnx) + 1) | ((unsigned)((int)floorf(mny) + 1) << 16);
        unsigned hi = (unsigned)((int)floorf(mxx) + 2) | ((unsigned)((int)floorf(mxy) + 2) << 16);
        lo = wave_reduce_pk<true>(lo);
        hi = wave_reduce_pk<false>(hi);
        if ((tid & 63) == 0) {
          bbw[(wave * L + l) * 2] = lo;
          bbw[(wave * L + l) * 2 + 1] = hi;
        }
      }
    }
    // The boxes must have reached the LDS before this wave arrives at barrier (A).  With the pre-pass in the loop
    // latch hipcc (ROCm 7.2) emits that s_barrier at the loop header WITHOUT the s_waitcnt lgkmcnt(0) it places before
    // every other barrier, and 1 launch in ~8 read stale boxes (caught by the adjoint test): wait explicitly.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  constexpr int BB_HALF = (WIN_BLOCK / 64) * WIN_MAXL * 2;
  // Items are handed out dynamically inside an XCD's chunk: the first round is static (workgroup slot), every further
  // item is nslots + the XCD's counter (`queue[xcd]`, zero between launches), fetched one item ahead by one lane and
  // passed through LDS.  Every processed item costs exactly one fetch, so the fetch that returns n_x - 1 (n_x = items
  // of this XCD) is the launch's last one and puts the counter back to zero: no per-launch memset, graph-replay safe.  With the static stride the 8 heads of a tile -- which share every 128-B line of the tile's windows,
  // two heads per line -- drifted apart by more than the ~6 us a line survives in the 4 MB L2 at this kernel's
  // traffic, so each head fetched the lines again; the counter starts the heads of a tile within ~2 us of each other.
  int item = xcd * chunk + slot0;
  int par = 0;
  stamp(-1);
  if (item < item_end) prepass(item, bb);
  while (item < item_end) {
    const unsigned *bbr = bb + par * BB_HALF;
    stamp(0);
    __syncthreads();                                               // (A) boxes visible; pool free
    stamp(1);
    // (the counter's answer is first looked at right before barrier (2): the round trip hides behind the staging)
    unsigned rfetch = 0u;
    if (queue && tid == 0) rfetch = atomicAdd(queue + xcd * WIN_QUEUE_STRIDE, 1u);

    // ---- windows (identical in every lane; kept in SGPRs) -------------------------------------------------------
    int wx0[L], wy0[L], wwid[L], wbase[L], wsize[L], in_lds[L];
    {
      int used = 0;
#pragma unroll
      for (int ll = 0; ll < L; ++ll) {
        const int l = L - 1 - ll;       // last level first: PCTrans orders levels coarse -> fine and the finest
                                        // level has the largest window and the most samples worth keeping in LDS
        const LevelWindow w = read_window(bbr, L, l);
        wx0[l] = w.x0;
        wy0[l] = w.y0;
        wwid[l] = w.wid;
        wsize[l] = w.size;
        in_lds[l] = used + w.size <= pool_px - 2 ? 1 : 0;
        wbase[l] = used + 2;                                        // pixels 0, 1 are the zero pixels
        used += in_lds[l] ? w.size : 0;
      }
    }
    stamp(2);

    // ---- stage the boxes that fit by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, every piece of every
    // level in flight at once; LDS address of a piece = wave-uniform base + lane*16 = dst + i*16 for
    // i = it*256 + wave*64 + lane; the global source is per lane (apron lanes read a 16-byte zero constant) ---------
    const ST *vimg = value + (long long)b * S * MD + m * D;         // this image, this head
#pragma unroll
    for (int l = 0; l < L; ++l) {
      if (in_lds[l] && wsize[l] > 0) {
        const float inv_w = 1.0f / (float)wwid[l];
        const int n16 = wsize[l] * QL;
        const ST *vlev = vimg + (long long)St[l] * MD;
        unsigned char *dst = pool + (size_t)wbase[l] * PXB;
        for (int it = 0; it * WIN_BLOCK < n16; ++it) {
          const int i = it * WIN_BLOCK + tid;
          if (i < n16) {
            const int px = i / QL, cc = i & (QL - 1);
            const int r = (int)(((float)px + 0.5f) * inv_w);
            const int y = wy0[l] + r, x = wx0[l] + px - r * wwid[l];
            const bool inside = y >= 0 && y < Hs[l] && x >= 0 && x < Ws[l];
            const ST *src = inside ? vlev + (long long)(y * Ws[l] + x) * MD + cc * VEC
                                   : reinterpret_cast<const ST *>(g_zero16);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src),
                (__attribute__((address_space(3))) void *)(dst + (size_t)(it * WIN_BLOCK + (tid & ~63)) * 16), 16, 0, 0);
          }
        }
      }
    }
    stamp(3);
    // ---- gather, one slot after the other.  Locations come from the registers the pre-pass filled; a slot's attention
    // weights (FUSED: logits) are only ISSUED here and consumed a whole slot later (`finish_points`), so the wave never
    // sits on a fresh load: the first slot's are in flight across barrier (2) together with the LDS-DMA pieces, slot
    // s+1's during slot s's gather. ------------------------------------------------------------------------------------
    float nx[L][PPL], ny[L][PPL], nw[L][PPL];
    f32x2 nr[(FUSED && !CARRY) ? L : 1];
    const long long rec_base = (long long)b * Lq * M + m;          // uniform
    auto issue_points = [&](int qi) {                               // qi: a valid query of image b
      const long long r = rec_base + (long long)qi * M;
      const float *wrec = attn + r * (L * P) + c;
#pragma unroll
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < PPL; ++k) nw[l][k] = wrec[l * P + k * QL];
      if constexpr (!CARRY) {
        const float *lrec = loc + r * (L * P * 2) + c * 2;
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
          for (int k = 0; k < PPL; ++k) {
            const f32x2 xy = *reinterpret_cast<const f32x2 *>(lrec + (l * P + k * QL) * 2);
            nx[l][k] = xy[0];
            ny[l][k] = xy[1];
          }
        if constexpr (FUSED) {
          const float *rrow = ref + b * ref_batch_stride + (long long)qi * (L * 2);
#pragma unroll
          for (int l = 0; l < L; ++l) nr[l] = *reinterpret_cast<const f32x2 *>(rrow + 2 * l);
        }
      }
    };
    issue_points((okmask & 1u) ? q0 : 0);
    if (tid == 0) {
      unsigned fetched = (unsigned)(item - xcd * chunk + nslots);   // static stride when there is no queue
      if (queue) {
        if (rfetch + 1u >= (unsigned)(item_end - xcd * chunk)) atomicExch(queue + xcd * WIN_QUEUE_STRIDE, 0u);
        fetched = (unsigned)nslots + rfetch;
      }
      next_idx[0] = fetched;
    }
    __syncthreads();                                               // (2) windows staged; next item known
    stamp(4);

    const ST *vlane = vimg + c * VEC;                               // + lane's channel slice (global path)
    const unsigned char *pool_lane = pool + c * 16;
    auto finish_points = [&](const int sn) {                        // sn (uniform): the slot being prepared
      if constexpr (!CARRY) {
        if constexpr (FUSED) {
#pragma unroll
          for (int l = 0; l < L; ++l)
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
              nx[l][k] = fmaf(nx[l][k], invW[l], nr[l][0]);
              ny[l][k] = fmaf(ny[l][k], invH[l], nr[l][1]);
            }
        }
      } else
      [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
        ((sn == Ss ? (void)[&] {
#pragma unroll
          for (int l = 0; l < L; ++l)
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
              nx[l][k] = lxy[Ss][l][k][0];
              ny[l][k] = lxy[Ss][l][k][1];
            }
        }() : (void)0), ...);
      }(std::make_integer_sequence<int, NS>{});
      if constexpr (FUSED) {   // softmax over the record's L*P logits: my points + DPP across the QL lanes
        float mx = -INFINITY;
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
          for (int k = 0; k < PPL; ++k) mx = fmaxf(mx, nw[l][k]);
        mx = fmaxf(mx, dpp_f<0xB1>(mx));
        if constexpr (QL == 4) mx = fmaxf(mx, dpp_f<0x4E>(mx));
        float sum = 0.f;
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
          for (int k = 0; k < PPL; ++k) {
            nw[l][k] = __expf(nw[l][k] - mx);
            sum += nw[l][k];
          }
        sum += dpp_f<0xB1>(sum);
        if constexpr (QL == 4) sum += dpp_f<0x4E>(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
          for (int k = 0; k < PPL; ++k) nw[l][k] *= inv;
      }
    };
    finish_points(0);

#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
      const bool qvalid = (okmask >> s) & 1u;
      const long long rec = rec_base + (long long)(q0 + s * dq) * M;          // used only when qvalid
      const int q_next = ((okmask >> (s + 1)) & 1u) ? q0 + (s + 1) * dq : 0;   // past the last slot: query 0, unused
      float sx[L][PPL], sy[L][PPL], sw[L][PPL];
#pragma unroll
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
          sx[l][k] = nx[l][k];
          sy[l][k] = ny[l][k];
          sw[l][k] = nw[l][k];
        }
      issue_points(q_next);

      f32x2 accp[4][VEC / 2];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < VEC / 2; ++e) accp[a][e] = f32x2{0.f, 0.f};

#pragma unroll
      for (int l = 0; l < L; ++l) {
        const int H = Hs[l], W = Ws[l];
        const bool lds = in_lds[l] != 0;
        const unsigned row_bytes = (unsigned)(wwid[l] * PXB);
        // owner side: geometry of my points on this level (corner offsets + bilinear * attention weights)
        int o1[PPL], o2[PPL], o3[PPL], o4[PPL];
        float g1[PPL], g2[PPL], g3[PPL], g4[PPL];
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
          const float h_im = sy[l][k] * H - 0.5f, w_im = sx[l][k] * W - 0.5f;
          const bool gate = qvalid && h_im > -1 && w_im > -1 && h_im < H && w_im < W;
          const float hf = floorf(h_im), wf = floorf(w_im);
          const int y0 = gate ? (int)hf : 0, x0 = gate ? (int)wf : 0;
          const float lh = gate ? h_im - hf : 0.f, lw = gate ? w_im - wf : 0.f;   // gated-out: weight 0 (cuh:290-296)
          const float wgt = gate ? sw[l][k] : 0.f;
          const float hh = 1.f - lh, hw = 1.f - lw;
          g1[k] = hh * hw * wgt;
          g2[k] = hh * lw * wgt;
          g3[k] = lh * hw * wgt;
          g4[k] = lh * lw * wgt;
          if (lds) {
            // all four corners lie inside the staged window (out-of-map ones are zeros): one offset is enough
            // (a gated-out sample reads the two zero pixels: offset 0, row step 0)
            o1[k] = gate ? (wbase[l] + __mul24(y0 - wy0[l], wwid[l]) + (x0 - wx0[l])) * PXB : 0;
            o2[k] = gate ? (int)row_bytes : 0;
            o3[k] = o4[k] = 0;
          } else {
            // element offsets from this image/head base, out-of-map corners -> -1 (load element 0, select 0)
            const bool top = gate && y0 >= 0, bot = gate && y0 + 1 <= H - 1;
            const bool lft = x0 >= 0, rgt = x0 + 1 <= W - 1;
            const int a = (St[l] + y0 * W + x0) * MD;
            o1[k] = (top && lft) ? a : -1;
            o2[k] = (top && rgt) ? a + MD : -1;
            o3[k] = (bot && lft) ? a + W * MD : -1;
            o4[k] = (bot && rgt) ? a + W * MD + MD : -1;
          }
        }

        // consumer side: every lane walks all P points, taking point p's geometry from its owner lane by DPP
        auto consume = [&](auto pc, auto lds_c) {
          constexpr int p = decltype(pc)::value;
          constexpr bool LDS = decltype(lds_c)::value;
          constexpr int k = p / QL;
          constexpr int ctrl = BcastCtrl<QL, p % QL>::value;
          const int a1 = dpp_i<ctrl>(o1[k]);
          int a2 = 0, a3 = 0, a4 = 0;
          const float w1 = dpp_f<ctrl>(g1[k]), w2 = dpp_f<ctrl>(g2[k]);
          const float w3 = dpp_f<ctrl>(g3[k]), w4 = dpp_f<ctrl>(g4[k]);
          v16 v1, v2, v3, v4;
          if constexpr (LDS) {
            const unsigned char *pa = pool_lane + (unsigned)a1, *pb = pa + (unsigned)dpp_i<ctrl>(o2[k]);
            v1 = *reinterpret_cast<const v16 *>(pa);
            v2 = *reinterpret_cast<const v16 *>(pa + PXB);          // immediate ds_read offset
            v3 = *reinterpret_cast<const v16 *>(pb);
            v4 = *reinterpret_cast<const v16 *>(pb + PXB);
          } else {
            a2 = dpp_i<ctrl>(o2[k]);
            a3 = dpp_i<ctrl>(o3[k]);
            a4 = dpp_i<ctrl>(o4[k]);
            v1 = *reinterpret_cast<const v16 *>(vlane + max(a1, 0));
            v2 = *reinterpret_cast<const v16 *>(vlane + max(a2, 0));
            v3 = *reinterpret_cast<const v16 *>(vlane + max(a3, 0));
            v4 = *reinterpret_cast<const v16 *>(vlane + max(a4, 0));
          }
          // explicit 2-wide FMA chains (v_pk_fma_f32): 4 packed FMAs per channel pair into the accumulator
          const f32x2 ww1 = {w1, w1}, ww2 = {w2, w2}, ww3 = {w3, w3}, ww4 = {w4, w4};
#pragma unroll
          for (int e = 0; e < VEC; e += 2) {
            f32x2 x1 = {Traits<T>::to_acc(v1[e]), Traits<T>::to_acc(v1[e + 1])};
            f32x2 x2 = {Traits<T>::to_acc(v2[e]), Traits<T>::to_acc(v2[e + 1])};
            f32x2 x3 = {Traits<T>::to_acc(v3[e]), Traits<T>::to_acc(v3[e + 1])};
            f32x2 x4 = {Traits<T>::to_acc(v4[e]), Traits<T>::to_acc(v4[e + 1])};
            if constexpr (!LDS) {
              if (a1 < 0) x1 = f32x2{0.f, 0.f};
              if (a2 < 0) x2 = f32x2{0.f, 0.f};
              if (a3 < 0) x3 = f32x2{0.f, 0.f};
              if (a4 < 0) x4 = f32x2{0.f, 0.f};
            }
            // one accumulator per corner: 4 independent FMA chains per channel pair, so consecutive packed FMAs never
            // wait on each other (a single chain stalls the wave on every FMA's latency at 3-4 waves per SIMD)
            accp[0][e / 2] = __builtin_elementwise_fma(ww1, x1, accp[0][e / 2]);
            accp[1][e / 2] = __builtin_elementwise_fma(ww2, x2, accp[1][e / 2]);
            accp[2][e / 2] = __builtin_elementwise_fma(ww3, x3, accp[2][e / 2]);
            accp[3][e / 2] = __builtin_elementwise_fma(ww4, x4, accp[3][e / 2]);
          }
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
      }

      // the next slot's front-end runs BEFORE this slot's store is issued: vmcnt is in order, so waiting for the point
      // loads after the store would also wait for the store to reach memory (that wait sat in the loop latch)
      finish_points(s + 1);
      // (pin the results here: hipcc otherwise sinks half of the arithmetic, and its vmcnt(0), below the store)
#pragma unroll
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < PPL; ++k) asm volatile("" : "+v"(nx[l][k]), "+v"(ny[l][k]), "+v"(nw[l][k]));
      if (qvalid) {
        v16 o;
#pragma unroll
        for (int e = 0; e < VEC / 2; ++e) {
          const f32x2 t = (accp[0][e] + accp[1][e]) + (accp[2][e] + accp[3][e]);
          o[2 * e] = Traits<T>::from_acc(t[0]);
          o[2 * e + 1] = Traits<T>::from_acc(t[1]);
        }
        *reinterpret_cast<v16 *>(out + rec * D + c * VEC) = o;
      }
    }
    stamp(5);
    par ^= 1;
    {                                                               // (written before barrier (2) of this item)
      const unsigned nxt = __builtin_amdgcn_readfirstlane(next_idx[0]);
      item = nxt < (unsigned)(item_end - xcd * chunk) ? xcd * chunk + (int)nxt : item_end;   // never out of the chunk
    }
    if (item < item_end) prepass(item, bb + par * BB_HALF);
    stamp(6);
  }
  if constexpr (STAMP) {
    if (tid == 0 && stamps)
      for (int i = 0; i < 8; ++i) stamps[(size_t)blockIdx.x * 8 + i] = t_sum[i];
  }
}

// One set of 8 per-XCD item counters per launch, from a small per-device ring zeroed once at allocation (the kernel
// leaves every counter at zero again, see its item loop).  Launches on one stream are ordered; 256 eager launches may
// be in flight across streams before a set is reused.  Returns nullptr (static item stride) when the ring cannot be
// allocated -- e.g. the very first launch of the process happening under stream capture -- or with PCT_WIN_QUEUE=0.
unsigned *win_queue_slot(hipStream_t stream)
{
  static const bool enabled = [] { const char *e = getenv("PCT_WIN_QUEUE"); return !(e && e[0] == '0'); }();
  if (!enabled) return nullptr;
  // RING sets cycle through the eager launches; CAPTURE_POOL sets are handed out ONCE each to launches recorded into a
  // HIP graph (a graph bakes the pointer in: its set must never be shared with an eager launch that could run at the
  // same time on another stream -- two launches on one set would skip / repeat items and leave the counters non-zero).
  // When the pool is exhausted a captured launch falls back to the static item stride.
  constexpr int MAX_DEV = 64, RING = 256, CAPTURE_POOL = 4096;
  static std::mutex mu;
  static unsigned *ring[MAX_DEV] = {};
  static unsigned seq[MAX_DEV] = {};
  static unsigned cap_used[MAX_DEV] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  const bool capturing = cap != hipStreamCaptureStatusNone;
  std::lock_guard<std::mutex> lock(mu);
  if (!ring[dev]) {
    // never allocate under stream capture (an allocation would invalidate the capture): static stride for this launch
    if (capturing) return nullptr;
    void *p = nullptr;
    const size_t bytes = (size_t)(RING + CAPTURE_POOL) * 8 * WIN_QUEUE_STRIDE * sizeof(unsigned);
    // (the memset runs on the null stream; launches may come from non-blocking streams: wait for it once)
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError();
      if (p) (void)hipFree(p);
      return nullptr;
    }
    ring[dev] = static_cast<unsigned *>(p);
  }
  if (capturing) {
    if (cap_used[dev] >= (unsigned)CAPTURE_POOL) return nullptr;
    return ring[dev] + 8 * WIN_QUEUE_STRIDE * (size_t)(RING + cap_used[dev]++);
  }
  return ring[dev] + (size_t)8 * WIN_QUEUE_STRIDE * (seq[dev]++ % RING);
}

int prepare_win_queue_device()
{
  hipStream_t null_stream = nullptr;
  return win_queue_slot(null_stream) ? 0 : 0;      // (nullptr = queues disabled or no memory: launches then use the static stride)
}

static unsigned long long *g_stamp_buffer = nullptr;
void set_win_stamp_buffer(void *p) { g_stamp_buffer = static_cast<unsigned long long *>(p); }
unsigned long long *win_stamp_buffer() { return g_stamp_buffer; }

// ---- launcher: returns -100 when this geometry is not covered (caller uses another kernel) ----------------------
// ref == nullptr: plain op; ref != nullptr: fused front-end (loc = raw offsets, attn = raw logits).
template <typename T>
int launch_msda_forward_win(const void *value, const int64_t *shapes, const int64_t *starts, const void *loc,
                            const void *attn, int N, int S, int M, int D, int L, int Lq, int P, void *out,
                            hipStream_t stream, const float *ref, long long ref_batch_stride)
{
  using ST = typename Traits<T>::store_t;
  constexpr int VEC = 16 / (int)sizeof(ST);
  if ((((uintptr_t)value | (uintptr_t)out) & 15u) || (((uintptr_t)loc) & 7u) || (((uintptr_t)attn) & 3u)) return -100;
  if (ref && (((uintptr_t)ref) & 7u)) return -100;
  if (D != 16 || (P != 4 && P != 8) || L < 3 || L > 5) return -100;
  if ((long long)N * Lq * M < 32768) return -100;                  // too small to fill a persistent grid
  if ((long long)N * ((long long)S + 64 * L) * M >= 0x7fffffffLL) return -100;   // item counter is 32-bit
  constexpr int QL = 16 / VEC;
  static const int ns_env = [] { const char *e = getenv("PCT_WIN_NS"); return e ? atoi(e) : 4; }();
  // P = 8 (config 5): twice the points per lane -> 2 slots per lane, but still the 50 KB pool / 3 workgroups per CU
  const int NS = P == 8 ? (sizeof(ST) == 2 ? 1 : 2) : ((ns_env == 1 || ns_env == 2) ? ns_env : 4);   // 16-bit: 128 queries per slot
  const int pool_bytes = P == 8 ? 50 * 1024 : (NS == 1 ? 28 * 1024 : (NS == 2 ? 36 * 1024 : 50 * 1024));
  const int pool_px = pool_bytes / (QL * 16);
  const size_t lds = (size_t)pool_px * QL * 16 + 2 * (WIN_BLOCK / 64) * WIN_MAXL * 2 * sizeof(unsigned) + 16;
  const int pyramid = Lq == S ? 1 : 0;
  const int wg_per_cu = (NS == 4 || P == 8) ? 3 : 4;
  const dim3 grid(256 * wg_per_cu), block(WIN_BLOCK);
  unsigned *queue = win_queue_slot(stream);                        // nullptr: static item stride
  const ST *v = static_cast<const ST *>(value);
  const float *lc = static_cast<const float *>(loc), *at = static_cast<const float *>(attn);
  ST *o = static_cast<ST *>(out);
  if constexpr (sizeof(ST) == 4) {
    if (g_stamp_buffer && !ref && L == 4 && NS == 4) {   // diagnostic build: per-phase cycle stamps (tools/stamp_msda.py)
      hipLaunchKernelGGL((msda_forward_win_kernel<T, 16, 4, 4, 4, false, true>), grid, block, lds, stream, v, shapes,
                         starts, lc, at, N, S, M, Lq, pyramid, pool_px, o, ref, ref_batch_stride, queue, g_stamp_buffer);
      return (int)hipGetLastError();
    }
  }
#define PCT_WINP(L_, P_, NS_, FU_)                                                                                    \
  hipLaunchKernelGGL((msda_forward_win_kernel<T, 16, L_, P_, NS_, FU_>), grid, block, lds, stream, v, shapes, starts, \
                     lc, at, N, S, M, Lq, pyramid, pool_px, o, ref, ref_batch_stride, queue, nullptr)
#define PCT_WIN(L_, NS_, FU_) PCT_WINP(L_, 4, NS_, FU_)
  if (P == 8) {
    constexpr int NS8 = sizeof(ST) == 2 ? 1 : 2;
    if (ref) {
      if (L == 3) PCT_WINP(3, 8, NS8, true);
      else if (L == 4) PCT_WINP(4, 8, NS8, true);
      else PCT_WINP(5, 8, NS8, true);
    } else {
      if (L == 3) PCT_WINP(3, 8, NS8, false);
      else if (L == 4) PCT_WINP(4, 8, NS8, false);
      else PCT_WINP(5, 8, NS8, false);
    }
    return (int)hipGetLastError();
  }
#define PCT_WIN_L(NS_, FU_)                 \
  if (L == 3) PCT_WIN(3, NS_, FU_);         \
  else if (L == 4) PCT_WIN(4, NS_, FU_);    \
  else PCT_WIN(5, NS_, FU_)
  if (ref) {
    if (NS == 4) { PCT_WIN_L(4, true); }
    else if (NS == 2) { PCT_WIN_L(2, true); }
    else { PCT_WIN_L(1, true); }
  } else {
    if (NS == 4) { PCT_WIN_L(4, false); }
    else if (NS == 2) { PCT_WIN_L(2, false); }
    else { PCT_WIN_L(1, false); }
  }
#undef PCT_WIN_L
#undef PCT_WIN
#undef PCT_WINP
  return (int)hipGetLastError();
}

template int launch_msda_forward_win<float>(const void *, const int64_t *, const int64_t *, const void *,
                                            const void *, int, int, int, int, int, int, int, void *, hipStream_t,
                                            const float *, long long);
template int launch_msda_forward_win<half_bits>(const void *, const int64_t *, const int64_t *, const void *,
                                                const void *, int, int, int, int, int, int, int, void *, hipStream_t,
                                                const float *, long long);
template int launch_msda_forward_win<bf16_bits>(const void *, const int64_t *, const int64_t *, const void *,
                                                const void *, int, int, int, int, int, int, int, void *, hipStream_t,
                                                const float *, long long);

}  // namespace pct
