// The encoder layer's whole feed-forward block in ONE kernel on MI355X (gfx950):
//     y = LayerNorm(x + W2 . relu(W1 . x + b1) + b2),        x, y [rows, 128] fp32, W1 [F, 128], W2 [128, F], F % 32 == 0
// (pixel_decoder/msdeformattn.py:122-131 of the reference: linear1, activation, dropout2, linear2, dropout3, norm2; F = 1024.)
// As two kernels (linear_k128_split.hip + linear_ln_split.hip) the [rows, F] hidden tensor -- 11.4 GB at the bench's 2.8 M rows --
// is written by one and read back by the other: 21 + 25 ms of the 102 ms step, both kernels' matrix pipes 47-62 % busy, half
// of their time memory.  Here the hidden activations never leave the registers.
//
// Same arithmetic as those kernels: every fp32 operand is the exact sum of three bf16 pieces, a product is evaluated from its six
// leading partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, the a1w1 chain and the five small terms in separate
// accumulators; the hidden value is the fp32 number (hi + lo) + b1, ReLU'd, split again exactly.
//
// Structure.  Workgroup = 4 waves = 128 rows, ONE workgroup per CU with the whole register file (512 registers per lane) and
// 112 KB of LDS.  A wave owns 32 rows for both products, so nothing but weights is shared:
//   * its rows of x are split once per tile into the B fragments of GEMM 1 (8 k-steps x 3 planes, 96 registers);
//   * the hidden dimension is walked in chunks of 32 units: GEMM 1 gives D1[hidden 32][rows 32] -- W1 pieces as the A operand,
//     the transposed product as everywhere in this library -- whose accumulator layout holds, per lane (row r, half h), hidden
//     units 8 q + 4 h + t: after bias, ReLU and split these registers ARE the B fragments of GEMM 2, y[cols][rows] += W2 . h,
//     under a permutation of the k slots that the W2 image in LDS is written in (slot (h, e) of k-step s <-> hidden
//     16 s + 4 h + e for e < 4, 16 s + 8 + 4 h + e - 4 else): no LDS round trip, no shuffles;
//   * both weight matrices are pre-split per call into the exact LDS image of every chunk (W1: 32 rows x 128 k, row stride
//     272 B; W2: 128 rows x 32 permuted k, row stride 80 B; both conflict-free for ds_read_b128), 56 KB per chunk, streamed from
//     L2 through registers into a double-buffered stage, one barrier per chunk;
//   * the row statistics of the LayerNorm are complete inside a wave (a lane and its partner in the other half hold a row's 128
//     columns).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "msda_win_common.hpp"    // func_attr_per_device

// knock-outs (WRONG RESULTS, timing only): 1 = no image transport inside the chunk loop, 2 = no bias / ReLU / split arithmetic,
// 4 = no barrier per chunk, 16 / 32 = of the transport only the LDS stores / only the loads, 8 = no residual loads, 64 = no y stores
#ifndef PCT_FFN_KO
#define PCT_FFN_KO 0
#endif
#if PCT_FFN_KO && !defined(PCT_EXPERIMENT_BUILD)
#error "PCT_FFN_KO gives wrong results: add -DPCT_EXPERIMENT_BUILD"
#endif

// PCT_FFN_STAMPS (experiment builds): s_memtime at the phase boundaries of every chunk, sums per workgroup in pct_ffn_stamps
#ifndef PCT_FFN_SPREAD
#define PCT_FFN_SPREAD 1      /* fragment reads spread between the MFMAs of a group instead of issued together at its start */
#endif
#ifndef PCT_FFN_STORE_NT
#define PCT_FFN_STORE_NT 0      /* y rows with the non-temporal hint: 6.79 -> 6.97 ms, off */
#endif
#ifndef PCT_FFN_STAMPS
#define PCT_FFN_STAMPS 0
#endif
#if PCT_FFN_STAMPS && !defined(PCT_EXPERIMENT_BUILD)
#error "PCT_FFN_STAMPS perturbs the kernel: add -DPCT_EXPERIMENT_BUILD"
#endif

namespace pct {

#if PCT_FFN_STAMPS
__device__ unsigned long long ffn_stamps[256 * 8];
__device__ __forceinline__ unsigned long long ff_now()
{
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define FF_STAMP(slot)                                   \
  do {                                                   \
    const unsigned long long now_ = ff_now();            \
    acc_t[slot] += now_ - last_t;                        \
    last_t = now_;                                       \
  } while (0)
#else
#define FF_STAMP(slot)
#endif

typedef float ff_f32x16 __attribute__((ext_vector_type(16)));
typedef float ff_f32x4 __attribute__((ext_vector_type(4)));
typedef float ff_f32x2 __attribute__((ext_vector_type(2)));
typedef int ff_i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 ff_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ff_bf16x2 __attribute__((ext_vector_type(2)));

constexpr int FF_BLOCK = 256;
constexpr int FF_BM = 128;                        // rows per workgroup tile
constexpr int FF_D = 128;                         // model width
constexpr int FF_CH = 32;                         // hidden units per chunk
constexpr int FF_W1ROW = 272;                     // bytes per W1 row of a plane (128 bf16 + 16 pad)
constexpr int FF_W1PLANE = FF_CH * FF_W1ROW;      // 8 704
constexpr int FF_W2ROW = 80;                      // bytes per W2 row of a plane (32 bf16 + 16 pad)
constexpr int FF_W2PLANE = FF_D * FF_W2ROW;       // 10 240
constexpr int FF_W2OFF = 3 * FF_W1PLANE;          // 26 112
constexpr int FF_STAGE = 57344;                   // 26 112 + 30 720 = 56 832, padded to 256 threads x 14 x 16 B
constexpr int FF_PIECES = FF_STAGE / (FF_BLOCK * 16);   // 14
constexpr int FF_HALF = FF_PIECES / 2;                  // 7
constexpr int FF_LDS = 2 * FF_STAGE + 3 * FF_D * 4;     // two stages, then b2, gamma, beta
constexpr int FF_B1OFF = 56832;                         // the chunk's 32 biases, in the padding

__device__ __forceinline__ void ff_split(const float x, const float y, unsigned &p1, unsigned &p2, unsigned &p3)
{
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(ff_f32x2{x, y}, ff_bf16x2));
  const float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);   // exact
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(ff_f32x2{rx, ry}, ff_bf16x2));
  const float sx = rx - __uint_as_float(p2 << 16), sy = ry - __uint_as_float(p2 & 0xffff0000u); // exact
  p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(ff_f32x2{sx, sy}, ff_bf16x2));
}
// eight consecutive floats -> three bf16x8 pieces
__device__ __forceinline__ void ff_split8(const ff_f32x4 a, const ff_f32x4 b, ff_bf16x8 (&out)[3])
{
  unsigned p[3][4];
  ff_split(a[0], a[1], p[0][0], p[1][0], p[2][0]);
  ff_split(a[2], a[3], p[0][1], p[1][1], p[2][1]);
  ff_split(b[0], b[1], p[0][2], p[1][2], p[2][2]);
  ff_split(b[2], b[3], p[0][3], p[1][3], p[2][3]);
#pragma unroll
  for (int q = 0; q < 3; ++q) out[q] = __builtin_bit_cast(ff_bf16x8, ff_i32x4{(int)p[q][0], (int)p[q][1], (int)p[q][2], (int)p[q][3]});
}

// W1 [F][128], W2 [128][F], b1 [F] fp32 -> img [F / 32][FF_STAGE bytes]: image c holds W2 and b1 of chunk c and W1 of chunk c + 1
// (mod F / 32): what one iteration of the kernel's chunk loop reads (see the header); one thread per element pair
__global__ __launch_bounds__(256) void ffn_split_weights_kernel(const float *__restrict__ w1, const float *__restrict__ w2,
                                                                const float *__restrict__ b1, const int F,
                                                                unsigned short *__restrict__ img)
{
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  const long long n1 = (long long)F * FF_D;
  const int nch = F / FF_CH;
  unsigned p1, p2, p3;
  if (e < n1) {                                                   // W1[u][k], k even: row u % 32, k-contiguous
    const int u = (int)(e / FF_D), k = (int)(e - (long long)u * FF_D);
    ff_split(w1[e], w1[e + 1], p1, p2, p3);
    const int c = (u / FF_CH + nch - 1) % nch;
    unsigned short *d = img + (size_t)c * (FF_STAGE / 2) + (size_t)(u % FF_CH) * (FF_W1ROW / 2) + k;
    *reinterpret_cast<unsigned *>(d) = p1;
    *reinterpret_cast<unsigned *>(d + FF_W1PLANE / 2) = p2;
    *reinterpret_cast<unsigned *>(d + FF_W1PLANE) = p3;
  } else if (e < 2 * n1) {                                        // W2[c][u], u even: chunk u / 32, row c, slot of u in the chunk
    const long long e2 = e - n1;
    const int c = (int)(e2 / F), u = (int)(e2 - (long long)c * F);
    ff_split(w2[e2], w2[e2 + 1], p1, p2, p3);
    const int uu = u % FF_CH;                                     // 0 .. 31 (even): s = uu / 16, inside: v = uu % 16
    const int s = uu >> 4, v = uu & 15;
    // hidden 16 s + v with v = 8 g + 4 h + t (g = 0, 1; h = 0, 1; t = 0 .. 3) sits in slot (h, e = 4 g + t) of k-step s:
    // bf16 index inside the row = (2 s + h) * 8 + 4 g + t
    const int g = v >> 3, h = (v >> 2) & 1, t = v & 3;
    unsigned short *d = img + (size_t)(u / FF_CH) * (FF_STAGE / 2) + (size_t)(FF_W2OFF / 2) + (size_t)c * (FF_W2ROW / 2) +
                        (2 * s + h) * 8 + 4 * g + t;
    *reinterpret_cast<unsigned *>(d) = p1;                        // (t even: t and t + 1 are adjacent slots)
    *reinterpret_cast<unsigned *>(d + FF_W2PLANE / 2) = p2;
    *reinterpret_cast<unsigned *>(d + FF_W2PLANE) = p3;
  } else if (e < 2 * n1 + F) {                                    // b1[u], u even
    const int u = (int)(e - 2 * n1);
    float *d = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(img) + (size_t)(u / FF_CH) * FF_STAGE + FF_B1OFF) + u % FF_CH;
    d[0] = b1[u];
    d[1] = b1[u + 1];
  }
}

__global__ __launch_bounds__(FF_BLOCK, 1) void ffn_fused_split_kernel(
    const float *__restrict__ X, const long long ldx, const unsigned short *__restrict__ img, const int F,
    const float *__restrict__ b2, const float *__restrict__ gamma, const float *__restrict__ beta,
    const float eps, const long long M, float *__restrict__ Y, const long long ldy)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char ff_smem[];    // two stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const long long ntiles = (M + FF_BM - 1) / FF_BM;
  const int nch = F / FF_CH;
#if PCT_FFN_STAMPS
  unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_t = ff_now();
#endif

  const auto img_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(img), 0, (int)((long long)nch * FF_STAGE),
                                                          0x00020000);
  // An image travels to LDS through registers, 14 pieces of 16 bytes per thread, ONE load and one store per MFMA group of six: the
  // CU's vector-memory path takes a 1 KiB wave-instruction per ~16 cycles, so the four waves' loads issued together in bursts of
  // seven backed up into the issuing waves (stamps: ~50 cycles per load, 700 per chunk).  Window position p = 0 .. 13 is piece
  // ORD[p] (the W2 / bias part first: see the tile seam below), stored six groups after it was requested, slot p % 7.
  ff_i32x4 gst[7];
  bool transport = true;
  auto t_piece = [](const int p) { return p < 7 ? p + 7 : p - 7; };
  auto t_load = [&](const int p, const int image) {
    if ((PCT_FFN_KO & (1 | 16)) && !transport) return;
    gst[p % 7] = __builtin_amdgcn_raw_buffer_load_b128(img_rsrc, tid * 16, image * FF_STAGE + t_piece(p) * (FF_BLOCK * 16), 0);
  };
  auto t_store = [&](const int p, unsigned char *stage) {
    if ((PCT_FFN_KO & 1) && !transport) return;
    if ((PCT_FFN_KO & 32) && !transport) {
      asm volatile("" ::"v"(gst[p % 7]));
      return;
    }
    *reinterpret_cast<ff_i32x4 *>(stage + t_piece(p) * (FF_BLOCK * 16) + tid * 16) = gst[p % 7];
  };

  // fragment addresses inside a stage: W1 (A operand of GEMM 1): hidden row r, k = 16 s + 8 h ..; W2 (A operand of GEMM 2):
  // output column 32 cb + r, slots of k-step s and half h
  const int w1_off = r * FF_W1ROW + 16 * h;                       // + 32 s, + plane
  const int w2_off = FF_W2OFF + r * FF_W2ROW + 16 * h;            // + cb * 32 rows, + 32 s, + plane

  // stage 1 <- the last image (W1 of chunk 0), stage 0 <- image 0; then what the groups before "iteration 0" would have done for
  // image 1 (into stage 1: its W2 part only, the W1 part there is still needed): positions 0 and 1 stored, 2 .. 7 requested
#pragma unroll
  for (int p0 = 0; p0 < 14; p0 += 7) {
#pragma unroll
    for (int p = p0; p < p0 + 7; ++p) t_load(p, nch - 1);
#pragma unroll
    for (int p = p0; p < p0 + 7; ++p) t_store(p, ff_smem + FF_STAGE);
#pragma unroll
    for (int p = p0; p < p0 + 7; ++p) t_load(p, 0);
#pragma unroll
    for (int p = p0; p < p0 + 7; ++p) t_store(p, ff_smem);
  }
  if (tid < FF_D) {                                               // b2, gamma, beta behind the stages
    float *par = reinterpret_cast<float *>(ff_smem + 2 * FF_STAGE);
    par[tid] = b2 ? b2[tid] : 0.f;
    par[FF_D + tid] = gamma[tid];
    par[2 * FF_D + tid] = beta[tid];
  }
  t_load(0, 1 % nch);
  t_load(1, 1 % nch);
  t_store(0, ff_smem + FF_STAGE);
  t_store(1, ff_smem + FF_STAGE);
#pragma unroll
  for (int p = 2; p < 8; ++p) t_load(p, 1 % nch);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  int cur = 0;                                                    // stage holding the image of the chunk about to be used
  transport = false;
  FF_STAMP(0);

  // a tile's rows of x are requested one tile ahead (under GEMM 2 of the last chunk of the tile before, when the fragments of the
  // current rows are no longer needed; a row-strided load touches 32 lines: 16 of them back to back cost 3 k cycles): lane (r, h) takes k = 16 s + 8 h .. + 7 of row r, s = 0 .. 7
  ff_i32x4 xraw[16];
  auto x_request = [&](const long long tile, const int s0, const int s1) {      // k-steps s0 .. s1 - 1
    const long long row0 = tile * FF_BM + 32 * wave;
    const long long left = M - row0;
    const unsigned xbytes = (unsigned)((left < 32 ? (left < 0 ? 0 : left) : 32) * ldx * 4);      // past the end: zeros
    const auto xq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X + (left > 0 ? row0 : 0) * ldx), 0, (int)xbytes, 0x00020000);
    const int xoff = (int)((r * ldx + 8 * h) * 4);
#pragma unroll
    for (int s = s0; s < s1; ++s) {
      xraw[2 * s] = __builtin_amdgcn_raw_buffer_load_b128(xq, xoff, s * 64, 0);
      xraw[2 * s + 1] = __builtin_amdgcn_raw_buffer_load_b128(xq, xoff, s * 64 + 16, 0);
    }
  };
  x_request(blockIdx.x, 0, 8);

  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // ---- this wave's 32 rows of x as GEMM 1's B fragments ---------------------------------------------------------------------
    const long long row0 = tile * FF_BM + 32 * wave;
    const long long left = M - row0;
    const unsigned xbytes = (unsigned)((left < 32 ? (left < 0 ? 0 : left) : 32) * ldx * 4);
    const auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X + row0 * ldx), 0, (int)xbytes, 0x00020000);
    ff_bf16x8 xp[8][3];
#pragma unroll
    for (int s = 0; s < 8; ++s) ff_split8(__builtin_bit_cast(ff_f32x4, xraw[2 * s]), __builtin_bit_cast(ff_f32x4, xraw[2 * s + 1]), xp[s]);
    FF_STAMP(1);
    ff_f32x16 y_hi[4], y_lo[4];                                   // D2[col block][cols 32][rows 32]
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int j = 0; j < 16; ++j) y_hi[cb][j] = y_lo[cb][j] = 0.f;

    // GEMM 1: D1[hidden 8 q + 4 h + t][row r] over k = 128 from the W1 part of stage `st`.  One wave per SIMD: nobody else covers an
    // LDS round trip, so the fragments of k-step s + 1 are requested before the MFMAs of step s
    ff_f32x16 h_hi, h_lo;
    ff_bf16x8 wa[2][3];
    auto w1_frags = [&](const unsigned char *st, const int s) {
#pragma unroll
      for (int p = 0; p < 3; ++p) wa[s & 1][p] = *reinterpret_cast<const ff_bf16x8 *>(st + w1_off + 32 * s + p * FF_W1PLANE);
    };
    auto gemm1_step = [&](const int s) {
      const ff_bf16x8 a1 = wa[s & 1][0], a2 = wa[s & 1][1], a3 = wa[s & 1][2];
      h_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xp[s][2], h_lo, 0, 0, 0);
      h_hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xp[s][0], h_hi, 0, 0, 0);
      h_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, xp[s][0], h_lo, 0, 0, 0);
      h_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xp[s][1], h_lo, 0, 0, 0);
      h_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xp[s][1], h_lo, 0, 0, 0);
      h_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xp[s][0], h_lo, 0, 0, 0);
    };
    // chunk 0's first product stands alone (its W1 is in the image the previous tile ended on); every later one runs under the
    // bias / ReLU / split arithmetic of the chunk before it
    {
      const unsigned char *pst = ff_smem + (cur ^ 1) * FF_STAGE;
#pragma unroll
      for (int j = 0; j < 16; ++j) h_hi[j] = h_lo[j] = 0.f;
      w1_frags(pst, 0);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s < 7) w1_frags(pst, s + 1);
        gemm1_step(s);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);        // the reads first (left alone the scheduler sinks them)
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_barrier();                               // iteration 0 refills that stage
    }

    // entering iteration `chunk`: wa[0] = step 0 of the W1 fragments and bq = the biases it needs, gst = the first half of image
    // chunk + 1 on its way
    ff_f32x4 bq[4];
    auto load_bias = [&](const unsigned char *st) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const ff_f32x4 *>(st + FF_B1OFF + (8 * q + 4 * h) * 4);
    };
    w1_frags(ff_smem + cur * FF_STAGE, 0);
    load_bias(ff_smem + cur * FF_STAGE);
    FF_STAMP(2);

    auto chunk_body = [&](auto last_tag, const int chunk) {
      constexpr bool LAST = decltype(last_tag)::value;
      const unsigned char *st = ff_smem + cur * FF_STAGE;         // image `chunk`: W2 and b1 of this chunk, W1 of the next
      unsigned char *nst = ff_smem + (cur ^ 1) * FF_STAGE;        // nobody reads it since the previous barrier: image chunk + 1
      const int nxt = LAST ? 0 : chunk + 1, nxt2 = (chunk + 2) % nch;
      // this chunk's hidden accumulators move to the vector side; the same registers then take the next chunk's product
      const ff_f32x16 c_hi = h_hi, c_lo = h_lo;
      if constexpr (!LAST) {
#pragma unroll
        for (int j = 0; j < 16; ++j) h_hi[j] = h_lo[j] = 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase A: GEMM 1 of chunk + 1 under bias, ReLU, split of this chunk: value pair m of the B fragments of GEMM 2 is k-step
      // sg = m / 4, word w = m % 4 (accumulator quartet q = 2 sg + w / 2, elements 2 (w % 2) and + 1); pairs 0 - 3 (k-step 0) behind
      // the odd steps here, pairs 4 - 7 (k-step 1, first used by group 4) behind groups 0 - 3 of phase B; image chunk + 1: positions
      // 2 .. 9 stored, 8 .. 13 requested ------------------------------------------------------------------------------------------
      unsigned hpw[2][3][4];
      ff_bf16x8 wb[2][3], wc[3];
      auto valu_slice = [&](const int m) {
        const int sg = m >> 2, w = m & 3, q = 2 * sg + (w >> 1), t0 = 2 * (w & 1), j0 = 4 * q + t0;
        const float v0 = fmaxf((c_hi[j0] + c_lo[j0]) + bq[q][t0], 0.f);
        const float v1 = fmaxf((c_hi[j0 + 1] + c_lo[j0 + 1]) + bq[q][t0 + 1], 0.f);
        if (PCT_FFN_KO & 2) {
          hpw[sg][0][w] = __float_as_uint(c_hi[j0]);
          hpw[sg][1][w] = __float_as_uint(c_lo[j0]);
          hpw[sg][2][w] = __float_as_uint(c_hi[j0 + 1]) ^ __float_as_uint(c_lo[j0 + 1]);
        } else
          ff_split(v0, v1, hpw[sg][0][w], hpw[sg][1][w], hpw[sg][2][w]);
      };
      auto h_frag = [&](const int sg, const int p) {
        return __builtin_bit_cast(ff_bf16x8, ff_i32x4{(int)hpw[sg][p][0], (int)hpw[sg][p][1], (int)hpw[sg][p][2], (int)hpw[sg][p][3]});
      };
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if constexpr (!LAST) {
          if (s < 7) w1_frags(st, s + 1);
          gemm1_step(s);
        }
        if (s == 7) {
#pragma unroll
          for (int p = 0; p < 3; ++p) wb[0][p] = *reinterpret_cast<const ff_bf16x8 *>(st + w2_off + p * FF_W2PLANE);
        }
        if (s & 1) valu_slice(s >> 1);
        t_store(s + 2, nst);
        if (s <= 5) t_load(s + 8, nxt);
        if constexpr (!LAST) {                                    // reads first; vector instructions, the load and the store between MFMAs
          if (!PCT_FFN_SPREAD) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (s & 1) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            if (PCT_FFN_SPREAD && !(i & 1)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i == 1 && s <= 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (i == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      FF_STAMP(3);
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase B: GEMM 2: D2[col 32 cb + 8 q + 4 h + t][row r] += W2 . h over this chunk's 32 hidden units; group g = (s, cb),
      // the next group's fragments requested before this group's MFMAs.  Image chunk + 1 is complete behind group 3 and the last two
      // groups' fragments are in registers before group 6, so the chunk's barrier sits after group 5: groups 6 and 7 cover it, the
      // first reads of the next iteration and the first two stores of image chunk + 2 into the stage just read (its W2 part: at a
      // tile seam the W1 part is still needed by the next tile's first product); image chunk + 2 is requested from group 0 on -------
      auto w2_frags = [&](ff_bf16x8 (&dst)[3], const int g) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          dst[p] = *reinterpret_cast<const ff_bf16x8 *>(st + w2_off + (g & 3) * 32 * FF_W2ROW + 32 * (g >> 2) + p * FF_W2PLANE);
      };
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int sg = g >> 2, cb = g & 3;
        if (g < 5) w2_frags(wb[(g + 1) & 1], g + 1);
        if (g == 5) {
          w2_frags(wb[0], 6);
          w2_frags(wc, 7);
        }
        if (g < 4) valu_slice(4 + g);
        t_load(g, nxt2);
        if constexpr (LAST) x_request(tile + gridDim.x, g, g + 1);     // the next tile's rows, two row-strided loads per group
        if (g < 4) t_store(10 + g, nst);
        if (g >= 6) t_store(g - 6, const_cast<unsigned char *>(st));   // behind the barrier: this stage is free, image chunk + 2
        if (g == 6) {
          if constexpr (!LAST) {
            w1_frags(nst, 0);
            load_bias(nst);
          }
        }
        const ff_bf16x8 a1 = g == 7 ? wc[0] : wb[g & 1][0], a2 = g == 7 ? wc[1] : wb[g & 1][1], a3 = g == 7 ? wc[2] : wb[g & 1][2];
        y_lo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, h_frag(sg, 2), y_lo[cb], 0, 0, 0);
        y_hi[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, h_frag(sg, 0), y_hi[cb], 0, 0, 0);
        y_lo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, h_frag(sg, 0), y_lo[cb], 0, 0, 0);
        y_lo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, h_frag(sg, 1), y_lo[cb], 0, 0, 0);
        y_lo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, h_frag(sg, 1), y_lo[cb], 0, 0, 0);
        y_lo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, h_frag(sg, 0), y_lo[cb], 0, 0, 0);
        if (g < 5 && !PCT_FFN_SPREAD) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        if (g == 5 && !PCT_FFN_SPREAD) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        if (g == 6 && !LAST && !PCT_FFN_SPREAD) __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (g < 4) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          if (PCT_FFN_SPREAD) {                                   // fragment reads one (two, group 5 and 6) per MFMA, not in a burst
            if (g < 5 && !(i & 1)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (g == 5) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (g == 6 && !LAST) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (g == 6 && !LAST && i == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          if (i == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          if (i == 3 && (g < 4 || g >= 6)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g == 5 && !(PCT_FFN_KO & 4)) {
          FF_STAMP(4);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          FF_STAMP(5);
        }
      }
      FF_STAMP(6);
      cur ^= 1;
    };
    for (int chunk = 0; chunk + 1 < nch; ++chunk) chunk_body(std::false_type{}, chunk);
    chunk_body(std::true_type{}, nch - 1);

    // ---- epilogue: y_*[cb][4 q + t] = y[row r][column 32 cb + 8 q + 4 h + t]; + b2 + residual, LayerNorm, store -----------------
    const unsigned ybytes = (unsigned)((left < 32 ? (left < 0 ? 0 : left) : 32) * ldy * 4);
    const auto yr = __builtin_amdgcn_make_buffer_rsrc(Y + row0 * ldy, 0, (int)ybytes, 0x00020000);
    float v[64];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * cb + 8 * q + 4 * h;
        const ff_f32x4 res = (PCT_FFN_KO & 8) ? ff_f32x4{0.f, 0.f, 0.f, 0.f}
                                              : __builtin_bit_cast(ff_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)((r * ldx + c0) * 4), 0, 0));
        const ff_f32x4 bb = *reinterpret_cast<const ff_f32x4 *>(ff_smem + 2 * FF_STAGE + c0 * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) v[16 * cb + 4 * q + t] = ((y_hi[cb][4 * q + t] + y_lo[cb][4 * q + t]) + bb[t]) + res[t];
      }
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) sm += v[j];
    sm += __shfl_xor(sm, 32);
    const float mean = sm * (1.f / 128.f);
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      v[j] -= mean;
      s2 += v[j] * v[j];
    }
    s2 += __shfl_xor(s2, 32);
    const float rstd = rsqrtf(s2 * (1.f / 128.f) + eps);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * cb + 8 * q + 4 * h;
        const ff_f32x4 gq = *reinterpret_cast<const ff_f32x4 *>(ff_smem + 2 * FF_STAGE + (FF_D + c0) * 4);
        const ff_f32x4 eq = *reinterpret_cast<const ff_f32x4 *>(ff_smem + 2 * FF_STAGE + (2 * FF_D + c0) * 4);
        ff_f32x4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = v[16 * cb + 4 * q + t] * rstd * gq[t] + eq[t];
        if (PCT_FFN_KO & 64)
          asm volatile("" ::"v"(o));
        else
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ff_i32x4, o), yr, (int)((r * ldy + c0) * 4), 0, PCT_FFN_STORE_NT ? 2 : 0);
      }
    FF_STAMP(7);
  }
#if PCT_FFN_STAMPS
  if (tid == 0 && blockIdx.x < 256)
    for (int i = 0; i < 8; ++i) ffn_stamps[blockIdx.x * 8 + i] = acc_t[i];
#endif
}

#define PCT_FF_STR2(x) #x
#define PCT_FF_STR(x) PCT_FF_STR2(x)
const char *ffn_fused_build_flags() { return "ffn: KO=" PCT_FF_STR(PCT_FFN_KO) " STAMPS=" PCT_FF_STR(PCT_FFN_STAMPS) " SPREAD=" PCT_FF_STR(PCT_FFN_SPREAD); }

// the kernel's LDS attribute for the current device, outside any stream capture (pct_prepare_device)
int prepare_ffn_device() { return func_attr_per_device(reinterpret_cast<const void *>(&ffn_fused_split_kernel), FF_LDS) == hipSuccess ? 0 : -4; }

// img_ws: (F / 32) * 57 344 bytes, 16-byte aligned, refilled on every call.  -4: geometry not covered.
int launch_ffn_fused_split(const float *x, long long ldx, const float *w1, const float *b1, const float *w2, const float *b2,
                           const float *gamma, const float *beta, float eps, int F, long long rows, void *img_ws, float *out,
                           long long ldo, hipStream_t stream)
{
  if (rows <= 0) return 0;
  if (F <= 0 || F % FF_CH) return -4;
  if ((long long)(F / FF_CH) * FF_STAGE > 0x7fffffffLL) return -4;
  if (32LL * (ldx > ldo ? ldx : ldo) * 4 > 0x7fffffffLL) return -4;
  if (func_attr_per_device(reinterpret_cast<const void *>(&ffn_fused_split_kernel), FF_LDS) != hipSuccess) return -4;
  const long long pairs = (long long)F * FF_D + F / 2;            // element pairs of both matrices and the bias
  hipLaunchKernelGGL(ffn_split_weights_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, w1, w2, b1, F,
                     static_cast<unsigned short *>(img_ws));
  const long long ntiles = (rows + FF_BM - 1) / FF_BM;
  const unsigned gx = (unsigned)(ntiles < 256 ? ntiles : 256);    // persistent: one workgroup per CU
  hipLaunchKernelGGL(ffn_fused_split_kernel, dim3(gx), dim3(FF_BLOCK), FF_LDS, stream, x, ldx,
                     static_cast<const unsigned short *>(img_ws), F, b2, gamma, beta, eps, rows, out, ldo);
  return (int)hipGetLastError();
}

#if PCT_FFN_STAMPS
extern "C" __attribute__((visibility("default"))) int pct_ffn_stamps_read(unsigned long long *host_out)
{
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ffn_stamps), sizeof(unsigned long long) * 256 * 8);
}
#endif

}  // namespace pct
