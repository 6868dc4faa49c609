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

namespace pct {

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

// W1 [F][128], W2 [128][F] fp32 -> img [F / 32][FF_STAGE bytes]: the LDS image of every chunk (see the header); one thread per
// element pair of either matrix
__global__ __launch_bounds__(256) void ffn_split_weights_kernel(const float *__restrict__ w1, const float *__restrict__ w2, const int F,
                                                                unsigned short *__restrict__ img)
{
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  const long long n1 = (long long)F * FF_D;
  unsigned p1, p2, p3;
  if (e < n1) {                                                   // W1[u][k], k even: chunk u / 32, row u % 32, k-contiguous
    const int u = (int)(e / FF_D), k = (int)(e - (long long)u * FF_D);
    ff_split(w1[e], w1[e + 1], p1, p2, p3);
    unsigned short *d = img + (size_t)(u / FF_CH) * (FF_STAGE / 2) + (size_t)(u % FF_CH) * (FF_W1ROW / 2) + k;
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
  }
}

__global__ __launch_bounds__(FF_BLOCK, 1) void ffn_fused_split_kernel(
    const float *__restrict__ X, const long long ldx, const unsigned short *__restrict__ img, const int F,
    const float *__restrict__ b1, const float *__restrict__ b2, const float *__restrict__ gamma, const float *__restrict__ beta,
    const float eps, const long long M, float *__restrict__ Y, const long long ldy)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char ff_smem[];    // two stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const long long ntiles = (M + FF_BM - 1) / FF_BM;
  const int nch = F / FF_CH;

  const auto img_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(img), 0, (int)((long long)nch * FF_STAGE),
                                                          0x00020000);
  ff_i32x4 gst[FF_PIECES];                                        // the next chunk's image on its way to LDS
  auto fetch_stage = [&](const int chunk) {
#pragma unroll
    for (int i = 0; i < FF_PIECES; ++i)
      gst[i] = __builtin_amdgcn_raw_buffer_load_b128(img_rsrc, tid * 16, chunk * FF_STAGE + i * (FF_BLOCK * 16), 0);
  };
  auto store_stage = [&](unsigned char *st) {
#pragma unroll
    for (int i = 0; i < FF_PIECES; ++i) *reinterpret_cast<ff_i32x4 *>(st + i * (FF_BLOCK * 16) + tid * 16) = gst[i];
  };

  // fragment addresses inside a stage: W1 (A operand of GEMM 1): hidden row r, k = 16 s + 8 h ..; W2 (A operand of GEMM 2):
  // output column 32 cb + r, slots of k-step s and half h
  const int w1_off = r * FF_W1ROW + 16 * h;                       // + 32 s, + plane
  const int w2_off = FF_W2OFF + r * FF_W2ROW + 16 * h;            // + cb * 32 rows, + 32 s, + plane

  fetch_stage(0);
  store_stage(ff_smem);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  int cur = 0;                                                    // stage holding the chunk about to be used

  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // ---- this wave's 32 rows of x as GEMM 1's B fragments: lane (r, h) takes k = 16 s + 8 h .. + 7 of row r ----------------
    const long long row0 = tile * FF_BM + 32 * wave;
    const long long left = M - row0;
    const unsigned xbytes = (unsigned)((left < 32 ? (left < 0 ? 0 : left) : 32) * ldx * 4);
    const auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X + row0 * ldx), 0, (int)xbytes, 0x00020000);
    ff_bf16x8 xp[8][3];
    {
      const int xoff = (int)((r * ldx + 8 * h) * 4);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const ff_f32x4 a = __builtin_bit_cast(ff_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, xoff, s * 64, 0));
        const ff_f32x4 b = __builtin_bit_cast(ff_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, xoff, s * 64 + 16, 0));
        ff_split8(a, b, xp[s]);
      }
    }
    ff_f32x16 y_hi[4], y_lo[4];                                   // D2[col block][cols 32][rows 32]
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int j = 0; j < 16; ++j) y_hi[cb][j] = y_lo[cb][j] = 0.f;

    for (int chunk = 0; chunk < nch; ++chunk) {
      const unsigned char *st = ff_smem + cur * FF_STAGE;
      // the next chunk's image (the next tile's first one after the last): requested now, stored behind this chunk's MFMAs
      fetch_stage(chunk + 1 < nch ? chunk + 1 : 0);
      const ff_f32x4 bq0 = *reinterpret_cast<const ff_f32x4 *>(b1 + chunk * FF_CH + 4 * h);
      const ff_f32x4 bq1 = *reinterpret_cast<const ff_f32x4 *>(b1 + chunk * FF_CH + 8 + 4 * h);
      const ff_f32x4 bq2 = *reinterpret_cast<const ff_f32x4 *>(b1 + chunk * FF_CH + 16 + 4 * h);
      const ff_f32x4 bq3 = *reinterpret_cast<const ff_f32x4 *>(b1 + chunk * FF_CH + 24 + 4 * h);
      __builtin_amdgcn_sched_barrier(0);

      // ---- GEMM 1: D1[hidden 8 q + 4 h + t][row r] over k = 128.  One wave per SIMD: nobody else covers an LDS round trip, so
      // the fragments of k-step s + 1 are requested before the MFMAs of step s (the compiler, left alone, read each fragment
      // right in front of its first MFMA and waited: ~2 400 of a chunk's 5 600 cycles) -----------------------------------------
      ff_f32x16 h_hi, h_lo;
#pragma unroll
      for (int j = 0; j < 16; ++j) h_hi[j] = h_lo[j] = 0.f;
      ff_bf16x8 wa[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) wa[0][p] = *reinterpret_cast<const ff_bf16x8 *>(st + w1_off + p * FF_W1PLANE);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s < 7) {
#pragma unroll
          for (int p = 0; p < 3; ++p)
            wa[(s + 1) & 1][p] = *reinterpret_cast<const ff_bf16x8 *>(st + w1_off + 32 * (s + 1) + p * FF_W1PLANE);
        }
        const ff_bf16x8 a1 = wa[s & 1][0], a2 = wa[s & 1][1], a3 = wa[s & 1][2];
        h_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xp[s][2], h_lo, 0, 0, 0);
        h_hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xp[s][0], h_hi, 0, 0, 0);
        h_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, xp[s][0], h_lo, 0, 0, 0);
        h_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xp[s][1], h_lo, 0, 0, 0);
        h_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xp[s][1], h_lo, 0, 0, 0);
        h_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xp[s][0], h_lo, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // GEMM 2's first fragments fly while the hidden values are finished
      ff_bf16x8 wb[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) wb[0][p] = *reinterpret_cast<const ff_bf16x8 *>(st + w2_off + p * FF_W2PLANE);
      // ---- bias, ReLU, split: the B fragments of GEMM 2 (k-step s: accumulator quartets 2 s and 2 s + 1) ----------------------
      ff_bf16x8 hp[2][3];
      {
        ff_f32x4 hv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          hv[0][t] = fmaxf((h_hi[t] + h_lo[t]) + bq0[t], 0.f);
          hv[1][t] = fmaxf((h_hi[4 + t] + h_lo[4 + t]) + bq1[t], 0.f);
          hv[2][t] = fmaxf((h_hi[8 + t] + h_lo[8 + t]) + bq2[t], 0.f);
          hv[3][t] = fmaxf((h_hi[12 + t] + h_lo[12 + t]) + bq3[t], 0.f);
        }
        ff_split8(hv[0], hv[1], hp[0]);
        ff_split8(hv[2], hv[3], hp[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- GEMM 2: D2[col 32 cb + 8 q + 4 h + t][row r] += W2 . h over this chunk's 32 hidden units; group g = (s, cb), the
      // next group's fragments requested before this group's MFMAs; the next chunk's image goes to the other stage (everyone left
      // it at the previous barrier) two 16-byte pieces per group -------------------------------------------------------------------
      unsigned char *nst = ff_smem + (cur ^ 1) * FF_STAGE;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int sg = g >> 2, cb = g & 3;
        if (g < 7) {
          const int s2 = (g + 1) >> 2, cb2 = (g + 1) & 3;
#pragma unroll
          for (int p = 0; p < 3; ++p)
            wb[(g + 1) & 1][p] = *reinterpret_cast<const ff_bf16x8 *>(st + w2_off + cb2 * 32 * FF_W2ROW + 32 * s2 + p * FF_W2PLANE);
        }
        const ff_bf16x8 a1 = wb[g & 1][0], a2 = wb[g & 1][1], a3 = wb[g & 1][2];
        y_lo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, hp[sg][2], y_lo[cb], 0, 0, 0);
        y_hi[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, hp[sg][0], y_hi[cb], 0, 0, 0);
        y_lo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, hp[sg][0], y_lo[cb], 0, 0, 0);
        y_lo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, hp[sg][1], y_lo[cb], 0, 0, 0);
        y_lo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, hp[sg][1], y_lo[cb], 0, 0, 0);
        y_lo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, hp[sg][0], y_lo[cb], 0, 0, 0);
#pragma unroll
        for (int i = 2 * g; i < 2 * g + 2 && i < FF_PIECES; ++i)
          *reinterpret_cast<ff_i32x4 *>(nst + i * (FF_BLOCK * 16) + tid * 16) = gst[i];
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      cur ^= 1;
    }

    // ---- epilogue: y_*[cb][4 q + t] = y[row r][column 32 cb + 8 q + 4 h + t]; + b2 + residual, LayerNorm, store -----------------
    const unsigned ybytes = (unsigned)((left < 32 ? (left < 0 ? 0 : left) : 32) * ldy * 4);
    const auto yr = __builtin_amdgcn_make_buffer_rsrc(Y + row0 * ldy, 0, (int)ybytes, 0x00020000);
    float v[64];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * cb + 8 * q + 4 * h;
        const ff_f32x4 res = __builtin_bit_cast(ff_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)((r * ldx + c0) * 4), 0, 0));
        const ff_f32x4 bb = b2 ? *reinterpret_cast<const ff_f32x4 *>(b2 + c0) : ff_f32x4{0.f, 0.f, 0.f, 0.f};
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
        const ff_f32x4 gq = *reinterpret_cast<const ff_f32x4 *>(gamma + c0);
        const ff_f32x4 eq = *reinterpret_cast<const ff_f32x4 *>(beta + c0);
        ff_f32x4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = v[16 * cb + 4 * q + t] * rstd * gq[t] + eq[t];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ff_i32x4, o), yr, (int)((r * ldy + c0) * 4), 0, 0);
      }
  }
}

// img_ws: (F / 32) * 57 344 bytes, 16-byte aligned, refilled on every call.  -4: geometry not covered.
int launch_ffn_fused_split(const float *x, long long ldx, const float *w1, const float *b1, const float *w2, const float *b2,
                           const float *gamma, const float *beta, float eps, int F, long long rows, void *img_ws, float *out,
                           long long ldo, hipStream_t stream)
{
  if (rows <= 0) return 0;
  if (F <= 0 || F % FF_CH) return -4;
  if ((long long)(F / FF_CH) * FF_STAGE > 0x7fffffffLL) return -4;
  if (32LL * (ldx > ldo ? ldx : ldo) * 4 > 0x7fffffffLL) return -4;
  static thread_local int attr_dev = -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -4;
  if (attr_dev != dev) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&ffn_fused_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * FF_STAGE) != hipSuccess)
      return -4;
    attr_dev = dev;
  }
  const long long pairs = (long long)F * FF_D;                    // element pairs of both matrices
  hipLaunchKernelGGL(ffn_split_weights_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, w1, w2, F,
                     static_cast<unsigned short *>(img_ws));
  const long long ntiles = (rows + FF_BM - 1) / FF_BM;
  const unsigned gx = (unsigned)(ntiles < 256 ? ntiles : 256);    // persistent: one workgroup per CU
  hipLaunchKernelGGL(ffn_fused_split_kernel, dim3(gx), dim3(FF_BLOCK), 2 * FF_STAGE, stream, x, ldx,
                     static_cast<const unsigned short *>(img_ws), F, b1, b2, gamma, beta, eps, rows, out, ldo);
  return (int)hipGetLastError();
}

}  // namespace pct
