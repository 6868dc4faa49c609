// fp32-accurate 1x1 convolution on NCHW maps for the pixel decoder's input projections on MI355X (gfx950):
//     y[n][co][p] = b[co] + sum_ci W[co][ci] * x[n][ci][p],      co = 128, ci % 16 == 0, p = H * W % 128 == 0
// (pixel_decoder/msdeformattn.py:213-226: `Conv2d(in_channels, conv_dim, kernel_size=1)` in front of GroupNorm(32, conv_dim),
// applied to res2 .. res5; with four encoder levels the res2 projection -- 256 -> 128 channels on 128 x 128 maps -- is a
// [2.1 M pixels x 256] x [256 x 128] product per step at batch 128).  As a strided-batched fp32 GEMM through the BLAS library
// (layers.Conv2d) the four levels took 2.78 ms per step at ~2 TB/s; here 1.76 ms at 2.3 - 3.3 TB/s (tools/bench_conv1x1.py).
//
// Same arithmetic as linear_k128_split.hip: every fp32 number is exactly the sum of three bf16 numbers, a product is
// evaluated from its six leading bf16 x bf16 terms (a1w1 | a1w2 + a2w1 | a2w2 + a1w3 + a3w1) accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 -- as accurate as an fp32 GEMM (tests/test_fused_ops_gpu.py, against fp64).  What differs is the
// operand layout: the reduction dimension of x is the SLOWEST one (NCHW: a pixel's channels are H * W floats apart), so the
// B operand of the MFMA -- 8 consecutive k for one pixel per lane -- cannot be read from rows.  A workgroup stages a
// [16 channels x 128 pixels] slab of x as it lies in memory (coalesced 512-byte rows), splits it once into three bf16
// planes in LDS, and the waves read their B fragments with ds_read_b64_tr_b16, the transposing LDS read (4 k x 16 pixels
// per 16-lane group, delivered k-major per pixel).  Row stride 320 B: the 32 lanes of a read pass hit 64 distinct banks.
// W (A operand) is split once per call by a small kernel into the exact LDS image of each 16-channel step
// ([step][plane][co][16 k], the two 16-byte halves of a row swapped on every second block of 8 rows so that a
// ds_read_b128 of 16 consecutive rows covers all 64 banks) and streamed step by step from L2.
// Workgroup = (image, 128 pixels), 4 waves = 2 x 2 blocks of 64 output channels x 64 pixels (4 MFMA tiles, 8 accumulators:
// the a1w1 chain and the five small terms apart, as in the Linear kernels); x and W of step s + 1 are fetched into registers
// while step s is multiplied, one barrier per step; 54 KB of LDS, two workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "attn_common.hpp"

namespace pct {

typedef float c1_f32x16 __attribute__((ext_vector_type(16)));
typedef float c1_f32x4 __attribute__((ext_vector_type(4)));
typedef float c1_f32x2 __attribute__((ext_vector_type(2)));
typedef int c1_i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 c1_bf16x2 __attribute__((ext_vector_type(2)));

constexpr int C1_CO = 128;                 // output channels
constexpr int C1_PT = 128;                 // pixels per workgroup
constexpr int C1_KS = 16;                  // channels per step
constexpr int C1_BLOCK = 256;
constexpr int C1_XROW = 320;               // bytes per k-row of an x plane in LDS (128 px x 2 B + 64 pad)
constexpr int C1_XPLANE = C1_KS * C1_XROW;
constexpr int C1_XBUF = 3 * C1_XPLANE;     // 15 360
constexpr int C1_WPLANE = C1_CO * 32;      // 128 rows x 16 k x 2 B
constexpr int C1_WBUF = 3 * C1_WPLANE;     // 12 288

// (x, y) -> three packed bf16 pairs (low half = x's piece), x = x1 + x2 + x3 exactly (as linear_k128_split.hip)
__device__ __forceinline__ void c1_split(const float x, const float y, unsigned &p1, unsigned &p2, unsigned &p3)
{
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(c1_f32x2{x, y}, c1_bf16x2));
  const float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(c1_f32x2{rx, ry}, c1_bf16x2));
  const float sx = rx - __uint_as_float(p2 << 16), sy = ry - __uint_as_float(p2 & 0xffff0000u);
  p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(c1_f32x2{sx, sy}, c1_bf16x2));
}

// W [128][K] fp32 -> ws [K / 16][3][128][16] bf16 in the LDS image order (see the header): one thread per element pair
__global__ __launch_bounds__(256) void conv1x1_split_weights_kernel(const float *__restrict__ w, const int K,
                                                                    unsigned short *__restrict__ ws)
{
  const int e = (blockIdx.x * 256 + threadIdx.x) * 2;            // element pair (co, k), k even
  if (e >= C1_CO * K) return;
  const int co = e / K, k = e - co * K;
  unsigned p1, p2, p3;
  c1_split(w[e], w[e + 1], p1, p2, p3);
  const int step = k / C1_KS, kk = k % C1_KS;
  const size_t base = (size_t)step * (C1_WBUF / 2) + (size_t)co * 16 + (size_t)((((kk >> 3) ^ ((co >> 3) & 1)) << 3) + (kk & 7));
  *reinterpret_cast<unsigned *>(ws + base) = p1;
  *reinterpret_cast<unsigned *>(ws + base + C1_WPLANE / 2) = p2;
  *reinterpret_cast<unsigned *>(ws + base + C1_WPLANE) = p3;
}

// TOKENS: the output goes to the encoder's token rows instead -- y[n][row_offset + p][co], channels fastest, which is what a lane
// holds (4 consecutive channels of one pixel per accumulator quartet: 16-byte stores, 16 per lane instead of 64 dword stores) --
// and the GroupNorm(32, 128) that follows the projection (msdeformattn.py:220-224) gets its statistics from here: 4 channels per
// group = exactly one accumulator quartet, so a wave forms (mean, sum of centred squares) of its 256 values per group in
// registers (two passes, as gn_stats_kernel does over the whole map) and writes them as a partial record; gn_finalize_kernel
// combines the records of an image (Chan's formula), gn_apply_tokens_kernel normalises the rows in place.  The [N, 128, H, W]
// intermediate, its two re-reads and the transposing copy are gone.
template <bool TOKENS>
__global__ __launch_bounds__(C1_BLOCK, 2) void conv1x1_nchw_split_kernel(const float *__restrict__ x,
                                                                         const unsigned short *__restrict__ ws,
                                                                         const float *__restrict__ bias, const int K,
                                                                         const int HW, float *__restrict__ y,
                                                                         const long long y_batch_stride,
                                                                         float *__restrict__ partial)
{
  __shared__ __attribute__((aligned(16))) unsigned char xs[2][C1_XBUF];
  __shared__ __attribute__((aligned(16))) unsigned char wsm[2][C1_WBUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = HW / C1_PT;
  const int n = blockIdx.x / tiles, p0 = (blockIdx.x - n * tiles) * C1_PT;
  const float *xn = x + (size_t)n * K * HW + p0;
  const int nsteps = K / C1_KS;

  // staging roles: x -- row tid / 16 of the step, 8 pixels from (tid % 16) * 8; W -- 48 consecutive bytes of the step's image
  const int sx_row = tid >> 4, sx_px = (tid & 15) * 8;
  const float *xsrc = xn + (size_t)sx_row * HW + sx_px;
  const int sx_off = sx_row * C1_XROW + sx_px * 2;
  c1_f32x4 gx[2];
  c1_i32x4 gw[3];
  auto fetch = [&](const int step) {
    const float *p = xsrc + (size_t)step * C1_KS * HW;
    gx[0] = *reinterpret_cast<const c1_f32x4 *>(p);
    gx[1] = *reinterpret_cast<const c1_f32x4 *>(p + 4);
    const c1_i32x4 *wp = reinterpret_cast<const c1_i32x4 *>(ws + (size_t)step * (C1_WBUF / 2)) + tid;
#pragma unroll
    for (int i = 0; i < 3; ++i) gw[i] = wp[i * C1_BLOCK];
  };
  auto stash = [&](const int buf) {
    unsigned a[4], b[4], c[4];
    c1_split(gx[0][0], gx[0][1], a[0], b[0], c[0]);
    c1_split(gx[0][2], gx[0][3], a[1], b[1], c[1]);
    c1_split(gx[1][0], gx[1][1], a[2], b[2], c[2]);
    c1_split(gx[1][2], gx[1][3], a[3], b[3], c[3]);
    unsigned char *d = xs[buf] + sx_off;
    *reinterpret_cast<c1_i32x4 *>(d) = c1_i32x4{(int)a[0], (int)a[1], (int)a[2], (int)a[3]};
    *reinterpret_cast<c1_i32x4 *>(d + C1_XPLANE) = c1_i32x4{(int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    *reinterpret_cast<c1_i32x4 *>(d + 2 * C1_XPLANE) = c1_i32x4{(int)c[0], (int)c[1], (int)c[2], (int)c[3]};
    c1_i32x4 *wd = reinterpret_cast<c1_i32x4 *>(wsm[buf]) + tid;
#pragma unroll
    for (int i = 0; i < 3; ++i) wd[i * C1_BLOCK] = gw[i];
  };

  // fragment addresses (fixed): wave = (channel half, pixel half)
  const int r = lane & 31, h = lane >> 5;
  const int co_half = wave >> 1, px_half = wave & 1;
  int a_off[2];                                                   // A operand: W[co][8 h .. 8 h + 7] of row block rb
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int co = co_half * 64 + rb * 32 + r;
    a_off[rb] = co * 32 + ((h ^ ((co >> 3) & 1)) << 4);
  }
  // B operand through the transposing read: 16-lane group g16 = lane / 16 covers pixels 16 (g16 % 2) .. + 15 and k = 8 (g16 / 2) ..;
  // lane 4 q + pp of the group supplies row q (+ 4 for the second read), columns 4 pp .. 4 pp + 3
  const int g16 = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  int b_off[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
    b_off[cb] = (8 * (g16 >> 1) + q4) * C1_XROW + (px_half * 64 + cb * 32 + 16 * (g16 & 1) + 4 * pp) * 2;

  c1_f32x16 acc_hi[2][2], acc_lo[2][2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc_hi[rb][cb][j] = acc_lo[rb][cb][j] = 0.f;

  fetch(0);
  stash(0);
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    if (step + 1 < nsteps) fetch(step + 1);
    __builtin_amdgcn_sched_barrier(0);                            // the fetches ahead of the MFMAs
    bf16x8 wa[2][3], xb[2][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        wa[rb][pl] = *reinterpret_cast<const bf16x8 *>(wsm[buf] + pl * C1_WPLANE + a_off[rb]);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const unsigned char *p = xs[buf] + pl * C1_XPLANE + b_off[cb];
        const bf16x4 lo = lds_read_tr16_b64(p), hi = lds_read_tr16_b64(p + 4 * C1_XROW);
        xb[cb][pl] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        acc_lo[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[rb][0], xb[cb][2], acc_lo[rb][cb], 0, 0, 0);
        acc_hi[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[rb][0], xb[cb][0], acc_hi[rb][cb], 0, 0, 0);
        acc_lo[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[rb][2], xb[cb][0], acc_lo[rb][cb], 0, 0, 0);
        acc_lo[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[rb][1], xb[cb][1], acc_lo[rb][cb], 0, 0, 0);
        acc_lo[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[rb][0], xb[cb][1], acc_lo[rb][cb], 0, 0, 0);
        acc_lo[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[rb][1], xb[cb][0], acc_lo[rb][cb], 0, 0, 0);
      }
    if (step + 1 < nsteps) stash(buf ^ 1);                       // (everyone left that buffer before the previous barrier)
    __syncthreads();
  }

  // epilogue: acc[4 q + t] = y[co = block + 8 q + 4 h + t][pixel = block + r]
  if constexpr (TOKENS) {
    float *yn = y + (size_t)n * y_batch_stride + (size_t)p0 * C1_CO;          // (y already points at the level's first row)
    const int tile = blockIdx.x - n * tiles;
    float *prec = partial + ((size_t)((size_t)n * tiles + tile) * 2 + px_half) * 64;      // [32 groups][mean, M2] of this wave pair
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int co0 = co_half * 64 + rb * 32 + 4 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const c1_f32x4 bq = bias ? *reinterpret_cast<const c1_f32x4 *>(bias + co0 + 8 * q) : c1_f32x4{0.f, 0.f, 0.f, 0.f};
        c1_f32x4 v[2];
        float sm = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[cb][t] = (acc_hi[rb][cb][4 * q + t] + acc_lo[rb][cb][4 * q + t]) + bq[t];
          sm += (v[cb][0] + v[cb][1]) + (v[cb][2] + v[cb][3]);
          *reinterpret_cast<c1_f32x4 *>(yn + (size_t)(px_half * 64 + cb * 32 + r) * C1_CO + co0 + 8 * q) = v[cb];
        }
        // the group's 256 values of this wave: 8 per lane over the 32 lanes of a half wave (same h)
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) sm += __shfl_xor(sm, o);
        const float mean = sm * (1.f / 256.f);
        float m2 = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float d = v[cb][t] - mean;
            m2 += d * d;
          }
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) m2 += __shfl_xor(m2, o);
        if (r == 0) {
          const int g = co_half * 16 + rb * 8 + 2 * q + h;
          *reinterpret_cast<c1_f32x2 *>(prec + 2 * g) = c1_f32x2{mean, m2};
        }
      }
    }
    return;
  }
  // a wave instruction writes 32 consecutive pixels of two channels (128 B each)
  float *yn = y + (size_t)n * C1_CO * HW + p0;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int co0 = co_half * 64 + rb * 32 + 4 * h;
    c1_f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bq[q] = bias ? *reinterpret_cast<const c1_f32x4 *>(bias + co0 + 8 * q) : c1_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      float *dst = yn + px_half * 64 + cb * 32 + r;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          dst[(size_t)(co0 + 8 * q + t) * HW] = (acc_hi[rb][cb][4 * q + t] + acc_lo[rb][cb][4 * q + t]) + bq[q][t];
    }
  }
}

// ---- GroupNorm(32, 128) on token rows from the partial records of conv1x1_nchw_split_kernel<true> --------------------------
// stats[(n * 32 + g) * 2 + {0, 1}] = mean, rstd: one wave per (image, group) combines T records of 256 values each
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float *__restrict__ partial, const int T, const float eps,
                                                         float *__restrict__ stats)
{
  const int n = blockIdx.x >> 5, g = blockIdx.x & 31, lane = threadIdx.x;
  const float *p = partial + (size_t)n * T * 64 + 2 * g;
  float sm = 0.f;
  for (int t = lane; t < T; t += 64) sm += p[(size_t)t * 64];
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) sm += __shfl_xor(sm, o);
  const float mean = sm / (float)T;
  float m2 = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float d = p[(size_t)t * 64] - mean;
    m2 += p[(size_t)t * 64 + 1] + 256.f * d * d;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m2 += __shfl_xor(m2, o);
  if (lane == 0) {
    stats[2 * blockIdx.x] = mean;
    stats[2 * blockIdx.x + 1] = rsqrtf(m2 / (256.f * (float)T) + eps);
  }
}

// rows [n][p][128] in place: one thread per (row, group of 4 channels)
__global__ __launch_bounds__(256) void gn_apply_tokens_kernel(float *__restrict__ y, const long long y_batch_stride, const int HW,
                                                              const float *__restrict__ stats, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta)
{
  const int n = blockIdx.y;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;              // float4 index inside the level's rows of image n
  if (i >= (long long)HW * 32) return;
  const int g = (int)(i & 31);
  const float mean = stats[2 * (n * 32 + g)], rstd = stats[2 * (n * 32 + g) + 1];
  const c1_f32x4 ga = *reinterpret_cast<const c1_f32x4 *>(gamma + 4 * g), be = *reinterpret_cast<const c1_f32x4 *>(beta + 4 * g);
  c1_f32x4 *p = reinterpret_cast<c1_f32x4 *>(y + (size_t)n * y_batch_stride) + i;
  const c1_f32x4 v = *p;
  c1_f32x4 o;
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = (v[t] - mean) * rstd * ga[t] + be[t];
  *p = o;
}

// -4: geometry not covered (the caller keeps the library convolution)
int launch_conv1x1_nchw_split(const float *x, const float *w, const float *bias, unsigned short *w_split_ws, int N, int K, int HW,
                              float *y, hipStream_t stream)
{
  if (N == 0) return 0;
  if (K % C1_KS != 0 || HW % C1_PT != 0 || K <= 0) return -4;
  if ((long long)N * (HW / C1_PT) >= 0x7fffffffLL) return -4;
  hipLaunchKernelGGL(conv1x1_split_weights_kernel, dim3((unsigned)((C1_CO * K / 2 + 255) / 256)), dim3(256), 0, stream, w, K,
                     w_split_ws);
  hipLaunchKernelGGL(conv1x1_nchw_split_kernel<false>, dim3((unsigned)(N * (HW / C1_PT))), dim3(C1_BLOCK), 0, stream, x, w_split_ws,
                     bias, K, HW, y, 0LL, static_cast<float *>(nullptr));
  return (int)hipGetLastError();
}

// 1x1 projection + GroupNorm(32, 128) + flatten into token rows: out[n][p][128] at `out` (the level's first row of image 0),
// images out_batch_stride floats apart.  partial_ws: N * (HW / 128) * 2 * 64 floats; stats: N * 64 floats.  Four launches.
int launch_conv1x1_groupnorm_tokens(const float *x, const float *w, const float *bias, unsigned short *w_split_ws, const float *gamma,
                                    const float *beta, float eps, int N, int K, int HW, float *partial_ws, float *stats,
                                    float *out, long long out_batch_stride, hipStream_t stream)
{
  if (N == 0) return 0;
  if (K % C1_KS != 0 || HW % C1_PT != 0 || K <= 0) return -4;
  if ((long long)N * (HW / C1_PT) >= 0x7fffffffLL || N > 65535) return -4;
  const int tiles = HW / C1_PT;
  hipLaunchKernelGGL(conv1x1_split_weights_kernel, dim3((unsigned)((C1_CO * K / 2 + 255) / 256)), dim3(256), 0, stream, w, K,
                     w_split_ws);
  hipLaunchKernelGGL(conv1x1_nchw_split_kernel<true>, dim3((unsigned)(N * tiles)), dim3(C1_BLOCK), 0, stream, x, w_split_ws, bias, K,
                     HW, out, out_batch_stride, partial_ws);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)(N * 32)), dim3(64), 0, stream, partial_ws, 2 * tiles, eps, stats);
  hipLaunchKernelGGL(gn_apply_tokens_kernel, dim3((unsigned)(((long long)HW * 32 + 255) / 256), (unsigned)N), dim3(256), 0, stream, out,
                     out_batch_stride, HW, stats, gamma, beta);
  return (int)hipGetLastError();
}

}  // namespace pct
