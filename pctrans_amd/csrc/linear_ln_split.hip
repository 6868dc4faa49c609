// y = LayerNorm(residual + x · Wᵀ + b) for a 128-wide output and any K (multiple of 32) on MI355X (gfx950): the encoder
// FFN's second Linear (K = dim_feedforward = 1024) fused with `src + dropout(src2)` and `norm2`
// (pixel_decoder/msdeformattn.py:122-131 of the reference: linear2, dropout3, norm2), fp32 in, fp32 out.
//
// Same arithmetic as linear_k128_split.hip: every fp32 operand is the exact sum of three bf16 pieces and a product is
// evaluated from its six leading partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (a1w1 into one
// accumulator, the five small terms into another) -- fp32 accuracy at 6/16 of the fp32 MFMA cycles.  At K = 1024 the
// weights no longer fit a wave's registers, so this is a tiled GEMM:
//   * W is split once per call by `split_bf16x3_kernel` into three bf16 planes [3][128][K] (768 KB at K = 1024: L2
//     resident); x is split in the kernel on its way from registers into LDS;
//   * workgroup tile 128 rows x 128 columns, 4 waves as 2 x 2 (64 x 64 each = 2 x 2 MFMA blocks, 128 accumulator
//     VGPRs), K step 32: per step and wave 24 ds_read_b128 feed 48 MFMAs;
//   * LDS stage = 3 planes x 128 rows x 80 B for x and the same for W (row stride 64 + 16 B: every 16-lane group of a
//     ds_read_b128 covers all 64 banks), 60 KB -> two workgroups per CU cover each other's barriers; the next step's
//     operands are prefetched into registers during the MFMAs;
//   * the MFMA computes the transposed tile (W pieces as the A operand), so a lane holds 4-column groups of ONE row:
//     bias / residual / LayerNorm / store are dwordx4, the row statistics are in-lane sums + one cross-half exchange +
//     one LDS hop between the two column halves.
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef PCT_LIN_PRIO
#define PCT_LIN_PRIO 0
#endif
#ifndef PCT_LIN_STAGGER
#define PCT_LIN_STAGGER 1          /* static priority for the second half of the grid: 4.605 -> 4.53 ms at 2.8 M rows (same-box A/B) */
#endif
#ifndef PCT_LLS_KO_ALIAS
#define PCT_LLS_KO_ALIAS 0    /* knock-out (WRONG RESULTS, timing only): x and residual rows of every tile read from the first tiles (L2-resident) */
#endif
#if PCT_LLS_KO_ALIAS && !defined(PCT_EXPERIMENT_BUILD)
#error "PCT_LLS_KO_ALIAS gives wrong results: add -DPCT_EXPERIMENT_BUILD"
#endif

namespace pct {

typedef float lls_f32x16 __attribute__((ext_vector_type(16)));
typedef float lls_f32x4 __attribute__((ext_vector_type(4)));
typedef float lls_f32x2 __attribute__((ext_vector_type(2)));
typedef int lls_i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 lls_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 lls_bf16x2 __attribute__((ext_vector_type(2)));

constexpr int LLS_BLOCK = 256;
constexpr int LLS_BM = 128;                       // rows per workgroup tile
constexpr int LLS_N = 128;                        // output columns (LayerNorm width)
constexpr int LLS_BK = 32;                        // K step
constexpr int LLS_ROWB = 80;                      // bytes per row and plane in a stage (32 bf16 + 16 pad)
constexpr int LLS_PLANE = 128 * LLS_ROWB;         // 10240
constexpr int LLS_OPER = 3 * LLS_PLANE;           // one operand's three planes

__device__ __forceinline__ void lls_split(const float x, const float y, unsigned &p1, unsigned &p2, unsigned &p3)
{
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(lls_f32x2{x, y}, lls_bf16x2));
  const float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);   // exact
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(lls_f32x2{rx, ry}, lls_bf16x2));
  const float sx = rx - __uint_as_float(p2 << 16), sy = ry - __uint_as_float(p2 & 0xffff0000u); // exact
  p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(lls_f32x2{sx, sy}, lls_bf16x2));
}

// out[p * count + i] = piece p of w[i]  (count even)
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float *__restrict__ w, const long long count,
                                                           unsigned short *__restrict__ out)
{
  const long long i = 2 * ((long long)blockIdx.x * 256 + threadIdx.x);
  if (i >= count) return;
  unsigned p1, p2, p3;
  lls_split(w[i], w[i + 1], p1, p2, p3);
  *reinterpret_cast<unsigned *>(out + i) = p1;
  *reinterpret_cast<unsigned *>(out + count + i) = p2;
  *reinterpret_cast<unsigned *>(out + 2 * count + i) = p3;
}

__global__ __launch_bounds__(LLS_BLOCK, 2) void linear_ln_split_kernel(
    const float *__restrict__ X, const long long ldx, const unsigned short *__restrict__ Wp, const int K,
    const float *__restrict__ bias, const float *__restrict__ R, const long long ldr, const float *__restrict__ gamma,
    const float *__restrict__ beta, const float eps, const long long M, float *__restrict__ Y, const long long ldy)
{
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * LLS_OPER];   // x planes | W planes
  __shared__ float red[2][LLS_BM];                                            // LayerNorm: per column-half row sums
  __shared__ __attribute__((aligned(16))) float gbuf[3][LLS_N];               // bias, gamma, beta
  unsigned char *const xs = lds, *const ws = lds + LLS_OPER;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
#if PCT_LIN_STAGGER
  // Two workgroups share a CU (and every SIMD's matrix pipe).  Left alone they fall into lockstep -- both in their MFMA block at
  // half rate each, then both splitting / staging with the matrix pipe idle.  A STATIC priority for the second half of the grid
  // (the workgroups dispatched onto already occupied CUs) lets that workgroup's MFMAs go first whenever both want the pipe, so the
  // pair settles into opposite phases: one multiplies while the other stages.
#ifndef PCT_LIN_STAGGER_BY
#define PCT_LIN_STAGGER_BY 0                                       /* 0: second half of the grid; 1: every second workgroup of an XCD */
#endif
  if (PCT_LIN_STAGGER_BY == 0 ? blockIdx.x >= (gridDim.x >> 1) : ((blockIdx.x >> 3) & 1)) __builtin_amdgcn_s_setprio(PCT_LIN_STAGGER);
#endif
                  // row half / column half of the 128 x 128 tile
  const int r = lane & 31, h = lane >> 5;

  if (tid < 128) {
    gbuf[0][tid] = bias ? bias[tid] : 0.f;
    gbuf[1][tid] = gamma[tid];
  } else {
    gbuf[2][tid - 128] = beta[tid - 128];
  }

  const long long ntiles = (M + LLS_BM - 1) / LLS_BM;
  const int ksteps = K / LLS_BK;
  auto tile_rsrc = [&](const float *base, const long long ld, const long long tile) {
    const long long row0 = tile * LLS_BM;
    const long long left = M - row0;                              // <= 0 past the end: an empty descriptor
    const unsigned bytes = (unsigned)((left < LLS_BM ? (left < 0 ? 0 : left) : LLS_BM) * ld * 4);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + row0 * ld), 0, (int)bytes, 0x00020000);
  };
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(Wp), 0, (int)(3LL * LLS_N * K * 2),
                                                        0x00020000);

  // x: thread -> row tid/2, 16 consecutive floats at 16*(tid%2) of the step's 32
  const int xg_voff = (int)(((tid >> 1) * ldx + 16 * (tid & 1)) * 4);
  const int xs_off = (tid >> 1) * LLS_ROWB + 32 * (tid & 1);
  // W pieces: 3 planes x 128 columns x 4 parts of 16 B per step = 1536 parts; thread takes part tid + 256 i, i.e.
  // plane i/2, column tid/4 + 64 (i%2), part tid%4: one lane offset + uniform / immediate terms
  const int wg_voff = (int)((((long long)(tid >> 2)) * K + 8 * (tid & 3)) * 2);
  const int ws_off = (tid >> 2) * LLS_ROWB + 16 * (tid & 3);

  lls_i32x4 gx[4], gw[6];
  auto fetch = [&](const long long tile, const int step) {
    const auto rs = tile_rsrc(X, ldx, PCT_LLS_KO_ALIAS ? tile % 64 : tile);
#pragma unroll
    for (int q = 0; q < 4; ++q) gx[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, xg_voff, step * (LLS_BK * 4) + 16 * q, 0);
#pragma unroll
    for (int i = 0; i < 6; ++i)
      gw[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wg_voff, ((i >> 1) * LLS_N + (i & 1) * 64) * K * 2 + step * (LLS_BK * 2), 0);
  };
  auto stash = [&]() {
    unsigned p1[8], p2[8], p3[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const lls_f32x4 v = __builtin_bit_cast(lls_f32x4, gx[q]);
      lls_split(v[0], v[1], p1[2 * q], p2[2 * q], p3[2 * q]);
      lls_split(v[2], v[3], p1[2 * q + 1], p2[2 * q + 1], p3[2 * q + 1]);
    }
    unsigned char *p = xs + xs_off;
    *reinterpret_cast<lls_i32x4 *>(p) = lls_i32x4{(int)p1[0], (int)p1[1], (int)p1[2], (int)p1[3]};
    *reinterpret_cast<lls_i32x4 *>(p + 16) = lls_i32x4{(int)p1[4], (int)p1[5], (int)p1[6], (int)p1[7]};
    *reinterpret_cast<lls_i32x4 *>(p + LLS_PLANE) = lls_i32x4{(int)p2[0], (int)p2[1], (int)p2[2], (int)p2[3]};
    *reinterpret_cast<lls_i32x4 *>(p + LLS_PLANE + 16) = lls_i32x4{(int)p2[4], (int)p2[5], (int)p2[6], (int)p2[7]};
    *reinterpret_cast<lls_i32x4 *>(p + 2 * LLS_PLANE) = lls_i32x4{(int)p3[0], (int)p3[1], (int)p3[2], (int)p3[3]};
    *reinterpret_cast<lls_i32x4 *>(p + 2 * LLS_PLANE + 16) = lls_i32x4{(int)p3[4], (int)p3[5], (int)p3[6], (int)p3[7]};
#pragma unroll
    for (int i = 0; i < 6; ++i)
      *reinterpret_cast<lls_i32x4 *>(ws + ws_off + (i >> 1) * LLS_PLANE + (i & 1) * 64 * LLS_ROWB) = gw[i];
  };

  // operand reads: lane (r, h) takes 16 B = 8 k at 32*ks + 16*h of its row
  const int xr_off = (64 * wr + r) * LLS_ROWB + 16 * h;           // + 32 rb rows, + plane, + 32 ks
  const int wr_off = (64 * wc + r) * LLS_ROWB + 16 * h;

  long long tile = blockIdx.x;
  fetch(tile, 0);
  for (; tile < ntiles; tile += gridDim.x) {
    lls_f32x16 acc_hi[2][2], acc_lo[2][2];                        // [row block][column block]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc_hi[a][b][j] = acc_lo[a][b][j] = 0.f;

    for (int step = 0; step < ksteps; ++step) {
      __syncthreads();                                            // everyone is done reading the previous stage
      stash();
      __syncthreads();                                            // stage complete
      // next step's operands (the next tile's first step after the last one; past the end: empty descriptor)
      const bool last = step + 1 == ksteps;
      fetch(last ? tile + gridDim.x : tile, last ? 0 : step + 1);
      __builtin_amdgcn_sched_barrier(0);
#if PCT_LIN_PRIO
      __builtin_amdgcn_s_setprio(PCT_LIN_PRIO);
#endif
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        lls_bf16x8 a[2][3], w[2][3];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            a[b][p] = *reinterpret_cast<const lls_bf16x8 *>(xs + xr_off + b * 32 * LLS_ROWB + p * LLS_PLANE + 32 * ks);
            w[b][p] = *reinterpret_cast<const lls_bf16x8 *>(ws + wr_off + b * 32 * LLS_ROWB + p * LLS_PLANE + 32 * ks);
          }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            acc_lo[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[cb][0], a[rb][2], acc_lo[rb][cb], 0, 0, 0);
            acc_hi[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[cb][0], a[rb][0], acc_hi[rb][cb], 0, 0, 0);
            acc_lo[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[cb][2], a[rb][0], acc_lo[rb][cb], 0, 0, 0);
            acc_lo[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[cb][1], a[rb][1], acc_lo[rb][cb], 0, 0, 0);
            acc_lo[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[cb][0], a[rb][1], acc_lo[rb][cb], 0, 0, 0);
            acc_lo[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[cb][1], a[rb][0], acc_lo[rb][cb], 0, 0, 0);
          }
      }
    }

#if PCT_LIN_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // ---- epilogue: acc[rb][cb][4q + i] = y[row 64 wr + 32 rb + r][column 64 wc + 32 cb + 8q + 4h + i] ----------------
    const auto rr = tile_rsrc(R, ldr, tile);
    const auto ry = tile_rsrc(Y, ldy, tile);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int row = 64 * wr + 32 * rb + r;
      float v[32];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c0 = 64 * wc + 32 * cb + 8 * q + 4 * h;
          const lls_f32x4 res = __builtin_bit_cast(
              lls_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)((row * ldr + c0) * 4), 0, 0));
          const lls_f32x4 bq = *reinterpret_cast<const lls_f32x4 *>(&gbuf[0][c0]);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            v[16 * cb + 4 * q + i] = ((acc_hi[rb][cb][4 * q + i] + acc_lo[rb][cb][4 * q + i]) + bq[i]) + res[i];
        }
      // two-pass LayerNorm over the row's 128 columns: my 32, the other half of the wave, the other column half
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) t += v[j];
      t += __shfl_xor(t, 32);
      if (h == 0) red[wc][row] = t;
      __syncthreads();
      const float mean = (red[0][row] + red[1][row]) * (1.f / 128.f);
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        v[j] -= mean;
        s2 += v[j] * v[j];
      }
      s2 += __shfl_xor(s2, 32);
      __syncthreads();                                            // red free again
      if (h == 0) red[wc][row] = s2;
      __syncthreads();
      const float rstd = rsqrtf((red[0][row] + red[1][row]) * (1.f / 128.f) + eps);
      __syncthreads();                                            // red free for the next row block
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c0 = 64 * wc + 32 * cb + 8 * q + 4 * h;
          const lls_f32x4 gq = *reinterpret_cast<const lls_f32x4 *>(&gbuf[1][c0]);
          const lls_f32x4 eq = *reinterpret_cast<const lls_f32x4 *>(&gbuf[2][c0]);
          lls_f32x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = v[16 * cb + 4 * q + i] * rstd * gq[i] + eq[i];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lls_i32x4, o), ry, (int)((row * ldy + c0) * 4), 0, 0);
        }
    }
  }
}

// w_pieces: workspace of 3 * 128 * K bf16 (16-byte aligned), filled here from w on `stream` before the GEMM
int launch_linear_ln_split(const float *x, long long ldx, const float *w, unsigned short *w_pieces, int K, const float *bias,
                           const float *residual, long long ldr, const float *gamma, const float *beta, float eps,
                           long long rows, float *out, long long ldo, hipStream_t stream)
{
  if (rows <= 0) return 0;
  if (K <= 0 || K % LLS_BK) return -4;
  if ((long long)LLS_BM * (ldx > ldr ? (ldx > ldo ? ldx : ldo) : (ldr > ldo ? ldr : ldo)) * 4 > 0x7fffffffLL) return -4;
  const long long count = (long long)LLS_N * K;
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((count / 2 + 255) / 256)), dim3(256), 0, stream, w, count, w_pieces);
  const long long ntiles = (rows + LLS_BM - 1) / LLS_BM;
  const unsigned gx = (unsigned)(ntiles < 512 ? ntiles : 512);    // persistent: 2 workgroups per CU
  hipLaunchKernelGGL(linear_ln_split_kernel, dim3(gx), dim3(LLS_BLOCK), 0, stream, x, ldx, w_pieces, K, bias, residual, ldr,
                     gamma, beta, eps, rows, out, ldo);
  return (int)hipGetLastError();
}

}  // namespace pct
