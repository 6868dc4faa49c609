// Skinny fp32 GEMM for the encoder's K = 128 projections on MI355X (gfx950): y = act(x · Wᵀ + b) and the fused
// y = LayerNorm(residual + x · Wᵀ + b).
//
// Reference: the Linear layers of MSDeformAttn (ops/modules/ms_deform_attn.py:64-67,96-110: value_proj,
// sampling_offsets, attention_weights, output_proj) and of the encoder layer (pixel_decoder/msdeformattn.py:100-131:
// linear1 + ReLU, `src + dropout(src2)` → `norm1`).  x is [rows, 128] with rows = N·S ≈ 1.4 M, W is [n, 128].
//
// These GEMMs are 45–365 GFLOP with 0.7–6 GB of traffic: at the fp32 MFMA rate (157 TFLOP/s) they sit right at the
// HBM/compute ridge, and hipBLASLt's generic tiles reach 93–114 TFLOP/s on them.  Shape-specialised design:
//   * a wave owns a 32-column slice of W for its whole life: 32 × 128 fp32 = 64 VGPRs per lane, loaded once, already
//     in the B-operand layout of v_mfma_f32_32x32x2_f32 — W never touches LDS and is never re-fetched;
//   * the k index of MFMA step t is permuted to k(t, h) = 8·(t/4) + 4·h + t%4 (h = lane/32) on BOTH operands, so a
//     lane's operands for four consecutive steps are 16 contiguous bytes of its row (one dwordx4 / ds_read_b128);
//   * the 4 waves of a workgroup cover 128 output columns of the same 32 rows: the A tile is fetched once per
//     workgroup with coalesced dwordx4 loads into a double-buffered LDS image (row stride 132 floats: every 16-lane
//     group of a ds_read_b128 lands on 64 different banks) while the previous tile's MFMAs run; one barrier per tile.
//     (Each wave loading the whole tile itself — no LDS — capped at 55 % MFMA utilisation on L2 bandwidth.)
//   * per tile a wave issues 64 MFMAs into two independent 16-VGPR accumulator chains;
//   * epilogue in the accumulator layout (lane = column, 16 rows per lane): bias, ReLU, or residual + two-pass
//     LayerNorm with DPP/readlane row sums and one LDS hop across the 4 waves; loads/stores go through buffer
//     descriptors re-based per tile (32-bit offsets, the row bound is the descriptor size), 128-B row segments.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace pct {

typedef float lin_f32x16 __attribute__((ext_vector_type(16)));
typedef float lin_f32x4 __attribute__((ext_vector_type(4)));

constexpr int LIN_K = 128;
constexpr int LIN_BLOCK = 256;

enum { LIN_EPI_BIAS = 0, LIN_EPI_BIAS_RELU = 1, LIN_EPI_RES_LN = 2 };

template <int CTRL>
__device__ __forceinline__ float lin_dpp(float v)
{
  const int i = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, true));
}

// sum over the 32 lanes of each wave half (lanes 0-31 / 32-63); result valid in every lane of the half
__device__ __forceinline__ float lin_half_sum(float v)
{
  v += lin_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += lin_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += lin_dpp<0x141>(v);   // row_half_mirror
  v += lin_dpp<0x140>(v);   // row_mirror  -> every lane of a 16-lane row holds the row's sum
  const int vi = __float_as_int(v);                                 // (readlane takes and returns the bit pattern)
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
  return (threadIdx.x & 32) ? r2 + r3 : r0 + r1;
}

constexpr int LIN_LDA = 132;                      // LDS row stride of the A image (floats)

template <int EPI, bool HAS_X2>
__global__ __launch_bounds__(LIN_BLOCK, 2) void linear_k128_kernel(
    const float *__restrict__ X, const long long ldx, const float *__restrict__ X2, const long long ldx2,
    const int x2_period, const float *__restrict__ W, const float *__restrict__ bias, const long long M, const int N, float *__restrict__ Y, const long long ldy, const float *__restrict__ R,
    const long long ldr, const float *__restrict__ gamma, const float *__restrict__ beta, const float eps)
{
  __shared__ __attribute__((aligned(16))) float abuf[2][32 * LIN_LDA];
  __shared__ float red[4][32];                     // LayerNorm: per-wave partial row sums
  __shared__ float stat[32];
  typedef int lin_i32x4 __attribute__((ext_vector_type(4)));

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int col = blockIdx.y * 128 + wave * 32 + r;          // my W row (B operand) == my output column (C layout)
  // n % 32 == 0: a wave whose 32-column slice lies past n only helps staging the A tiles (wave-uniform)
  const bool active = (int)blockIdx.y * 128 + wave * 32 < N;

  float wreg[64];
#pragma unroll
  for (int t = 0; t < 64; ++t) wreg[t] = 0.f;
  if (active) {
    const float *wp = W + (long long)col * LIN_K + 4 * h;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const lin_f32x4 t = *reinterpret_cast<const lin_f32x4 *>(wp + 8 * u);
      wreg[4 * u] = t[0];
      wreg[4 * u + 1] = t[1];
      wreg[4 * u + 2] = t[2];
      wreg[4 * u + 3] = t[3];
    }
  }
  const float bcol = (bias && active) ? bias[col] : 0.f;
  float gcol = 1.f, becol = 0.f;
  if constexpr (EPI == LIN_EPI_RES_LN) {
    gcol = gamma[col];
    becol = beta[col];
  }

  const long long ntiles = (M + 31) / 32;
  // Buffer descriptors re-based on the tile's first row (SALU only): 32-bit lane offsets, the row bound is the
  // descriptor's size -- loads of rows past the end return 0 and their stores are dropped by the hardware.
  auto tile_rsrc = [&](const float *base, const long long ld, const long long tile) {
    const long long row0 = tile * 32;
    const long long left = M - row0;
    const unsigned bytes = (unsigned)((left < 32 ? left : 32) * ld * 4);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + row0 * ld), 0, (int)bytes, 0x00020000);
  };
  // A tile = 32 rows x 128 floats = 1024 float4; thread tid moves float4 number tid + 256*q (row 8q + tid/32)
  const int g_voff = (int)((tid >> 5) * ldx * 4) + 16 * (tid & 31);
  const int g_step = (int)(8 * ldx * 4);
  const int s_off = (tid >> 5) * LIN_LDA + 4 * (tid & 31);
  // HAS_X2: the A operand is X + X2 (the encoder's `query = src + pos`), added on the way into LDS.  X2 has
  // x2_period rows and repeats (pos is the same for every image of the batch: row i of X pairs with row i % period).
  const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(HAS_X2 ? X2 : X), 0,
                                                     (int)((long long)x2_period * ldx2 * 4), 0x00020000);
  auto fetch = [&](const long long tile, lin_i32x4 (&g)[4], lin_i32x4 (&g2)[4]) {
    const auto rs = tile_rsrc(X, ldx, tile);
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, g_voff, q * g_step, 0);
    if constexpr (HAS_X2) {
      const int base = (int)((tile * 32) % x2_period);              // uniform
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int row = base + (tid >> 5) + 8 * q;
        row = row >= x2_period ? row - x2_period : row;
        g2[q] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)(row * ldx2 * 4) + 16 * (tid & 31), 0, 0);
      }
    }
  };
  auto stash = [&](float *dst, const lin_i32x4 (&g)[4], const lin_i32x4 (&g2)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      lin_f32x4 v = __builtin_bit_cast(lin_f32x4, g[q]);
      if constexpr (HAS_X2) v += __builtin_bit_cast(lin_f32x4, g2[q]);
      *reinterpret_cast<lin_f32x4 *>(dst + s_off + q * 8 * LIN_LDA) = v;
    }
  };

  const int y_voff = (int)((4 * h * ldy + col) * 4);
  const int r_voff = (int)((4 * h * ldr + col) * 4);
  const int a_off = r * LIN_LDA + 4 * h;

  lin_i32x4 g[4], g2[4] = {};
  long long tile = blockIdx.x;
  if (tile < ntiles) {
    fetch(tile, g, g2);
    stash(abuf[0], g, g2);
  }
  __syncthreads();
  for (int buf = 0; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    const long long nxt = tile + gridDim.x;
    if (nxt < ntiles) fetch(nxt, g, g2);                          // in flight during the MFMAs

    const bool compute = active || EPI == LIN_EPI_RES_LN;
    // two accumulator chains: consecutive MFMAs never wait on each other's result
    lin_f32x16 acc0, acc1;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc0[j] = acc1[j] = 0.f;
    if (compute) {
    const float *ap = abuf[buf] + a_off;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const lin_f32x4 a = *reinterpret_cast<const lin_f32x4 *>(ap + 8 * u);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], wreg[4 * u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], wreg[4 * u + 1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], wreg[4 * u + 2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], wreg[4 * u + 3], acc1, 0, 0, 0);
    }
    }
    // before this tile's stores are issued: vmcnt counts loads and stores in issue order, a stash placed after the
    // stores would wait for them to reach memory
    if (nxt < ntiles) stash(abuf[buf ^ 1], g, g2);

    if (compute) {
    // accumulator layout: acc[j] = C[row = 8*(j/4) + 4*h + j%4][column = r]
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (acc0[j] + acc1[j]) + bcol;

    if constexpr (EPI == LIN_EPI_RES_LN) {
      const auto rr = tile_rsrc(R, ldr, tile);
#pragma unroll
      for (int j = 0; j < 16; ++j)
        v[j] += __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, r_voff, (int)((8 * (j / 4) + (j % 4)) * ldr * 4), 0));
      // pass 1: mean.  Row sums over my wave's 32 columns, then across the 4 waves through LDS.
      float part[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) part[j] = lin_half_sum(v[j]);
      if (r == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) red[wave][8 * (j / 4) + 4 * h + (j % 4)] = part[j];
      }
      __syncthreads();
      if (tid < 32) stat[tid] = (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) * (1.f / 128.f);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] -= stat[8 * (j / 4) + 4 * h + (j % 4)];
      // pass 2: variance of the centred values
#pragma unroll
      for (int j = 0; j < 16; ++j) part[j] = lin_half_sum(v[j] * v[j]);
      __syncthreads();                                            // stat / red free again
      if (r == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) red[wave][8 * (j / 4) + 4 * h + (j % 4)] = part[j];
      }
      __syncthreads();
      if (tid < 32)
        stat[tid] = rsqrtf((red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) * (1.f / 128.f) + eps);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = v[j] * stat[8 * (j / 4) + 4 * h + (j % 4)] * gcol + becol;
    } else if constexpr (EPI == LIN_EPI_BIAS_RELU) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    const auto ry = tile_rsrc(Y, ldy, tile);
#pragma unroll
    for (int j = 0; j < 16; ++j)
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v[j]), ry, y_voff, (int)((8 * (j / 4) + (j % 4)) * ldy * 4), 0);
    }

    __syncthreads();                  // next image complete; every wave is done reading this one
  }
}

int launch_linear_k128_split(const float *, long long, const float *, long long, long long, const float *, const float *,
                             long long, int, int, float *, long long, const float *, long long, const float *,
                             const float *, float, hipStream_t);

// n must be a multiple of 32 (LayerNorm variant: exactly 128); x rows 16-byte aligned.
// Default: the split-bf16 kernel of linear_k128_split.hip (same results to fp32 accuracy, 6/16 of the MFMA cycles);
// PCT_LIN_KERNEL=f32 selects the fp32-MFMA kernel of this file (A/B).
int launch_linear_k128(const float *x, long long ldx, const float *x2, long long ldx2, long long x2_period, const float *w,
                       const float *bias,
                       long long rows, int n, int epi,
                       float *y, long long ldy, const float *residual, long long ldr, const float *gamma,
                       const float *beta, float eps, hipStream_t stream)
{
  if (rows <= 0) return 0;
  static const bool use_f32 = [] { const char *e = getenv("PCT_LIN_KERNEL"); return e && e[0] == 'f'; }();
  if (!use_f32) {
    const int rc = launch_linear_k128_split(x, ldx, x2, ldx2, x2_period, w, bias, rows, n, epi, y, ldy, residual, ldr,
                                            gamma, beta, eps, stream);
    if (rc != -100) return rc;                                     // -100: operands not 16-byte aligned for its epilogue
  }
  const long long ntiles = (rows + 31) / 32;
  static const int wgs = [] { const char *e = getenv("PCT_LIN_WGS"); const int v = e ? atoi(e) : 3; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
  const long long cap = 256LL * wgs;                               // persistent: `wgs` workgroups per CU (LDS allows 4)
  const unsigned gx = (unsigned)(ntiles < cap ? ntiles : cap);
  const dim3 grid(gx, (unsigned)((n + 127) / 128)), block(LIN_BLOCK);
#define PCT_LIN(EPI_, X2_)                                                                                          \
  hipLaunchKernelGGL((linear_k128_kernel<EPI_, X2_>), grid, block, 0, stream, x, ldx, x2, ldx2, (int)x2_period, w, bias, rows, n, \
                     y, ldy, \
                     residual, ldr, gamma, beta, eps)
  if (x2) {
    if (epi == LIN_EPI_BIAS) PCT_LIN(LIN_EPI_BIAS, true);
    else if (epi == LIN_EPI_BIAS_RELU) PCT_LIN(LIN_EPI_BIAS_RELU, true);
    else PCT_LIN(LIN_EPI_RES_LN, true);
  } else {
    if (epi == LIN_EPI_BIAS) PCT_LIN(LIN_EPI_BIAS, false);
    else if (epi == LIN_EPI_BIAS_RELU) PCT_LIN(LIN_EPI_BIAS_RELU, false);
    else PCT_LIN(LIN_EPI_RES_LN, false);
  }
#undef PCT_LIN
  return (int)hipGetLastError();
}

}  // namespace pct
