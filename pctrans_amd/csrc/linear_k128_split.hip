// fp32-accurate skinny GEMM on the bf16 matrix cores of MI355X (gfx950): y = act(x · Wᵀ + b) and the fused
// y = LayerNorm(residual + x · Wᵀ + b) for the encoder's K = 128 projections -- same contract, operands, epilogues and
// reference lines as linear_k128.hip (ops/modules/ms_deform_attn.py:64-67,96-110; pixel_decoder/msdeformattn.py:100-131).
//
// Why: these GEMMs are bound by the fp32 MFMA rate (157 TFLOP/s peak; linear_k128.hip sustains 123), while the bf16
// MFMA rate is 16x that.  Every fp32 number is EXACTLY the sum of three bf16 numbers
//     a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)
// (round-to-nearest leaves a residual below half an ulp of the 8-bit piece, so each piece contributes 9 bits: 27 >= 24;
// both subtractions are exact in fp32).  A product a·w = sum_ij ai·wj is therefore evaluated from its six leading terms
//     a1w1 | a1w2 + a2w1 | a2w2 + a1w3 + a3w1          (relative size 1 | 2^-9 | 2^-18)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the dropped terms (a2w3, a3w2,
// a3w3) are below 2^-26 of the product, i.e. under a quarter of the rounding error fp32 itself commits on it.  The
// result is as accurate as an fp32 GEMM (measured against fp64 in tests/test_fused_ops_gpu.py) at 6/16 of its MFMA
// cycles.  Not a reduced-precision path: no input is rounded.
//
// Structure (as linear_k128.hip): a wave owns 32 output columns for its whole life, their W rows pre-split into
// 3 x 32 VGPRs in the B-operand layout; the 4 waves of a workgroup share a 32-row A tile that the workgroup fetches
// with coalesced dwordx4 buffer loads, splits once (11 VALU per element pair) and keeps as three bf16 planes in a
// double-buffered LDS image (row stride 272 B: every 16-lane group of a ds_read_b128 covers all 64 banks); per tile a
// wave issues 48 MFMAs -- the a1w1 chain into one accumulator, the five small terms into a second one, so the small
// terms are rounded at their own magnitude.  The MFMA computes the TRANSPOSED tile (W pieces as the A operand, x pieces
// as B): a lane then holds, for ONE row of x, four groups of four consecutive output columns, so the epilogue is
// dwordx4 all the way (bias/gamma/beta vectors, residual loads, stores -- 4 store instructions per lane and tile
// instead of 16; with dword stores the kernel was store-issue bound at ~60 cycles per wave store), and a LayerNorm row
// sum is 15 in-lane adds + one cross-half exchange + one LDS hop across the 4 waves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#ifndef PCT_LIN_PRIO
#define PCT_LIN_PRIO 0
#endif
#ifndef PCT_LIN_INTERLEAVE
#define PCT_LIN_INTERLEAVE 1        /* 16 x 16 form: the next tile's split / LDS stores between the MFMAs (linear1 4.17 -> 3.95 ms, same-box A/B) */
#endif
// output rows carry the non-temporal hint (aux bit 1 of the buffer store): written once, read by the NEXT kernel, far larger
// than the L2 -- merged projections 1.127 -> 1.102 ms at 1.39 M rows, three same-box alternations
constexpr int SPL_STORE_NT = 2;
#ifndef PCT_LIN_IL_DELAY
#define PCT_LIN_IL_DELAY 1          /* merged projections 1.117 -> 1.102 ms (same-box A/B) */
#endif
#ifndef PCT_LIN_STAGGER
#define PCT_LIN_STAGGER 0
#endif

namespace pct {

typedef float spl_f32x16 __attribute__((ext_vector_type(16)));
typedef float spl_f32x4 __attribute__((ext_vector_type(4)));
typedef float spl_f32x2 __attribute__((ext_vector_type(2)));
typedef int spl_i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 spl_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 spl_bf16x2 __attribute__((ext_vector_type(2)));

constexpr int SPL_K = 128;
constexpr int SPL_BLOCK = 256;
constexpr int SPL_ROWB = 272;                     // bytes per row of a bf16 plane (256 + 16 pad)
constexpr int SPL_PLANE = 32 * SPL_ROWB;          // one plane of a 32-row tile
constexpr int SPL_IMAGE = 3 * SPL_PLANE;          // three planes
constexpr int SPL_OLD = 36;                       // row stride (floats) of a wave's 32 x 32 output scratch

enum { SPL_EPI_BIAS = 0, SPL_EPI_BIAS_RELU = 1, SPL_EPI_RES_LN = 2 };

template <int CTRL>
__device__ __forceinline__ float spl_dpp(float v)
{
  const int i = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, true));
}

// sum over the 32 lanes of each wave half; result valid in every lane of the half
__device__ __forceinline__ float spl_half_sum(float v)
{
  v += spl_dpp<0xB1>(v);
  v += spl_dpp<0x4E>(v);
  v += spl_dpp<0x141>(v);
  v += spl_dpp<0x140>(v);
  const int vi = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
  return (threadIdx.x & 32) ? r2 + r3 : r0 + r1;
}

// (x, y) -> three packed bf16 pairs (low half = x's piece), x = x1 + x2 + x3 exactly
__device__ __forceinline__ void spl_split(const float x, const float y, unsigned &p1, unsigned &p2, unsigned &p3)
{
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(spl_f32x2{x, y}, spl_bf16x2));     // v_cvt_pk_bf16_f32 (RNE)
  const float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);   // exact
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(spl_f32x2{rx, ry}, spl_bf16x2));
  const float sx = rx - __uint_as_float(p2 << 16), sy = ry - __uint_as_float(p2 & 0xffff0000u); // exact
  p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(spl_f32x2{sx, sy}, spl_bf16x2));
}

// Up to SPL_MAXSEG Linear layers that read the same rows in ONE launch (segment s: y_s = x_s W_s^T + b_s with
// x_s = x or x + x2): their column slices are dealt to the same XCD like the slices of a single wide layer, so x is
// read from HBM once for all of them (value_proj, sampling_offsets and attention_weights of MSDeformAttn all read src).
constexpr int SPL_MAXSEG = 4;
struct SplSegs {
  const float *w[SPL_MAXSEG];
  const float *bias[SPL_MAXSEG];
  float *y[SPL_MAXSEG];
  long long ldy[SPL_MAXSEG];
  int n[SPL_MAXSEG];
  int add_x2[SPL_MAXSEG];
  int nseg;
};

// M16: the tile's products on v_mfma_f32_16x16x32_bf16 (four 16 x 16 sub-tiles, K = 32 per instruction) instead of
// v_mfma_f32_32x32x16_bf16 -- the same flops per clock and the same fragments' worth of registers, but on a chip that holds
// its clock down under matrix load the 16 x 16 shape is granted a higher clock (MI355X_MICROARCH.md, DVFS item 7: 1.12-1.15x
// in bare loops on random data).  Bias / ReLU epilogues only.
#ifndef PCT_LIN_M16
#define PCT_LIN_M16 1
#endif
typedef float spl_f32x4acc __attribute__((ext_vector_type(4)));

template <int EPI, bool HAS_X2, bool M16 = false>
__global__ __launch_bounds__(SPL_BLOCK, 2) void linear_k128_split_kernel(
    const float *__restrict__ X, const long long ldx, const float *__restrict__ X2, const long long ldx2,
    const int x2_period, const SplSegs segs, const long long M, const float *__restrict__ R, const long long ldr,
    const float *__restrict__ gamma, const float *__restrict__ beta, const float eps)
{
  __shared__ __attribute__((aligned(16))) unsigned char abuf[2][SPL_IMAGE];
  __shared__ float red[4][32];                     // LayerNorm: per-wave partial row sums
  __shared__ __attribute__((aligned(16))) float oscr[4][32 * SPL_OLD];   // per-wave output transpose

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // provably wave-uniform (descriptor selects below)
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

  const int r = lane & 31, h = lane >> 5;
  // 1-D grid of nslots x nslices workgroups (nslots a multiple of 8).  Hardware deals workgroup g to XCD g % 8: the
  // nslices column slices of one tile slot are given ids that differ by multiples of 8, so they run on the same XCD
  // at about the same time and the A tile they all read comes out of that XCD's L2 once (with a (tiles, slices) grid
  // the slices are whole passes apart and x is streamed from HBM once per 128 output columns).
  int nslices = 0;
#pragma unroll
  for (int g = 0; g < SPL_MAXSEG; ++g) nslices += g < segs.nseg ? (segs.n[g] + 127) / 128 : 0;
  const int nslots = gridDim.x / nslices;
  const int within = blockIdx.x >> 3;
  const int slot = (within / nslices) * 8 + (blockIdx.x & 7);
  // my segment and my slice inside it (uniform)
  int slice = within % nslices, seg = 0;
#pragma unroll
  for (int g = 0; g + 1 < SPL_MAXSEG; ++g) {
    const int cnt = (segs.n[g] + 127) / 128;
    if (g + 1 < segs.nseg && seg == g && slice >= cnt) {
      slice -= cnt;
      seg = g + 1;
    }
  }
  const float *__restrict__ W = segs.w[0];
  const float *__restrict__ bias = segs.bias[0];
  float *__restrict__ Y = segs.y[0];
  long long ldy = segs.ldy[0];
  int N = segs.n[0];
  bool add_x2 = segs.add_x2[0] != 0;
#pragma unroll
  for (int g = 1; g < SPL_MAXSEG; ++g)
    if (seg == g) {
      W = segs.w[g];
      bias = segs.bias[g];
      Y = segs.y[g];
      ldy = segs.ldy[g];
      N = segs.n[g];
      add_x2 = segs.add_x2[g] != 0;
    }
  const int col = slice * 128 + wave * 32 + r;               // my W row (A operand of the transposed product)
  const bool active = slice * 128 + wave * 32 < N;           // n % 32 == 0: wave-uniform

  // W pieces: step t of the MFMA loop covers k = 16t + 8h + 0..7 on both operands
  spl_i32x4 w1[8], w2[8], w3[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) w1[t] = w2[t] = w3[t] = spl_i32x4{0, 0, 0, 0};
  if (active) {
    // M16: piece t = 4 cw + s holds W row cbase + 16 cw + lane % 16, k = 32 s + 8 (lane / 16) .. + 7 (A operand of the 16 x 16 x 32 form)
    const float *wp = M16 ? W + (long long)(slice * 128 + wave * 32 + (lane & 15)) * SPL_K + 8 * (lane >> 4)
                          : W + (long long)col * SPL_K + 8 * h;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float *wt = M16 ? wp + (long long)(t >> 2) * 16 * SPL_K + 32 * (t & 3) : wp + 16 * t;
      const spl_f32x4 lo = *reinterpret_cast<const spl_f32x4 *>(wt);
      const spl_f32x4 hi = *reinterpret_cast<const spl_f32x4 *>(wt + 4);
      unsigned a, b, c;
      spl_split(lo[0], lo[1], a, b, c); w1[t][0] = (int)a; w2[t][0] = (int)b; w3[t][0] = (int)c;
      spl_split(lo[2], lo[3], a, b, c); w1[t][1] = (int)a; w2[t][1] = (int)b; w3[t][1] = (int)c;
      spl_split(hi[0], hi[1], a, b, c); w1[t][2] = (int)a; w2[t][2] = (int)b; w3[t][2] = (int)c;
      spl_split(hi[2], hi[3], a, b, c); w1[t][3] = (int)a; w2[t][3] = (int)b; w3[t][3] = (int)c;
    }
  }
  // accumulator layout of the transposed tile: acc[4q + i] = y[row r][column cbase + 8q + 4h + i]
  const int cbase = slice * 128 + wave * 32;
  // LayerNorm: gamma / beta live in LDS (32 fewer VGPRs; 8 ds_read_b128 per tile), visible after the first barrier
  __shared__ __attribute__((aligned(16))) float gbuf[2][128];
  if constexpr (EPI == SPL_EPI_RES_LN) {
    if (tid < 128) gbuf[0][tid] = gamma[tid];
    else gbuf[1][tid - 128] = beta[tid - 128];
  }
  spl_f32x4 bvec[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // (M16: a lane holds columns cbase + 16 cw + 4 (lane / 16) + 0..3 of its x row: bvec[cw])
    const int bc = M16 ? cbase + 16 * (q & 1) + 4 * (lane >> 4) : cbase + 8 * q + 4 * h;
    bvec[q] = (bias && active) ? *reinterpret_cast<const spl_f32x4 *>(bias + bc) : spl_f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const long long ntiles = (M + 31) / 32;
  // buffer descriptors re-based on the tile's first row: rows past the end load 0 and their stores are dropped
  auto tile_rsrc = [&](const float *base, const long long ld, const long long tile) {
    const long long row0 = tile * 32;
    const long long left = M - row0;                              // <= 0 past the end: an empty descriptor
    const unsigned bytes = (unsigned)((left < 32 ? (left < 0 ? 0 : left) : 32) * ld * 4);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + row0 * ld), 0, (int)bytes, 0x00020000);
  };
  // A tile = 32 rows x 128 floats = 1024 float4; thread tid moves float4 number tid + 256*q (row 8q + tid/32)
  const int g_voff = (int)((tid >> 5) * ldx * 4) + 16 * (tid & 31);
  const int g_step = (int)(8 * ldx * 4);
  const int s_off = (tid >> 5) * SPL_ROWB + 8 * (tid & 31);         // 4 bf16 = 8 B per float4 and plane
  const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(HAS_X2 ? X2 : X), 0,
                                                     (int)((long long)x2_period * ldx2 * 4), 0x00020000);
  auto fetch = [&](const long long tile, spl_i32x4 (&g)[4]) {
    const auto rs = tile_rsrc(X, ldx, tile);
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, g_voff, q * g_step, 0);
  };
  // the x2 rows (the positional term, shared by the batch: L2 hits) are fetched one tile ahead only: one register set
  auto fetch2 = [&](const long long tile, spl_i32x4 (&g2)[4]) {
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
  auto stash_q = [&](unsigned char *dst, const int q, const spl_i32x4 (&g)[4], const spl_i32x4 (&g2)[4]) {
    spl_f32x4 v = __builtin_bit_cast(spl_f32x4, g[q]);
    if constexpr (HAS_X2) {
      if (add_x2) v += __builtin_bit_cast(spl_f32x4, g2[q]);
    }
    unsigned a0, b0, c0, a1, b1, c1;
    spl_split(v[0], v[1], a0, b0, c0);
    spl_split(v[2], v[3], a1, b1, c1);
    unsigned char *p = dst + s_off + q * 8 * SPL_ROWB;
    *reinterpret_cast<uint2 *>(p) = make_uint2(a0, a1);
    *reinterpret_cast<uint2 *>(p + SPL_PLANE) = make_uint2(b0, b1);
    *reinterpret_cast<uint2 *>(p + 2 * SPL_PLANE) = make_uint2(c0, c1);
  };
  auto stash = [&](unsigned char *dst, const spl_i32x4 (&g)[4], const spl_i32x4 (&g2)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) stash_q(dst, q, g, g2);
  };

  // stores: lane -> row lane/8 (+8i), 16 B at column 4*(lane%8): a wave instruction writes 8 full 128-B lines
  const int y_voff = (int)(((lane >> 3) * ldy + cbase + 4 * (lane & 7)) * 4);
  const int y_step = (int)(8 * ldy * 4);
  const int r_voff = (int)((r * ldr + cbase + 4 * h) * 4);
  const int a_off = r * SPL_ROWB + 16 * h;

  // Two tiles of x are in flight per workgroup (register sets ga / gb): under load a fetch takes longer than one
  // tile's MFMAs, and with a single tile of prefetch the iteration time is the memory latency (measured 3.2 us per
  // tile whatever n).  Issue order inside an iteration is what the in-order vmcnt makes cheap: [residual rows of this
  // tile, x of tile+2] -> MFMAs -> split + stash of tile+1 (the oldest loads) -> epilogue (waits only for the residual,
  // which is older than the far fetch) -> stores.
  spl_i32x4 ga[4], gb[4], g2[4] = {};
  auto body = [&](const long long tile, const int buf, spl_i32x4 (&gn)[4], spl_i32x4 (&gf)[4]) {
    const long long far = tile + 2 * (long long)nslots;
    spl_i32x4 res[4];
    if constexpr (EPI == SPL_EPI_RES_LN) {
      const auto rr = tile_rsrc(R, ldr, tile);
#pragma unroll
      for (int q = 0; q < 4; ++q) res[q] = __builtin_amdgcn_raw_buffer_load_b128(rr, r_voff, 32 * q, 0);
    }
    fetch2(tile + nslots, g2);
    fetch(far, gf);                                               // past the end: empty descriptor, returns zeros
    __builtin_amdgcn_sched_barrier(0);                            // keep the fetches ahead of the MFMAs

    if constexpr (M16) {
      static_assert(EPI != SPL_EPI_RES_LN || !M16, "16 x 16 form: bias / ReLU epilogues only");
      // sub-tile (cw, rx): output columns cbase + 16 cw .., x rows 16 rx ..; acc[t] = y[row 16 rx + lane % 16][col cbase + 16 cw + 4 (lane / 16) + t]
      spl_f32x4acc ahi[2][2], alo[2][2];
#pragma unroll
      for (int cw = 0; cw < 2; ++cw)
#pragma unroll
        for (int rx = 0; rx < 2; ++rx) ahi[cw][rx] = alo[cw][rx] = spl_f32x4acc{0.f, 0.f, 0.f, 0.f};
      {
        const unsigned char *ap = abuf[buf] + (lane & 15) * SPL_ROWB + 16 * (lane >> 4);     // x row lane % 16 (+ 16 rx), k = 8 (lane / 16) ..
        spl_bf16x8 n1[2], n2[2], n3[2];
#pragma unroll
        for (int rx = 0; rx < 2; ++rx) {
          n1[rx] = *reinterpret_cast<const spl_bf16x8 *>(ap + rx * 16 * SPL_ROWB);
          n2[rx] = *reinterpret_cast<const spl_bf16x8 *>(ap + rx * 16 * SPL_ROWB + SPL_PLANE);
          n3[rx] = *reinterpret_cast<const spl_bf16x8 *>(ap + rx * 16 * SPL_ROWB + 2 * SPL_PLANE);
        }
#pragma unroll
        for (int sk = 0; sk < 4; ++sk) {
          spl_bf16x8 x1[2], x2p[2], x3[2];
#pragma unroll
          for (int rx = 0; rx < 2; ++rx) {
            x1[rx] = n1[rx];
            x2p[rx] = n2[rx];
            x3[rx] = n3[rx];
          }
          if (sk < 3) {
#pragma unroll
            for (int rx = 0; rx < 2; ++rx) {
              n1[rx] = *reinterpret_cast<const spl_bf16x8 *>(ap + rx * 16 * SPL_ROWB + 64 * (sk + 1));
              n2[rx] = *reinterpret_cast<const spl_bf16x8 *>(ap + rx * 16 * SPL_ROWB + 64 * (sk + 1) + SPL_PLANE);
              n3[rx] = *reinterpret_cast<const spl_bf16x8 *>(ap + rx * 16 * SPL_ROWB + 64 * (sk + 1) + 2 * SPL_PLANE);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#if PCT_LIN_INTERLEAVE
          // the NEXT tile's x, quarter sk: split into bf16 planes and stored into the other LDS image BETWEEN this step's MFMAs
          // (an MFMA holds the SIMD's vector issue for 8 of its 16 cycles: one vector instruction per MFMA rides along) instead
          // of in a block of ~100 vector instructions behind the last MFMA with the matrix pipe idle
          // (with a second operand x2 -- fetched for the next tile at the top of THIS tile, from L2 -- the first quarters wait
          // PCT_LIN_IL_DELAY steps for it and the last ones follow the loop)
          if (sk >= (HAS_X2 ? PCT_LIN_IL_DELAY : 0)) stash_q(abuf[buf ^ 1], sk - (HAS_X2 ? PCT_LIN_IL_DELAY : 0), gn, g2);
#endif
#pragma unroll
          for (int cw = 0; cw < 2; ++cw) {
            const spl_bf16x8 b1 = __builtin_bit_cast(spl_bf16x8, w1[4 * cw + sk]);
            const spl_bf16x8 b2 = __builtin_bit_cast(spl_bf16x8, w2[4 * cw + sk]);
            const spl_bf16x8 b3 = __builtin_bit_cast(spl_bf16x8, w3[4 * cw + sk]);
#pragma unroll
            for (int rx = 0; rx < 2; ++rx) {
              alo[cw][rx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, x3[rx], alo[cw][rx], 0, 0, 0);
              ahi[cw][rx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, x1[rx], ahi[cw][rx], 0, 0, 0);
              alo[cw][rx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b3, x1[rx], alo[cw][rx], 0, 0, 0);
              alo[cw][rx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, x2p[rx], alo[cw][rx], 0, 0, 0);
              alo[cw][rx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, x2p[rx], alo[cw][rx], 0, 0, 0);
              alo[cw][rx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, x1[rx], alo[cw][rx], 0, 0, 0);
            }
          }
#if PCT_LIN_INTERLEAVE
#pragma unroll
          for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);      // one vector instruction
          }
#endif
        }
      }
#if !PCT_LIN_INTERLEAVE
      stash(abuf[buf ^ 1], gn, g2);
#else
#pragma unroll
      for (int q = 4 - (HAS_X2 ? PCT_LIN_IL_DELAY : 0); q < 4; ++q) stash_q(abuf[buf ^ 1], q, gn, g2);
#endif
      {
        const auto ry = tile_rsrc(Y, ldy, active ? tile : ntiles);
        float *sw = oscr[wave];
#pragma unroll
        for (int cw = 0; cw < 2; ++cw)
#pragma unroll
          for (int rx = 0; rx < 2; ++rx) {
            spl_f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              v[t] = (ahi[cw][rx][t] + alo[cw][rx][t]) + bvec[cw][t];
              if constexpr (EPI == SPL_EPI_BIAS_RELU) v[t] = fmaxf(v[t], 0.f);
            }
            *reinterpret_cast<spl_f32x4 *>(sw + (16 * rx + (lane & 15)) * SPL_OLD + 16 * cw + 4 * (lane >> 4)) = v;
          }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const spl_f32x4 o = *reinterpret_cast<const spl_f32x4 *>(sw + ((lane >> 3) + 8 * i) * SPL_OLD + 4 * (lane & 7));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(spl_i32x4, o), ry, y_voff, i * y_step, SPL_STORE_NT);
        }
      }
    } else {
    spl_f32x16 acc_hi, acc_lo;                                    // a1w1 | the five small terms
#pragma unroll
    for (int j = 0; j < 16; ++j) acc_hi[j] = acc_lo[j] = 0.f;
    {                                                             // (a wave past n multiplies zeros: no branch, so
      const unsigned char *ap = abuf[buf] + a_off;                //  the compiler's vmcnt bookkeeping stays exact)
      // the three pieces of step t+1 are read while step t's MFMAs run (the wave issues in order: reads placed
      // after the MFMAs would leave the matrix pipe idle for an LDS round trip per step)
      spl_bf16x8 n1 = *reinterpret_cast<const spl_bf16x8 *>(ap);
      spl_bf16x8 n2 = *reinterpret_cast<const spl_bf16x8 *>(ap + SPL_PLANE);
      spl_bf16x8 n3 = *reinterpret_cast<const spl_bf16x8 *>(ap + 2 * SPL_PLANE);
#if PCT_LIN_PRIO
      __builtin_amdgcn_s_setprio(PCT_LIN_PRIO);                   // (the wave feeding the matrix pipe before its SIMD-mate's split / epilogue)
#endif
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const spl_bf16x8 a1 = n1, a2 = n2, a3 = n3;
        if (t < 7) {
          n1 = *reinterpret_cast<const spl_bf16x8 *>(ap + 32 * (t + 1));
          n2 = *reinterpret_cast<const spl_bf16x8 *>(ap + 32 * (t + 1) + SPL_PLANE);
          n3 = *reinterpret_cast<const spl_bf16x8 *>(ap + 32 * (t + 1) + 2 * SPL_PLANE);
        }
        __builtin_amdgcn_sched_barrier(0);
        const spl_bf16x8 b1 = __builtin_bit_cast(spl_bf16x8, w1[t]);
        const spl_bf16x8 b2 = __builtin_bit_cast(spl_bf16x8, w2[t]);
        const spl_bf16x8 b3 = __builtin_bit_cast(spl_bf16x8, w3[t]);
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a3, acc_lo, 0, 0, 0);
        acc_hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc_hi, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b3, a1, acc_lo, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a2, acc_lo, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a2, acc_lo, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a1, acc_lo, 0, 0, 0);
      }
    }
#if PCT_LIN_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    stash(abuf[buf ^ 1], gn, g2);

    {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = (acc_hi[j] + acc_lo[j]) + bvec[j / 4][j % 4];

      if constexpr (EPI == SPL_EPI_RES_LN) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] += __int_as_float(res[j / 4][j % 4]);
        // two-pass LayerNorm over the row's 128 columns: my 16, the other half of the wave (lane ^ 32), the 4 waves
        auto row_sum = [&](const float (&x)[16]) {
          float t = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7])) +
                    (((x[8] + x[9]) + (x[10] + x[11])) + ((x[12] + x[13]) + (x[14] + x[15])));
          t += __shfl_xor(t, 32);
          return t;
        };
        float part = row_sum(v);
        if (h == 0) red[wave][r] = part;
        __syncthreads();
        const float mean = ((red[0][r] + red[1][r]) + (red[2][r] + red[3][r])) * (1.f / 128.f);
        float sq[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          v[j] -= mean;
          sq[j] = v[j] * v[j];
        }
        part = row_sum(sq);
        __syncthreads();                                          // red free again
        if (h == 0) red[wave][r] = part;
        __syncthreads();
        const float rstd = rsqrtf(((red[0][r] + red[1][r]) + (red[2][r] + red[3][r])) * (1.f / 128.f) + eps);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const spl_f32x4 gq = *reinterpret_cast<const spl_f32x4 *>(&gbuf[0][wave * 32 + 8 * q + 4 * h]);
          const spl_f32x4 bq = *reinterpret_cast<const spl_f32x4 *>(&gbuf[1][wave * 32 + 8 * q + 4 * h]);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[4 * q + i] = v[4 * q + i] * rstd * gq[i] + bq[i];
        }
      } else if constexpr (EPI == SPL_EPI_BIAS_RELU) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      const auto ry = tile_rsrc(Y, ldy, active ? tile : ntiles);   // a wave past n stores nothing
      // through the wave's own LDS scratch (in-order DS ops of one wave: no barrier): the accumulator layout has a
      // row's 32 columns in 8 pieces over two lanes; stored directly, every instruction touched 32 lines for 32 B
      // each (measured slower than dword stores); re-read so that 8 lanes hold one row's 128 contiguous bytes
      float *sw = oscr[wave];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<spl_f32x4 *>(sw + r * SPL_OLD + 8 * q + 4 * h) = spl_f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const spl_f32x4 o = *reinterpret_cast<const spl_f32x4 *>(sw + ((lane >> 3) + 8 * i) * SPL_OLD + 4 * (lane & 7));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(spl_i32x4, o), ry, y_voff, i * y_step, SPL_STORE_NT);
      }
    }
    }
    __syncthreads();                  // next image complete; every wave is done reading this one
  };

  long long tile = slot;
  fetch(tile, ga);
  fetch2(tile, g2);
  stash(abuf[0], ga, g2);
  fetch(tile + nslots, ga);
  // the per-column constants are complete before the loop (otherwise their pending loads cost a vmcnt(0) per tile)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    asm volatile("" : "+v"(bvec[q]));
  }
  __syncthreads();
  for (; tile < ntiles; tile += 2 * (long long)nslots) {
    body(tile, 0, ga, gb);
    if (tile + nslots < ntiles) body(tile + nslots, 1, gb, ga);
  }
}

static int spl_launch(const float *x, long long ldx, const float *x2, long long ldx2, long long x2_period, const SplSegs &segs,
                      long long rows, int epi, const float *residual, long long ldr, const float *gamma, const float *beta,
                      float eps, hipStream_t stream)
{
  const long long ntiles = (rows + 31) / 32;
  static const int wgs = [] { const char *e = getenv("PCT_LIN_WGS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 2 ? 2 : v); }();
  int nslices = 0;
  for (int g = 0; g < segs.nseg; ++g) nslices += (segs.n[g] + 127) / 128;
  // persistent: 2 workgroups per CU (VGPRs) when one slice covers n, else tile slots x slices (see the kernel)
  long long nslots = 256LL * wgs / (nslices < 8 ? nslices : 8) / 8 * 8;      // all resident at once
  if (nslots > (ntiles + 7) / 8 * 8) nslots = (ntiles + 7) / 8 * 8;
  if (nslots < 8) nslots = 8;
  const dim3 grid((unsigned)(nslots * nslices)), block(SPL_BLOCK);
  static const bool m16 = [] { const char *e = getenv("PCT_LIN_M16"); return e ? e[0] != '0' : (PCT_LIN_M16 != 0); }();
#define PCT_SPL(EPI_, X2_)                                                                                           \
  do {                                                                                                               \
    if (m16 && EPI_ != SPL_EPI_RES_LN)                                                                               \
      hipLaunchKernelGGL((linear_k128_split_kernel<EPI_, X2_, EPI_ != SPL_EPI_RES_LN>), grid, block, 0, stream, x, ldx, x2, ldx2, \
                         (int)x2_period, segs, rows, residual, ldr, gamma, beta, eps);                               \
    else                                                                                                             \
      hipLaunchKernelGGL((linear_k128_split_kernel<EPI_, X2_, false>), grid, block, 0, stream, x, ldx, x2, ldx2,      \
                         (int)x2_period, segs, rows, residual, ldr, gamma, beta, eps);                               \
  } while (0)
  if (x2) {
    if (epi == SPL_EPI_BIAS) PCT_SPL(SPL_EPI_BIAS, true);
    else if (epi == SPL_EPI_BIAS_RELU) PCT_SPL(SPL_EPI_BIAS_RELU, true);
    else return -100;                                              // not instantiated (nothing uses it; it would spill)
  } else {
    if (epi == SPL_EPI_BIAS) PCT_SPL(SPL_EPI_BIAS, false);
    else if (epi == SPL_EPI_BIAS_RELU) PCT_SPL(SPL_EPI_BIAS_RELU, false);
    else PCT_SPL(SPL_EPI_RES_LN, false);
  }
#undef PCT_SPL
  return (int)hipGetLastError();
}

// n must be a multiple of 32 (LayerNorm variant: exactly 128); x rows 16-byte aligned.  Returns -100 when the
// operands do not allow the dwordx4 epilogue (caller uses the fp32-MFMA kernel).
int launch_linear_k128_split(const float *x, long long ldx, const float *x2, long long ldx2, long long x2_period,
                             const float *w, const float *bias, long long rows, int n, int epi, float *y, long long ldy,
                             const float *residual, long long ldr, const float *gamma, const float *beta, float eps,
                             hipStream_t stream)
{
  if (rows <= 0) return 0;
  if ((((uintptr_t)y | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)gamma | (uintptr_t)beta) & 15u) || (ldy & 3) ||
      (ldr & 3))
    return -100;
  SplSegs segs = {};
  segs.w[0] = w;
  segs.bias[0] = bias;
  segs.y[0] = y;
  segs.ldy[0] = ldy;
  segs.n[0] = n;
  segs.add_x2[0] = x2 ? 1 : 0;
  segs.nseg = 1;
  return spl_launch(x, ldx, x2, ldx2, x2_period, segs, rows, epi, residual, ldr, gamma, beta, eps, stream);
}

// several Linear layers over the same rows (bias epilogue only); every n[s] a multiple of 32, 1 <= nseg <= SPL_MAXSEG;
// use_add[s] != 0: segment s multiplies x + x2.  Returns -100 when an operand is not 16-byte aligned.
int launch_linear_k128_split_multi(const float *x, long long ldx, const float *x2, long long ldx2, long long x2_period,
                                   int nseg, const float *const *w, const float *const *bias, const int *n,
                                   const int *use_add, float *const *y, const long long *ldy, long long rows,
                                   hipStream_t stream)
{
  if (rows <= 0) return 0;
  if (nseg < 1 || nseg > SPL_MAXSEG) return -4;
  SplSegs segs = {};
  segs.nseg = nseg;
  bool any_add = false;
  for (int g = 0; g < nseg; ++g) {
    if ((((uintptr_t)y[g] | (uintptr_t)bias[g] | (uintptr_t)w[g]) & 15u) || (ldy[g] & 3)) return -100;
    segs.w[g] = w[g];
    segs.bias[g] = bias[g];
    segs.y[g] = y[g];
    segs.ldy[g] = ldy[g];
    segs.n[g] = n[g];
    segs.add_x2[g] = (use_add[g] && x2) ? 1 : 0;
    any_add |= segs.add_x2[g] != 0;
  }
  return spl_launch(x, ldx, any_add ? x2 : nullptr, ldx2, x2_period, segs, rows, SPL_EPI_BIAS, nullptr, 0, nullptr, nullptr,
                    0.f, stream);
}

}  // namespace pct
