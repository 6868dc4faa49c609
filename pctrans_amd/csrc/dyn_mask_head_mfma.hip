// Per-query dynamic mask head on CDNA4 MFMA (gfx950, wave64) for the bf16-autocast configuration, in two kernels:
//
//   A  dmh_logits_mfma_kernel : the 3-layer per-query MLP  (mask2former_transformer_decoder.py:699-719)
//        x0 = relu(W0[q] . [rel_x, rel_y, feat] + b0[q]),  x1 = relu(W1[q] . x0 + b1[q]),  logit = W2[q] . x1 + b2[q]
//      as chained v_mfma_f32_16x16x16_bf16.  One MFMA tile = 2 queries x 8 hidden rows (M = 16) by 16 pixels (N = 16);
//      layer 0 contracts over the 16 feature channels (K = 16 exactly), layer 1 over the 16 hidden rows of the query
//      pair with a block-diagonal W1.  The accumulator layout of layer 0 (lane (g, px) holds rows 4g..4g+3) IS the
//      B-operand layout of a K = 16 MFMA (lane (g, px) supplies k = 4g..4g+3), so x0 goes from accumulator registers to
//      operand registers with a relu and a bf16 pack -- no LDS, no shuffles.  The two relative-coordinate inputs and
//      the biases enter as the fp32 initial accumulator (more accurate than the reference's bf16 conv inputs), layer 2
//      (8 MACs) is a 4-FMA partial dot per lane plus one cross-group add.  Feature fragments of a wave's 8 pixel tiles
//      stay in registers across all query pairs; per-pair weights are fetched once per pair.
//   B  dmh_resize_kernel : bilinear x2 upsample (align_corners = False) of the bf16 logit planes + the boolean
//      attention mask  sigmoid(resize to (th, tw)) < 0.5   (:689-695), a pure streaming pass through an LDS row band.
//
// Under autocast the reference's convolutions take bf16 inputs/weights with fp32 accumulation and emit bf16; here the
// hidden activations are likewise rounded to bf16 between layers 0 and 1 and the logits are stored as bf16.
#include "dmh_common.hpp"

namespace pct {

constexpr int DMM_PT = 8;        // pixel tiles (16 px each) per wave
constexpr int DMM_BLOCK = 256;

// feat [N, 16, H, W] fp32, ref [N, Q, 2], params [N, Q, G] fp32 (G = 233 with rel coords, 217 without),
// logits [N, Q, H*W] bf16
template <bool REL>
__global__ __launch_bounds__(DMM_BLOCK) void dmh_logits_mfma_kernel(const float *__restrict__ feat,
                                                                    const float *__restrict__ ref,
                                                                    const float *__restrict__ params, const int Q,
                                                                    const int H, const int W, const int stride,
                                                                    const int blocks_per_image,
                                                                    __bf16 *__restrict__ logits)
{
  constexpr int C = DMH_C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, g = lane >> 4;                    // MFMA column (pixel / weight row) and lane group
  const int HW = H * W;
  const int n = blockIdx.x / blocks_per_image;
  const int pb = blockIdx.x - n * blocks_per_image;
  const int px_base = (pb * (DMM_BLOCK / 64) + wave) * (DMM_PT * 16);

  // ---- B operands of layer 0 for my pixel column of each tile (kept for all query pairs; dmh_common.hpp) -------------
  dm_u32x4 fb[DMM_PT];
  const float *fimg = feat + (size_t)n * C * HW;
#pragma unroll
  for (int t = 0; t < DMM_PT; ++t) {
    const int px = min(px_base + col * DMM_PT + t, HW - 1);        // lane `col` owns DMM_PT CONSECUTIVE pixels (one 16-B store)
    fb[t] = dmh_pixel_operand<REL>(fimg, HW, W, px, g, stride);
  }

  const int npairs = (Q + 1) / 2;
  for (int pr = 0; pr < npairs; ++pr) {
    DmhPair<REL> pw;
    pw.load(params, ref, n, Q, pr, col, g, H, W, stride);
    __bf16 *lrow = logits + ((size_t)n * Q + min(2 * pr + (g >> 1), Q - 1)) * HW;
    const bool writer = (g & 1) == 0 && 2 * pr + (g >> 1) < Q;   // groups 0 / 2 hold the reduced logit of q0 / q1

    float outv[DMM_PT];
    pw.template tiles<4>(fb, outv);
    pw.template tiles<4>(fb + 4, outv + 4);
    if (writer) {                                                // the lane's 8 consecutive pixels: one 16-byte store
      const int px0 = px_base + col * DMM_PT;
      if (px0 + DMM_PT <= HW && ((HW & 7) == 0)) {
        typedef __bf16 dm_bf16x8 __attribute__((ext_vector_type(8)));
        dm_bf16x8 o;
#pragma unroll
        for (int t = 0; t < DMM_PT; ++t) o[t] = (__bf16)outv[t];
        *reinterpret_cast<dm_bf16x8 *>(lrow + px0) = o;
      } else {
#pragma unroll
        for (int t = 0; t < DMM_PT; ++t)
          if (px0 + t < HW) lrow[px0 + t] = (__bf16)outv[t];
      }
    }
  }
}

// logits [N*Q, H, W] bf16 -> up [N*Q, 2H, 2W] bf16, amask [N*Q, th*tw] bytes
__global__ __launch_bounds__(256) void dmh_resize_kernel(const __bf16 *__restrict__ logits, const int H, const int W,
                                                         const int TR, const int nbands, const int th, const int tw,
                                                         __bf16 *__restrict__ up, unsigned char *__restrict__ amask)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *tile = reinterpret_cast<float *>(smem_raw);
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int nq = (int)(lb / (unsigned)nbands);
  const int band = (int)(lb - (unsigned)nq * nbands);
  const int r0 = band * TR, r1 = min(r0 + TR, H);
  const int lo = max(r0 - 1, 0), hi = min(r1, H - 1);
  const int npx = (hi - lo + 1) * W;
  const __bf16 *src = logits + (size_t)nq * H * W + (size_t)lo * W;
  if ((W & 7) == 0) {        // 16-byte loads: rows are multiples of 8 bf16 and the band starts on a row boundary
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    for (int i = threadIdx.x * 8; i < npx; i += 256 * 8) {
      const bf16x8_t v = *reinterpret_cast<const bf16x8_t *>(src + i);
#pragma unroll
      for (int k = 0; k < 8; ++k) tile[i + k] = (float)v[k];
    }
  } else {
    for (int i = threadIdx.x; i < npx; i += 256) tile[i] = (float)src[i];
  }
  __syncthreads();

  {
    const int OW = 2 * W, OH = 2 * H;
    __bf16 *uplane = up + (size_t)nq * OH * OW;
    const int orow0 = 2 * r0, orows = 2 * (r1 - r0);
    if ((W & 3) == 0) {
      // exact-x2 fast path: a lane takes 4 input columns of one input row y and emits the 8 outputs of BOTH output rows
      // 2y and 2y+1 (two 16-byte stores).  At scale 1/2 PyTorch's source index is dst/2 - 1/4: weights are exactly
      // (1/4, 3/4) / (3/4, 1/4), clamped at the borders (lambda = 0 at the low edge, replicated index at the high
      // edge), so the generic formula below is evaluated with constant weights -- bit-identical to the generic path.
      typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
      const int cpr = W / 4;                                        // 4-column groups per row
      const int nrows_b = r1 - r0;
      for (int i = threadIdx.x; i < nrows_b * cpr; i += 256) {
        const int ry = i / cpr, cg = i - ry * cpr;
        const int y = r0 + ry, x0 = cg * 4;
        const float *rm = tile + (max(y - 1, 0) - lo) * W;         // rows y-1, y, y+1 (clamped like the index clamp)
        const float *rc = tile + (y - lo) * W;
        const float *rp = tile + (min(y + 1, H - 1) - lo) * W;
        float m[6], c[6], p[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int x = min(max(x0 - 1 + k, 0), W - 1);
          m[k] = rm[x];
          c[k] = rc[x];
          p[k] = rp[x];
        }
        // output row 2y: source rows (y-1, y), weights (1/4, 3/4); y = 0: source rows (0, 1), weights (1, 0).
        // output row 2y+1: source rows (y, y+1), weights (3/4, 1/4).  Register selects only (no runtime-indexed arrays).
        const bool top = y == 0;
        const float hA0 = top ? 1.f : 0.25f, hA1 = top ? 0.f : 0.75f;
        float a0[6], a1[6];                                         // the two source rows of output row 2y
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          a0[k] = top ? c[k] : m[k];
          a1[k] = top ? p[k] : c[k];
        }
        bf16x8_t oa, ob;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // even output column 2xx: source cols (xx-1, xx) = entries (k, k+1), weights (1/4, 3/4);
          // xx = 0: source cols (0, 1) = entries (k+1, k+2), weights (1, 0)
          const bool left = x0 + k == 0;
          const float wl0 = left ? 1.f : 0.25f, wl1 = left ? 0.f : 0.75f;
          const float a0l = left ? a0[k + 1] : a0[k], a0r = left ? a0[k + 2] : a0[k + 1];
          const float a1l = left ? a1[k + 1] : a1[k], a1r = left ? a1[k + 2] : a1[k + 1];
          const float c_l = left ? c[k + 1] : c[k], c_r = left ? c[k + 2] : c[k + 1];
          const float p_l = left ? p[k + 1] : p[k], p_r = left ? p[k + 2] : p[k + 1];
          oa[2 * k] = (__bf16)dm_blend(hA0, hA1, wl0, wl1, a0l, a0r, a1l, a1r);
          ob[2 * k] = (__bf16)dm_blend(0.75f, 0.25f, wl0, wl1, c_l, c_r, p_l, p_r);
          // odd output column 2xx+1: source cols (xx, xx+1) = entries (k+1, k+2), weights (3/4, 1/4)
          oa[2 * k + 1] = (__bf16)dm_blend(hA0, hA1, 0.75f, 0.25f, a0[k + 1], a0[k + 2], a1[k + 1], a1[k + 2]);
          ob[2 * k + 1] = (__bf16)dm_blend(0.75f, 0.25f, 0.75f, 0.25f, c[k + 1], c[k + 2], p[k + 1], p[k + 2]);
        }
        *reinterpret_cast<bf16x8_t *>(uplane + (size_t)(2 * y) * OW + 2 * x0) = oa;
        *reinterpret_cast<bf16x8_t *>(uplane + (size_t)(2 * y + 1) * OW + 2 * x0) = ob;
      }
    } else {
      for (int i = threadIdx.x; i < orows * OW; i += 256) {
        const int orr = i / OW, ox = i - orr * OW;
        const int oy = orow0 + orr;
        int y0, y1, x0, x1;
        float ly, lxx;
        dm_bilinear_src(oy, 0.5f, H, y0, y1, ly);
        dm_bilinear_src(ox, 0.5f, W, x0, x1, lxx);
        const float *ra = tile + (y0 - lo) * W, *rb = tile + (y1 - lo) * W;
        uplane[(size_t)oy * OW + ox] = (__bf16)dm_blend(1.f - ly, ly, 1.f - lxx, lxx, ra[x0], ra[x1], rb[x0], rb[x1]);
      }
    }
  }
  {
    const float sh = (float)H / (float)th, sw = (float)W / (float)tw;
    unsigned char *mplane = amask + (size_t)nq * th * tw;
    // only target rows whose upper source row can fall into [r0, r1): ty in [(r0+0.5)/sh - 1.5, (r1+0.5)/sh + 0.5]
    const int ty_lo = max(0, (int)(((float)r0 + 0.5f) / sh - 1.5f));
    const int ty_hi = min(th, (int)(((float)r1 + 0.5f) / sh + 0.5f) + 1);
    for (int i = ty_lo * tw + threadIdx.x; i < ty_hi * tw; i += 256) {
      const int ty = i / tw, tx = i - ty * tw;
      int y0, y1, x0, x1;
      float ly, lxx;
      dm_bilinear_src(ty, sh, H, y0, y1, ly);
      if (y0 < r0 || y0 >= r1) continue;
      dm_bilinear_src(tx, sw, W, x0, x1, lxx);
      const float *ra = tile + (y0 - lo) * W, *rb = tile + (y1 - lo) * W;
      mplane[i] = dm_mask_byte(ly, lxx, ra[x0], ra[x1], rb[x0], rb[x1]);
    }
  }
}

// `scratch` : [N, Q, H*W] bf16 workspace for the logits (caller-allocated; no allocation inside the launch path)
int launch_dyn_mask_head_mfma(const float *feat, const float *ref, const float *params, int N, int C, int Q, int H,
                              int W, int stride, int rel_coord, int th, int tw, void *scratch, void *up,
                              unsigned char *amask, hipStream_t stream)
{
  if (C != 16) return -4;
  if ((long long)N * Q == 0 || H == 0 || W == 0) return 0;
  const int HW = H * W;
  const int px_per_block = (DMM_BLOCK / 64) * DMM_PT * 16;
  const int bpi = (HW + px_per_block - 1) / px_per_block;
  __bf16 *lg = static_cast<__bf16 *>(scratch);
  if (rel_coord)
    hipLaunchKernelGGL((dmh_logits_mfma_kernel<true>), dim3((unsigned)(N * bpi)), dim3(DMM_BLOCK), 0, stream, feat, ref,
                       params, Q, H, W, stride, bpi, lg);
  else
    hipLaunchKernelGGL((dmh_logits_mfma_kernel<false>), dim3((unsigned)(N * bpi)), dim3(DMM_BLOCK), 0, stream, feat, ref,
                       params, Q, H, W, stride, bpi, lg);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;

  int TR = H;
  const int max_rows = (16 * 1024) / (W * 4) - 2;                 // <= 16 KB of LDS per block: many blocks per CU
  if (max_rows < 1) return -4;
  if (TR > max_rows) TR = max_rows;
  while (TR > 8 && (long long)N * Q * ((H + TR - 1) / TR) < 4096) TR = (TR + 1) / 2;
  const int nbands = (H + TR - 1) / TR;
  const long long nblk = (long long)N * Q * nbands;
  if (nblk > 0x7fffffffLL) return -4;
  hipLaunchKernelGGL(dmh_resize_kernel, dim3((unsigned)nblk), dim3(256), (size_t)(TR + 2) * W * sizeof(float), stream, lg,
                     H, W, TR, nbands, th, tw, static_cast<__bf16 *>(up), amask);
  return (int)hipGetLastError();
}

}  // namespace pct
