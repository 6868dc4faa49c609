// Fused per-query dynamic mask head for MI355X (gfx950, wave64).
//
// Replaces, per call, the reference's dynamic_mask_with_coords + mask_heads_forward
// (transformer_decoder/mask2former_transformer_decoder.py:647-719): for every (image n, query q)
//     x0 = relu(W0[q] . [rel_x, rel_y, feat(0..C-1)] + b0[q])      8 x (C+2)
//     x1 = relu(W1[q] . x0 + b1[q])                                 8 x 8
//     logit = W2[q] . x1 + b2[q]                                    1 x 8
// with rel = ref[q] * (W*stride, H*stride) - (x*stride + stride/2, y*stride + stride/2)   (:654-670, :929-943),
// then   up   = bilinear x2 (align_corners=False) of the logit plane                        (:693-695)
//        mask = sigmoid(bilinear resize of the logit plane to (th, tw)) < 0.5               (:689-691)
// The reference materialises a [1, N*Q*(C+2), H, W] input tensor (118 MB / image at Q=100, 128x128) and runs three
// grouped convolutions with N*Q groups; here one block owns a band of rows of ONE (n, q) plane: the 233 generated
// parameters are block-uniform (scalar loads), each lane evaluates the 216-MAC MLP for 4 pixels at a time from
// coalesced feature loads, the logit band (+1 halo row each side) lives only in LDS, and the block writes the
// x2-upsampled rows with 16-byte stores plus the attention-mask rows it owns.  HBM traffic per call is the output
// itself plus one L2-resident read of the feature map.
//
// Numerics: fp32 throughout (>= the reference under fp32; under bf16 autocast the reference's convs round their
// inputs/outputs to bf16 -- with OutT = bf16 the logits are rounded to bf16 where the reference's conv output is,
// and the sigmoid threshold is evaluated on bf16-rounded values like torch's bf16 kernels do).
#include "msda_common.hpp"

namespace pct {

constexpr int DMH_BLOCK = 256;
constexpr int DMH_HID = 8;     // dynamic_mask_channels (mask2former_transformer_decoder.py:419)
constexpr int DMH_PX = 4;      // pixels per lane per iteration

__device__ __forceinline__ float bf16_round(float v) { return (float)(__bf16)v; }

// PyTorch upsample_bilinear2d source index (align_corners = False): src = scale * (dst + 0.5) - 0.5, clamped at 0
__device__ __forceinline__ void bilinear_src(int dst, float scale, int in_size, int &i0, int &i1, float &lam)
{
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  lam = src - (float)i0;
}

template <int C, bool REL, typename OutT>
__global__ __launch_bounds__(DMH_BLOCK) void dyn_mask_head_kernel(
    const float *__restrict__ feat,    // [N, C, H, W]
    const float *__restrict__ ref,     // [N, Q, 2]  normalised (x, y)
    const float *__restrict__ params,  // [N, Q, G]  w0 | w1 | w2 | b0 | b1 | b2   (parse_dynamic_params order)
    const int Q, const int H, const int W, const int stride, const int TR, const int nbands, const int th,
    const int tw, OutT *__restrict__ up,       // [N, Q, 2H, 2W]
    unsigned char *__restrict__ amask)           // [N, Q, th*tw]   1 = may not attend
{
  constexpr int CIN = REL ? C + 2 : C;
  constexpr int G = CIN * DMH_HID + DMH_HID * DMH_HID + DMH_HID + DMH_HID + DMH_HID + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *tile = reinterpret_cast<float *>(smem_raw);       // logits of rows [lo, hi], row-major, W per row

  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int nq = (int)(lb / (unsigned)nbands);                // n * Q + q
  const int band = (int)(lb - (unsigned)nq * nbands);
  const int n = nq / Q;
  const int r0 = band * TR;
  const int r1 = min(r0 + TR, H);                             // band rows [r0, r1)
  const int lo = max(r0 - 1, 0), hi = min(r1, H - 1);         // rows held in LDS [lo, hi]
  const int nrows = hi - lo + 1;
  const int HW = H * W;

  const float *pq = params + (size_t)nq * G;                  // block-uniform -> scalar loads
  const float *fimg = feat + (size_t)n * C * HW;

  float rx = 0.f, ry = 0.f;
  if constexpr (REL) {
    rx = ref[(size_t)nq * 2] * (float)(W * stride);
    ry = ref[(size_t)nq * 2 + 1] * (float)(H * stride);
  }
  const float half = (float)(stride / 2);

  // ---- phase 1: logits of the band (+ halo rows) into LDS ---------------------------------------------------
  const int npx = nrows * W;
  for (int base = 0; base < npx; base += DMH_BLOCK * DMH_PX) {
    // The 233 parameters are loop-invariant; hoisted out of the loop they exceed the SGPR file and get spilled to
    // VGPR lanes (one v_readlane per use).  Launder the pointer so they are re-fetched from the scalar cache per
    // iteration, layer by layer, straight into SGPR operands of the FMAs.
    // The constant address space makes the uniform loads scalar (s_load) even though LDS/global stores precede
    // them in the loop (the kernel never writes `params`).
    typedef const __attribute__((address_space(4))) float *cfloat_p;
    unsigned long long praw = (unsigned long long)pq;
    asm volatile("" : "+s"(praw));
    cfloat_p p = (cfloat_p)praw;
    cfloat_p w0 = p, w1 = p + CIN * DMH_HID, w2 = w1 + DMH_HID * DMH_HID;
    cfloat_p b0 = w2 + DMH_HID, b1 = b0 + DMH_HID, b2 = b1 + DMH_HID;
    float f[DMH_PX][C];
    float relx[DMH_PX], rely[DMH_PX];
    int idx[DMH_PX];
#pragma unroll
    for (int j = 0; j < DMH_PX; ++j) {
      const int i = base + j * DMH_BLOCK + (int)threadIdx.x;
      idx[j] = i;
      const int ic = i < npx ? i : npx - 1;                    // clamp: tail lanes recompute the last pixel
      const int y = lo + ic / W, x = ic - (ic / W) * W;
      const float *fp = fimg + (size_t)y * W + x;
#pragma unroll
      for (int c = 0; c < C; ++c) f[j][c] = fp[(size_t)c * HW];
      relx[j] = rx - ((float)(x * stride) + half);
      rely[j] = ry - ((float)(y * stride) + half);
    }
    float h0[DMH_PX][DMH_HID];
#pragma unroll
    for (int k = 0; k < DMH_HID; ++k) {
      cfloat_p wk = w0 + k * CIN;
      const float bk = b0[k];
#pragma unroll
      for (int j = 0; j < DMH_PX; ++j) {
        float a = bk;
        if constexpr (REL) {
          a = fmaf(wk[0], relx[j], a);
          a = fmaf(wk[1], rely[j], a);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) a = fmaf(wk[(REL ? 2 : 0) + c], f[j][c], a);
        h0[j][k] = fmaxf(a, 0.f);
      }
    }
    float h1[DMH_PX][DMH_HID];
#pragma unroll
    for (int k = 0; k < DMH_HID; ++k) {
      cfloat_p wk = w1 + k * DMH_HID;
      const float bk = b1[k];
#pragma unroll
      for (int j = 0; j < DMH_PX; ++j) {
        float a = bk;
#pragma unroll
        for (int c = 0; c < DMH_HID; ++c) a = fmaf(wk[c], h0[j][c], a);
        h1[j][k] = fmaxf(a, 0.f);
      }
    }
    const float bo = b2[0];
#pragma unroll
    for (int j = 0; j < DMH_PX; ++j) {
      float a = bo;
#pragma unroll
      for (int c = 0; c < DMH_HID; ++c) a = fmaf(w2[c], h1[j][c], a);
      if constexpr (sizeof(OutT) == 2) a = bf16_round(a);       // the reference's conv output is bf16 under autocast
      tile[idx[j] < npx ? idx[j] : npx] = a;                     // tail lanes hit the dummy slot after the tile
    }
  }
  __syncthreads();

  // ---- phase 2: x2 bilinear upsample of the band's rows, 4 output pixels (16 B fp32 / 8 B bf16) per lane -------
  {
    const int OW = 2 * W, OH = 2 * H;
    OutT *uplane = up + (size_t)nq * OH * OW;
    const int orow0 = 2 * r0, orows = 2 * (r1 - r0);
    const int qpr = OW / 4;                                     // 4-pixel groups per output row (W even)
    const bool vec_ok = (OW % 4) == 0;
    if (vec_ok) {
      for (int i = threadIdx.x; i < orows * qpr; i += DMH_BLOCK) {
        const int orr = i / qpr, oq = i - orr * qpr;
        const int oy = orow0 + orr;
        int y0, y1;
        float ly;
        bilinear_src(oy, 0.5f, H, y0, y1, ly);
        const float *ra = tile + (y0 - lo) * W, *rb = tile + (y1 - lo) * W;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int x0, x1;
          float lx;
          bilinear_src(oq * 4 + k, 0.5f, W, x0, x1, lx);
          o[k] = (1.f - ly) * ((1.f - lx) * ra[x0] + lx * ra[x1]) + ly * ((1.f - lx) * rb[x0] + lx * rb[x1]);
        }
        OutT *dst = uplane + (size_t)oy * OW + oq * 4;
        if constexpr (sizeof(OutT) == 4) {
          *reinterpret_cast<vec_t<float, 4> *>(dst) = vec_t<float, 4>{o[0], o[1], o[2], o[3]};
        } else {
          *reinterpret_cast<vec_t<__bf16, 4> *>(dst) =
              vec_t<__bf16, 4>{(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
        }
      }
    } else {
      for (int i = threadIdx.x; i < orows * OW; i += DMH_BLOCK) {
        const int orr = i / OW, ox = i - orr * OW;
        const int oy = orow0 + orr;
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_src(oy, 0.5f, H, y0, y1, ly);
        bilinear_src(ox, 0.5f, W, x0, x1, lx);
        const float *ra = tile + (y0 - lo) * W, *rb = tile + (y1 - lo) * W;
        const float o = (1.f - ly) * ((1.f - lx) * ra[x0] + lx * ra[x1]) + ly * ((1.f - lx) * rb[x0] + lx * rb[x1]);
        uplane[(size_t)oy * OW + ox] = (OutT)o;
      }
    }
  }

  // ---- phase 3: attention-mask rows whose upper source row lies in this band ---------------------------------
  {
    const float sh = (float)H / (float)th, sw = (float)W / (float)tw;
    unsigned char *mplane = amask + (size_t)nq * th * tw;
    for (int i = threadIdx.x; i < th * tw; i += DMH_BLOCK) {
      const int ty = i / tw, tx = i - ty * tw;
      int y0, y1, x0, x1;
      float ly, lx;
      bilinear_src(ty, sh, H, y0, y1, ly);
      if (y0 < r0 || y0 >= r1) continue;
      bilinear_src(tx, sw, W, x0, x1, lx);
      const float *ra = tile + (y0 - lo) * W, *rb = tile + (y1 - lo) * W;
      float v = (1.f - ly) * ((1.f - lx) * ra[x0] + lx * ra[x1]) + ly * ((1.f - lx) * rb[x0] + lx * rb[x1]);
      float s;
      if constexpr (sizeof(OutT) == 2) {
        v = bf16_round(v);
        s = bf16_round(1.f / (1.f + expf(-v)));
      } else {
        s = 1.f / (1.f + expf(-v));
      }
      mplane[i] = s < 0.5f ? 1 : 0;
    }
  }
}

// out_dtype: 0 = f32, 2 = bf16
int launch_dyn_mask_head(const float *feat, const float *ref, const float *params, int N, int C, int Q, int H, int W,
                         int stride, int rel_coord, int th, int tw, int out_dtype, void *up, unsigned char *amask,
                         hipStream_t stream)
{
  if (C != 16) return -4;
  if ((long long)N * Q == 0 || H == 0 || W == 0) return 0;
  // rows per band: enough blocks to fill the chip several times over, LDS tile <= 48 KB
  int TR = H;
  const int max_rows = (48 * 1024 - 16) / (W * 4) - 2;
  if (max_rows < 1) return -4;
  if (TR > max_rows) TR = max_rows;
  while (TR > 8 && (long long)N * Q * ((H + TR - 1) / TR) < 4096) TR = (TR + 1) / 2;
  const int nbands = (H + TR - 1) / TR;
  const long long nblk = (long long)N * Q * nbands;
  if (nblk > 0x7fffffffLL) return -4;
  const size_t lds = ((size_t)(TR + 2) * W + 4) * sizeof(float);   // + dummy slot for tail lanes
  const dim3 grid((unsigned)nblk), block(DMH_BLOCK);
#define PCT_DMH(REL_, OUT_)                                                                                   \
  hipLaunchKernelGGL((dyn_mask_head_kernel<16, REL_, OUT_>), grid, block, lds, stream, feat, ref, params, Q, H, \
                     W, stride, TR, nbands, th, tw, static_cast<OUT_ *>(up), amask)
  if (out_dtype == 0) {
    if (rel_coord) PCT_DMH(true, float);
    else PCT_DMH(false, float);
  } else if (out_dtype == 2) {
    if (rel_coord) PCT_DMH(true, __bf16);
    else PCT_DMH(false, __bf16);
  } else {
    return -1;
  }
#undef PCT_DMH
  return (int)hipGetLastError();
}

}  // namespace pct
