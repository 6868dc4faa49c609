// MSDeformAttn backward for MI355X (gfx950, wave64).
//
// Semantics: ops/src/cuda/ms_deform_im2col_cuda.cuh:92-164 (one sample) summed over the channels of a head as
// the reference's `..._shm_blocksize_aware_reduce_v1` family does (cuh:306-408; dispatch cuh:1033-1323):
//   grad_value[b, corner, m, c]  += corner_weight * grad_out[b,q,m,c] * w[b,q,m,l,p]        (scatter-add)
//   grad_attn[b,q,m,l,p]          = sum_c grad_out[c] * bilinear(value)[c]
//   grad_loc[b,q,m,l,p,(x,y)]     = (W_l, H_l) * sum_c d bilinear / d(w,h) * grad_out[c] * w
//
// Mapping: identical to the forward (one lane = VEC channels of one (b,q,m) record; the record's (x,y,weight)
// triples staged once per block in LDS).  The reference runs this with blockDim = D = 16 threads per record and a
// serial LDS sum by thread 0 (cuh:385-400); here the D/VEC lanes of a record sit side by side in one wave and
// the three channel sums are xor-butterflies in registers.  Their results overwrite the record's staged
// (x,y,weight) slots in LDS and leave the block as coalesced 16-byte stores.  grad_value uses hardware float
// atomics (global_atomic_add_f32 / _f64), 64 contiguous bytes per record-corner.
#include <stdlib.h>
#include <string.h>

#include "msda_common.hpp"

namespace pct {

constexpr int BWD_BLOCK = 256;

template <typename A>
__device__ __forceinline__ void hw_atomic_add(A *p, A v)
{
  unsafeAtomicAdd(p, v);  // global_atomic_add_f32 / global_atomic_add_f64, no CAS loop
}

// SHFL: the CV lanes of a record form an aligned power-of-two group inside one wave -> butterfly reduce and
//       plain stores.  Otherwise (odd channel counts, D/VEC > 64): atomics into pre-zeroed grad_loc / grad_attn.
template <typename A, int VEC, int CVT, int PT, bool SHFL>
__global__ __launch_bounds__(BWD_BLOCK) void msda_backward_kernel(
    const A *__restrict__ grad_out, const A *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const A *__restrict__ loc, const A *__restrict__ attn, const int S,
    const int M, const int D, const int L, const int Lq, const int P_rt, const int CV_rt,
    const long long total_lanes, const int rec_stride, A *__restrict__ grad_value, A *__restrict__ grad_loc,
    A *__restrict__ grad_attn)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  A *lds = reinterpret_cast<A *>(smem_raw);

  const int P = PT > 0 ? PT : P_rt;
  const int CV = CVT > 0 ? CVT : CV_rt;
  const int LP = L * P;

  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const long long lane0 = (long long)lb * BWD_BLOCK;
  long long lane_end = lane0 + BWD_BLOCK;
  if (lane_end > total_lanes) lane_end = total_lanes;
  const long long rec0 = lane0 / CV;
  const int nrec = (int)((lane_end - 1) / CV - rec0) + 1;
  const int nl = nrec * LP * 2, nw = nrec * LP;
  const A *gl = loc + rec0 * LP * 2;
  const A *gw = attn + rec0 * LP;

  for (int i = threadIdx.x; i < nl; i += BWD_BLOCK) {
    const int r = i / (LP * 2), o = i - r * (LP * 2);
    lds[r * rec_stride + o] = gl[i];
  }
  for (int i = threadIdx.x; i < nw; i += BWD_BLOCK) {
    const int r = i / LP, o = i - r * LP;
    lds[r * rec_stride + LP * 2 + o] = gw[i];
  }
  __syncthreads();

  const long long gl_lane = lane0 + threadIdx.x;
  const bool active = gl_lane < total_lanes;
  const long long rec = active ? gl_lane / CV : rec0;
  const int cv = active ? (int)(gl_lane - rec * CV) : 0;
  const int m = (int)(rec % M);
  const long long b = rec / ((long long)M * Lq);
  const int MD = M * D;
  const long long img = b * (long long)S * MD + m * D + cv * VEC;
  const A *vb = value + img;
  A *gvb = grad_value + img;
  A *rl = lds + (int)(rec - rec0) * rec_stride;
  A *rw = rl + LP * 2;

  A top[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) top[k] = active ? grad_out[rec * D + cv * VEC + k] : (A)0;

  auto sample = [&](const int l, const int p, const int H, const int W, const int lvl_off) {
    const int sidx = l * P + p;
    const A loc_w = rl[2 * sidx], loc_h = rl[2 * sidx + 1], weight_in = rw[sidx];
    const A h_im = loc_h * H - (A)0.5;
    const A w_im = loc_w * W - (A)0.5;
    const bool gate = active && h_im > -1 && w_im > -1 && h_im < H && w_im < W;
    const A hf = floor(h_im), wf = floor(w_im);
    const int h_low = gate ? (int)hf : 0, w_low = gate ? (int)wf : 0;
    // gated-out samples get exactly-zero gradients (the reference skips them): neutralise NaN/Inf operands
    const A lh = gate ? h_im - hf : (A)0, lw = gate ? w_im - wf : (A)0;
    const A weight = gate ? weight_in : (A)0;
    const A hh = 1 - lh, hw = 1 - lw;
    const bool tp = gate && h_low >= 0, bt = gate && h_low + 1 <= H - 1;
    const bool lf = w_low >= 0, rg = w_low + 1 <= W - 1;
    const bool ok1 = tp && lf, ok2 = tp && rg, ok3 = bt && lf, ok4 = bt && rg;
    const int o1 = lvl_off + (h_low * W + w_low) * MD;
    const int o2 = o1 + MD, o3 = o1 + W * MD, o4 = o3 + MD;
    const A w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    A g_w = 0, g_h = 0, g_a = 0;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const A top_grad_value = top[k] * weight;
      A grad_h_weight = 0, grad_w_weight = 0;
      A v1 = 0, v2 = 0, v3 = 0, v4 = 0;
      if (ok1) { v1 = vb[o1 + k]; grad_h_weight -= hw * v1; grad_w_weight -= hh * v1;
                 hw_atomic_add(gvb + o1 + k, w1 * top_grad_value); }
      if (ok2) { v2 = vb[o2 + k]; grad_h_weight -= lw * v2; grad_w_weight += hh * v2;
                 hw_atomic_add(gvb + o2 + k, w2 * top_grad_value); }
      if (ok3) { v3 = vb[o3 + k]; grad_h_weight += hw * v3; grad_w_weight -= lh * v3;
                 hw_atomic_add(gvb + o3 + k, w3 * top_grad_value); }
      if (ok4) { v4 = vb[o4 + k]; grad_h_weight += lw * v4; grad_w_weight += lh * v4;
                 hw_atomic_add(gvb + o4 + k, w4 * top_grad_value); }
      const A val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
      g_a += top[k] * val;
      g_w += W * grad_w_weight * top_grad_value;
      g_h += H * grad_h_weight * top_grad_value;
    }
    if constexpr (SHFL) {
#pragma unroll
      for (int o = CVT / 2; o >= 1; o >>= 1) {
        g_w += __shfl_xor(g_w, o);
        g_h += __shfl_xor(g_h, o);
        g_a += __shfl_xor(g_a, o);
      }
      if (cv == 0 && active) {  // every lane of the record has read slot sidx already (same wave, in order)
        // a gated-out sample is skipped by the reference (cuh:352): exact zeros even when grad_out is not finite
        rl[2 * sidx] = gate ? g_w : (A)0;
        rl[2 * sidx + 1] = gate ? g_h : (A)0;
        rw[sidx] = gate ? g_a : (A)0;
      }
    } else {
      if (gate) {
        hw_atomic_add(grad_loc + (rec * LP + sidx) * 2, g_w);
        hw_atomic_add(grad_loc + (rec * LP + sidx) * 2 + 1, g_h);
        hw_atomic_add(grad_attn + rec * LP + sidx, g_a);
      }
    }
  };

  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const int lvl_off = (int)starts[l] * MD;
    if constexpr (PT > 0) {
#pragma unroll
      for (int p = 0; p < PT; ++p) sample(l, p, H, W, lvl_off);
    } else {
      for (int p = 0; p < P; ++p) sample(l, p, H, W, lvl_off);
    }
  }

  if constexpr (SHFL) {
    __syncthreads();
    A *ol = grad_loc + rec0 * LP * 2;
    A *ow = grad_attn + rec0 * LP;
    for (int i = threadIdx.x; i < nl; i += BWD_BLOCK) {
      const int r = i / (LP * 2), o = i - r * (LP * 2);
      ol[i] = lds[r * rec_stride + o];
    }
    for (int i = threadIdx.x; i < nw; i += BWD_BLOCK) {
      const int r = i / LP, o = i - r * LP;
      ow[i] = lds[r * rec_stride + LP * 2 + o];
    }
  }
}

int launch_msda_backward_win(const float *value, const int64_t *shapes, const int64_t *starts, const float *loc,
                             const float *attn, const float *grad_out, int N, int S, int M, int D, int L, int Lq,
                             int P, float *grad_value, float *grad_loc, float *grad_attn, hipStream_t stream);
// msda_backward_col.hip: the pyramid-column kernel and the backward's kernel choice (0 = auto, 1 = windowed, 2 = generic,
// 3 = pyramid-column) / record of the kernel launched last (1 = windowed, 2 = generic, 3 = pyramid-column)
int launch_msda_backward_col(const float *value, const int64_t *shapes, const int64_t *starts, const float *loc,
                             const float *attn, const float *grad_out, int N, int S, int M, int D, int L, int Lq,
                             int P, float *grad_value, float *grad_loc, float *grad_attn, bool forced, hipStream_t stream);
int msda_bwd_kernel_choice();
void note_msda_bwd_kernel(int k);

// Zero-fill by a kernel of our own instead of hipMemsetAsync.  A memset recorded into a HIP graph by torch.cuda.graph does
// not replay reliably on ROCm 7.2 (a stand-alone capture of the same memsets does: tools/micro/graph_memset_replay.hip): from the second replay on, the node wrote a garbage dword into every fourth / second element of
// grad_value (tools/diag_bwd_graph2.py: the windowed and the column backward both came out 0.26 x max|grad_value| off in
// exactly those channels, the first replay and every eager launch being right) -- the same defect the work-queue counters
// ran into in round 1.  Any 4-byte aligned range; 16-byte stores over the aligned middle.
__global__ __launch_bounds__(256) void zero_fill_kernel(unsigned *__restrict__ p, const size_t head, const size_t n16,
                                                        const size_t tail)
{
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (i0 < head) p[i0] = 0u;
  u32x4 *q = reinterpret_cast<u32x4 *>(p + head);
  for (size_t i = i0; i < n16; i += stride) q[i] = u32x4{0u, 0u, 0u, 0u};
  if (i0 < tail) p[head + n16 * 4 + i0] = 0u;
}
static hipError_t zero_fill(void *ptr, const size_t bytes, hipStream_t stream)
{
  if (bytes == 0) return hipSuccess;
#if defined(PCT_EXPERIMENT_BUILD) && defined(PCT_ZERO_BY_MEMSET)
  return hipMemsetAsync(ptr, 0, bytes, stream);              // diagnostic build only (tools/diag_memset_node.py): the runtime's memset node
#endif
  // (every caller passes fp32 / fp64 tensors: 4-byte aligned, a multiple of 4 bytes.  Anything else is refused rather than
  // handed to hipMemsetAsync, whose captured form is what this function exists to avoid)
  if (((uintptr_t)ptr & 3u) || (bytes & 3u)) return hipErrorInvalidValue;
  const size_t words = bytes / 4;
  size_t head = ((16 - ((uintptr_t)ptr & 15u)) & 15u) / 4;
  if (head > words) head = words;
  const size_t n16 = (words - head) / 4, tail = words - head - n16 * 4;
  const size_t want = (n16 + 255) / 256;
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
  hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, stream, static_cast<unsigned *>(ptr), head, n16, tail);
  return hipGetLastError();
}

template <typename A>
int launch_msda_backward(const void *value, const int64_t *shapes, const int64_t *starts, const void *loc,
                         const void *attn, const void *grad_out, int N, int S, int M, int D, int L, int Lq,
                         int P, void *grad_value, void *grad_loc, void *grad_attn, hipStream_t stream)
{
  constexpr int VECW = 16 / (int)sizeof(A);
  const bool vec = (D % VECW) == 0;   // the backward touches channels one scalar at a time: no 16-B alignment need
  const int VEC = vec ? VECW : 1;
  const int CV = D / VEC;
  const bool shfl = vec && CV <= 64 && (CV & (CV - 1)) == 0;
  const long long total_lanes = (long long)N * Lq * M * CV;
  const int LP = L * P;

  hipError_t e = zero_fill(grad_value, sizeof(A) * (size_t)N * S * M * D, stream);
  if (e != hipSuccess) return (int)e;
  if (!shfl) {
    e = zero_fill(grad_loc, sizeof(A) * (size_t)N * Lq * M * LP * 2, stream);
    if (e != hipSuccess) return (int)e;
    e = zero_fill(grad_attn, sizeof(A) * (size_t)N * Lq * M * LP, stream);
    if (e != hipSuccess) return (int)e;
  }
  if (total_lanes == 0) return 0;
  if constexpr (sizeof(A) == 4) {
    // PCT_MSDA_BWD_KERNEL = auto (pyramid-column when the queries are the pyramid's own pixels and the problem fills the
    // chip, else windowed, else generic) | col | win | generic; pct_msda_set_bwd_kernel_choice overrides the environment
    const int mode = msda_bwd_kernel_choice();
    if (shfl && (mode == 0 || mode == 3) && Lq == S) {
      const int rc = launch_msda_backward_col(
          static_cast<const float *>(value), shapes, starts, static_cast<const float *>(loc),
          static_cast<const float *>(attn), static_cast<const float *>(grad_out), N, S, M, D, L, Lq, P,
          static_cast<float *>(grad_value), static_cast<float *>(grad_loc), static_cast<float *>(grad_attn), mode == 3, stream);
      if (rc != -100) {
        note_msda_bwd_kernel(3);
        return rc;
      }
    }
    if (shfl && mode != 2 && (mode == 1 || Lq == S)) {
      const int rc = launch_msda_backward_win(
          static_cast<const float *>(value), shapes, starts, static_cast<const float *>(loc),
          static_cast<const float *>(attn), static_cast<const float *>(grad_out), N, S, M, D, L, Lq, P,
          static_cast<float *>(grad_value), static_cast<float *>(grad_loc), static_cast<float *>(grad_attn), stream);
      if (rc != -100) {
        note_msda_bwd_kernel(1);
        return rc;
      }
    }
  }
  note_msda_bwd_kernel(2);
  const long long nblk = (total_lanes + BWD_BLOCK - 1) / BWD_BLOCK;
  if (nblk > 0x7fffffffLL) return -4;
  const int rec_stride = sizeof(A) == 4 ? padded_record_stride(LP * 3) : LP * 3 + 1;
  const int nrec_max = (BWD_BLOCK + CV - 2) / CV + 1;
  const size_t lds_bytes = (size_t)nrec_max * rec_stride * sizeof(A);
  if (lds_bytes > 64 * 1024) return -4;

  const dim3 grid((unsigned)nblk), block(BWD_BLOCK);
  const A *go = static_cast<const A *>(grad_out), *v = static_cast<const A *>(value);
  const A *lc = static_cast<const A *>(loc), *at = static_cast<const A *>(attn);
  A *gv = static_cast<A *>(grad_value), *gl = static_cast<A *>(grad_loc), *ga = static_cast<A *>(grad_attn);

#define PCT_LAUNCH(VEC_, CVT_, PT_, SH_)                                                                     \
  hipLaunchKernelGGL((msda_backward_kernel<A, VEC_, CVT_, PT_, SH_>), grid, block, lds_bytes, stream, go, v, \
                     shapes, starts, lc, at, S, M, D, L, Lq, P, CV, total_lanes, rec_stride, gv, gl, ga)
#define PCT_CV_CASE(CVT_)                        \
  case CVT_:                                     \
    if (P == 4) PCT_LAUNCH(VECW, CVT_, 4, true); \
    else PCT_LAUNCH(VECW, CVT_, 0, true);        \
    break;
  if (shfl) {
    switch (CV) {
      PCT_CV_CASE(1) PCT_CV_CASE(2) PCT_CV_CASE(4) PCT_CV_CASE(8) PCT_CV_CASE(16) PCT_CV_CASE(32) PCT_CV_CASE(64)
    }
  } else if (vec) {
    PCT_LAUNCH(VECW, 0, 0, false);
  } else {
    PCT_LAUNCH(1, 0, 0, false);
  }
#undef PCT_CV_CASE
#undef PCT_LAUNCH
  return (int)hipGetLastError();
}

template int launch_msda_backward<float>(const void *, const int64_t *, const int64_t *, const void *,
                                         const void *, const void *, int, int, int, int, int, int, int, void *,
                                         void *, void *, hipStream_t);
template int launch_msda_backward<double>(const void *, const int64_t *, const int64_t *, const void *,
                                          const void *, const void *, int, int, int, int, int, int, int, void *,
                                          void *, void *, hipStream_t);

}  // namespace pct
