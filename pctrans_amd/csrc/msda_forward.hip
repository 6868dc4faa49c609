// MSDeformAttn forward for MI355X (gfx950, wave64).
//
// Semantics: the reference forward kernel, ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 with the bilinear
// fetch of cuh:38-89 -- out[b,q,m,:] = sum_{l,p} w[b,q,m,l,p] * bilinear(value_l[b,:,m,:], loc*(W_l,H_l)-0.5),
// a sample contributing only when -1 < h_im < H_l and -1 < w_im < W_l, corners outside the map reading 0.
//
// Mapping (not the reference's one-thread-per-output-scalar):
//   * one lane owns VEC contiguous channels (16 B) of one (b, q, m) record, so a bilinear corner of one head
//     (D = 16 fp32 -> 64 B) is fetched by D/VEC adjacent lanes as one coalesced segment, and a wave's output
//     store is 1 KiB contiguous;
//   * the (x, y, weight) triples of every record a block owns are contiguous in HBM; they are staged ONCE per
//     block into LDS with coalesced loads (padded record stride -> conflict-free broadcast reads) instead of
//     being re-read from global memory by each of the D threads that share them (cuh:286-288);
//   * corner loads are unconditional on clamped addresses and masked by select afterwards (no divergent
//     branches around loads; an Inf/NaN in an unread texel can not leak through a 0 * v product);
//   * the point loop is unrolled at compile time for P = 4 / 8 so 4*P 16-byte gathers are in flight per lane;
//   * logical block order is remapped so that each XCD's L2 serves one contiguous range of queries.
#include <stdlib.h>

#include <atomic>

#include "msda_common.hpp"

namespace pct {

constexpr int FWD_BLOCK = 256;

template <typename T, int VEC>
__device__ __forceinline__ void load_channels(const typename Traits<T>::store_t *p,
                                              typename Traits<T>::acc_t (&dst)[VEC])
{
  using S = typename Traits<T>::store_t;
  const vec_t<S, VEC> v = *reinterpret_cast<const vec_t<S, VEC> *>(p);
#pragma unroll
  for (int k = 0; k < VEC; ++k) dst[k] = Traits<T>::to_acc(v[k]);
}
template <typename T>
__device__ __forceinline__ void load_channels_scalar(const typename Traits<T>::store_t *p,
                                                     typename Traits<T>::acc_t (&dst)[1])
{
  dst[0] = Traits<T>::to_acc(*p);
}

// T: storage tag of value/out.  LT: dtype of sampling_loc / attn_weight in HBM (== acc type).
// VEC: channels per lane.  CVT: lanes per record (D / VEC) when known at compile time, 0 = runtime.
// PT: points per level when known at compile time, 0 = runtime.
template <typename T, int VEC, int CVT, int PT>
__global__ __launch_bounds__(FWD_BLOCK) void msda_forward_kernel(
    const typename Traits<T>::store_t *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const typename Traits<T>::acc_t *__restrict__ loc,
    const typename Traits<T>::acc_t *__restrict__ attn, const int S, const int M, const int D, const int L,
    const int Lq, const int P_rt, const int CV_rt, const long long total_lanes, const int rec_stride,
    typename Traits<T>::store_t *__restrict__ out)
{
  using A = typename Traits<T>::acc_t;
  using ST = typename Traits<T>::store_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  A *lds = reinterpret_cast<A *>(smem_raw);

  const int P = PT > 0 ? PT : P_rt;
  const int CV = CVT > 0 ? CVT : CV_rt;
  const int LP = L * P;

  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const long long lane0 = (long long)lb * FWD_BLOCK;           // first global lane of this block
  long long lane_end = lane0 + FWD_BLOCK;
  if (lane_end > total_lanes) lane_end = total_lanes;
  const long long rec0 = lane0 / CV;                            // first record (b,q,m) of this block
  const int nrec = (int)((lane_end - 1) / CV - rec0) + 1;

  // ---- stage (x, y) pairs and weights of the block's records into LDS -------------------------------------
  {
    const A *gl = loc + rec0 * LP * 2;
    const A *gw = attn + rec0 * LP;
    const int nl = nrec * LP * 2;
    const int nw = nrec * LP;
    if constexpr (sizeof(A) == 4) {
      // 16-byte loads when every record starts 16-B aligned (LP*2 and LP multiples of 4, e.g. P = 4 / 8)
      if ((LP & 3) == 0 && (((uintptr_t)gl | (uintptr_t)gw) & 15u) == 0) {
        using f4 = vec_t<float, 4>;
        using f2 = vec_t<float, 2>;
        for (int i = threadIdx.x * 4; i < nl; i += FWD_BLOCK * 4) {
          const f4 v = *reinterpret_cast<const f4 *>(gl + i);
          const int r = i / (LP * 2), o = i - r * (LP * 2);
          float *d = lds + r * rec_stride + o;                 // 8-B aligned (rec_stride and o are even)
          *reinterpret_cast<f2 *>(d) = f2{v[0], v[1]};
          *reinterpret_cast<f2 *>(d + 2) = f2{v[2], v[3]};
        }
        for (int i = threadIdx.x * 4; i < nw; i += FWD_BLOCK * 4) {
          const f4 v = *reinterpret_cast<const f4 *>(gw + i);
          const int r = i / LP, o = i - r * LP;
          float *d = lds + r * rec_stride + LP * 2 + o;
          *reinterpret_cast<f2 *>(d) = f2{v[0], v[1]};
          *reinterpret_cast<f2 *>(d + 2) = f2{v[2], v[3]};
        }
      } else {
        for (int i = threadIdx.x; i < nl; i += FWD_BLOCK) {
          const int r = i / (LP * 2), o = i - r * (LP * 2);
          lds[r * rec_stride + o] = gl[i];
        }
        for (int i = threadIdx.x; i < nw; i += FWD_BLOCK) {
          const int r = i / LP, o = i - r * LP;
          lds[r * rec_stride + LP * 2 + o] = gw[i];
        }
      }
    } else {
      for (int i = threadIdx.x; i < nl; i += FWD_BLOCK) {
        const int r = i / (LP * 2), o = i - r * (LP * 2);
        lds[r * rec_stride + o] = gl[i];
      }
      for (int i = threadIdx.x; i < nw; i += FWD_BLOCK) {
        const int r = i / LP, o = i - r * LP;
        lds[r * rec_stride + LP * 2 + o] = gw[i];
      }
    }
  }
  __syncthreads();

  const long long gl_lane = lane0 + threadIdx.x;
  if (gl_lane >= total_lanes) return;
  const long long rec = gl_lane / CV;                           // (b*Lq + q)*M + m
  const int cv = (int)(gl_lane - rec * CV);
  const int m = (int)(rec % M);
  const long long b = rec / ((long long)M * Lq);
  const int MD = M * D;
  const ST *vb = value + b * (long long)S * MD + m * D + cv * VEC;
  const A *rl = lds + (int)(rec - rec0) * rec_stride;          // this record's (x,y) pairs
  const A *rw = rl + LP * 2;                                    // ... and weights

  A acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0;

  auto sample = [&](const int H, const int W, const int lvl_off, const A loc_w, const A loc_h, const A weight) {
    const A h_im = loc_h * H - (A)0.5;
    const A w_im = loc_w * W - (A)0.5;
    const bool gate = h_im > -1 && w_im > -1 && h_im < H && w_im < W;   // false for NaN
    const A hf = floor(h_im), wf = floor(w_im);
    const int h_low = gate ? (int)hf : 0, w_low = gate ? (int)wf : 0;   // keep the index math in range
    // a gated-out sample must contribute exactly 0 (the reference skips it, cuh:290-296): its fractional parts
    // and weight may be NaN/Inf (NaN location, Inf*0), so neutralise them instead of relying on 0 * x
    const A lh = gate ? h_im - hf : (A)0, lw = gate ? w_im - wf : (A)0;
    const A wgt = gate ? weight : (A)0;
    const A hh = 1 - lh, hw = 1 - lw;
    const bool top = gate && h_low >= 0, bot = gate && h_low + 1 <= H - 1;
    const bool lft = w_low >= 0, rgt = w_low + 1 <= W - 1;
    const bool ok1 = top && lft, ok2 = top && rgt, ok3 = bot && lft, ok4 = bot && rgt;
    const int o1 = lvl_off + (h_low * W + w_low) * MD;
    const int o2 = o1 + MD, o3 = o1 + W * MD, o4 = o3 + MD;
    A v1[VEC], v2[VEC], v3[VEC], v4[VEC];
    if constexpr (VEC > 1) {
      load_channels<T, VEC>(vb + (ok1 ? o1 : 0), v1);
      load_channels<T, VEC>(vb + (ok2 ? o2 : 0), v2);
      load_channels<T, VEC>(vb + (ok3 ? o3 : 0), v3);
      load_channels<T, VEC>(vb + (ok4 ? o4 : 0), v4);
    } else {
      load_channels_scalar<T>(vb + (ok1 ? o1 : 0), v1);
      load_channels_scalar<T>(vb + (ok2 ? o2 : 0), v2);
      load_channels_scalar<T>(vb + (ok3 ? o3 : 0), v3);
      load_channels_scalar<T>(vb + (ok4 ? o4 : 0), v4);
    }
    const A w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const A a1 = ok1 ? v1[k] : (A)0, a2 = ok2 ? v2[k] : (A)0;
      const A a3 = ok3 ? v3[k] : (A)0, a4 = ok4 ? v4[k] : (A)0;
      acc[k] += (w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4) * wgt;
    }
  };

  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const int lvl_off = (int)starts[l] * MD;
    const A *pl = rl + l * P * 2;
    const A *pw = rw + l * P;
    if constexpr (PT > 0) {
#pragma unroll
      for (int p = 0; p < PT; ++p) sample(H, W, lvl_off, pl[2 * p], pl[2 * p + 1], pw[p]);
    } else {
      for (int p = 0; p < P; ++p) sample(H, W, lvl_off, pl[2 * p], pl[2 * p + 1], pw[p]);
    }
  }

  ST *op = out + rec * D + cv * VEC;
  if constexpr (VEC > 1) {
    vec_t<ST, VEC> o;
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = Traits<T>::from_acc(acc[k]);
    *reinterpret_cast<vec_t<ST, VEC> *>(op) = o;
  } else {
    *op = Traits<T>::from_acc(acc[0]);
  }
}

// specialised kernels for PCTrans' geometry; each returns -100 when it does not cover the call
template <typename T>
int launch_msda_forward_win(const void *, const int64_t *, const int64_t *, const void *, const void *, int, int, int,
                            int, int, int, int, void *, hipStream_t, const float *, long long);   // msda_forward_win.hip
template <typename T>
int launch_msda_forward_dpp(const void *, const int64_t *, const int64_t *, const void *, const void *, int, int, int,
                            int, int, int, int, void *, hipStream_t, const float *, long long);   // msda_forward_dpp.hip

int launch_msda_forward_col(const void *, const int64_t *, const int64_t *, const void *, const void *, int, int, int,
                            int, int, int, int, void *, hipStream_t, const float *, long long, bool);   // msda_forward_col.hip
template <typename T>
int launch_msda_forward_col16(const void *, const int64_t *, const int64_t *, const void *, const void *, int, int, int,
                              int, int, int, int, void *, hipStream_t, const float *, long long);  // msda_forward_col16.hip

// Kernel choice for PCTrans' geometry.  Default "auto": the windowed-LDS kernel when the queries are the pyramid's
// own pixels (Lq == S: neighbouring queries sample neighbouring texels, the case it is built for) and the problem fills
// its persistent grid (except where it was measured slower, see below), else the quad-owner kernel.
// PCT_MSDA_KERNEL = auto | win | dpp | generic | col (development A/B).
static std::atomic<int> g_kernel_override{-1};     // pct_msda_set_kernel_choice (diagnostic): -1 = follow the environment
static std::atomic<int> g_last_kernel{0};          // pct_msda_last_kernel (diagnostic): what the last forward call launched
void set_msda_kernel_choice(int v) { g_kernel_override.store((v >= 0 && v <= 4) ? v : -1, std::memory_order_relaxed); }
int msda_last_kernel() { return g_last_kernel.load(std::memory_order_relaxed); }
void note_msda_kernel(int k) { g_last_kernel.store(k, std::memory_order_relaxed); }

int msda_kernel_choice()
{
  const int ov = g_kernel_override.load(std::memory_order_relaxed);
  if (ov >= 0) return ov;
  static const int v = [] {
    const char *e = getenv("PCT_MSDA_KERNEL");
    if (!e) return 0;
    if (e[0] == 'w') return 1;
    if (e[0] == 'd') return 3;
    if (e[0] == 'g') return 2;
    if (e[0] == 'c') return 4;
    return 0;
  }();
  return v;
}

// shared by the plain and the fused entry points: -100 = geometry not covered by the specialised kernels
template <typename T>
int launch_msda_forward_special(const void *value, const int64_t *shapes, const int64_t *starts, const void *loc,
                                const void *attn, int N, int S, int M, int D, int L, int Lq, int P, void *out,
                                hipStream_t stream, const float *ref, long long ref_batch_stride)
{
  const int choice = msda_kernel_choice();
  if (choice == 2) return -100;
  int rc = -100;
  // 16-bit values with 8 points per level (SURVEY config 5): the windowed variant (2 lanes per head-pixel, 4 points per
  // lane and level, 20-40 spilled VGPRs) loses to the quad-owner kernel (P3, N = 1: 0.49 vs 0.34 ms) -- measured, so
  // "auto" skips it there; fp32 with 8 points and 16-bit with 4 points stay on the windowed kernel (faster)
  const bool win_loses = sizeof(typename Traits<T>::store_t) == 2 && P == 8;
  // ... and a persistent grid of 768 workgroups needs about three work items each to beat the quad-owner kernel
  // (P2: N = 1 0.046 vs 0.040 ms, N = 2 equal, N = 4 0.089 vs 0.113 ms; P4, N = 1: 0.024 vs 0.014 ms)
  const long long per_item = P == 8 ? 128 : (sizeof(typename Traits<T>::store_t) == 2 ? 512 : 256);   // queries per item
  const bool win_small = (long long)N * ((Lq + per_item - 1) / per_item) * M < 3 * 768;
  // the pyramid-column kernel (fp32, 4 points): persistent grid of 768 workgroups with items of <= 256 queries.  Measured
  // crossover against the windowed / quad-owner kernels (model-like locations, profiles/r02_small_batch.txt): with 4
  // levels it wins from one 512^2 image on (N * S * M = 174 k); with 3 levels from about 450 k (quad-owner before that)
  if constexpr (sizeof(typename Traits<T>::store_t) == 4) {
    const bool col_big = (long long)N * S * M >= (L >= 4 ? 160000LL : 450000LL);
    if (choice == 4 || (choice == 0 && Lq == S && P == 4 && col_big)) {
      rc = launch_msda_forward_col(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P, out, stream, ref, ref_batch_stride, false);
      if (rc != -100) note_msda_kernel(4);
    }
  }
  if constexpr (sizeof(typename Traits<T>::store_t) == 2) {
    // 16-bit values: the column kernel's 16-bit variant (4 or 8 points; two lanes per (query, head) with 8 points)
    const bool col_big = (long long)N * S * M * (P / 4) >= (L >= 4 ? 160000LL : 450000LL);
    if (choice == 4 || (choice == 0 && Lq == S && (P == 4 || P == 8) && col_big)) {
      rc = launch_msda_forward_col16<T>(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P, out, stream, ref,
                                        ref_batch_stride);
      if (rc != -100) note_msda_kernel(4);
    }
  }
  if (rc == -100 && (choice == 1 || (choice == 0 && Lq == S && !win_loses && !win_small))) {
    rc = launch_msda_forward_win<T>(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P, out, stream, ref,
                                    ref_batch_stride);
    if (rc != -100) note_msda_kernel(1);
  }
  if (rc == -100) {
    rc = launch_msda_forward_dpp<T>(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P, out, stream, ref,
                                    ref_batch_stride);
    if (rc != -100) note_msda_kernel(3);
  }
  return rc;
}
// piece-plane operands (pct_ms_deform_attn_forward_planes_f32): the pyramid-column kernel or nothing (-100)
int launch_msda_forward_planes(const void *value, const int64_t *shapes, const int64_t *starts, const void *loc,
                               const void *attn, int N, int S, int M, int D, int L, int Lq, int P, void *out,
                               hipStream_t stream, const float *ref, long long ref_batch_stride)
{
  const int rc = launch_msda_forward_col(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P, out, stream, ref,
                                         ref_batch_stride, true);
  if (rc != -100) note_msda_kernel(4);
  return rc;
}

template int launch_msda_forward_special<float>(const void *, const int64_t *, const int64_t *, const void *,
                                                const void *, int, int, int, int, int, int, int, void *, hipStream_t,
                                                const float *, long long);

// ---- host-side launcher ---------------------------------------------------------------------------------
template <typename T>
int launch_msda_forward(const void *value, const int64_t *shapes, const int64_t *starts, const void *loc,
                        const void *attn, int N, int S, int M, int D, int L, int Lq, int P, void *out,
                        hipStream_t stream)
{
  using A = typename Traits<T>::acc_t;
  using ST = typename Traits<T>::store_t;
  if constexpr (sizeof(A) == 4) {
    const int rc = launch_msda_forward_special<T>(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P, out, stream,
                                                  nullptr, 0);
    if (rc != -100) return rc;
  }
  note_msda_kernel(2);
  constexpr int VECW = 16 / (int)sizeof(ST);                    // channels in one 16-byte lane load
  const bool aligned16 = (((uintptr_t)value | (uintptr_t)out) & 15u) == 0 &&
                         (((uintptr_t)loc | (uintptr_t)attn) & 15u) == 0;
  const bool vec = aligned16 && (D % VECW) == 0;
  const int VEC = vec ? VECW : 1;
  const int CV = D / VEC;
  const long long total_lanes = (long long)N * Lq * M * CV;
  if (total_lanes == 0) return 0;
  const long long nblk = (total_lanes + FWD_BLOCK - 1) / FWD_BLOCK;
  if (nblk > 0x7fffffffLL) return -4;
  const int LP = L * P;
  const int rec_stride = sizeof(A) == 4 ? padded_record_stride(LP * 3) : LP * 3 + 1;
  const int nrec_max = (FWD_BLOCK + CV - 2) / CV + 1;
  const size_t lds_bytes = (size_t)nrec_max * rec_stride * sizeof(A);
  if (lds_bytes > 64 * 1024) return -4;

  const dim3 grid((unsigned)nblk), block(FWD_BLOCK);
  const ST *v = static_cast<const ST *>(value);
  const A *lc = static_cast<const A *>(loc);
  const A *at = static_cast<const A *>(attn);
  ST *o = static_cast<ST *>(out);

#define PCT_LAUNCH(VEC_, CVT_, PT_)                                                                        \
  hipLaunchKernelGGL((msda_forward_kernel<T, VEC_, CVT_, PT_>), grid, block, lds_bytes, stream, v, shapes, \
                     starts, lc, at, S, M, D, L, Lq, P, CV, total_lanes, rec_stride, o)
  if (vec) {
    if (D == 16 && P == 4) {
      PCT_LAUNCH(VECW, 16 / VECW, 4);
    } else if (D == 16 && P == 8) {
      PCT_LAUNCH(VECW, 16 / VECW, 8);
    } else if (D == 32 && P == 4) {
      PCT_LAUNCH(VECW, 32 / VECW, 4);
    } else if (D == 32 && P == 8) {
      PCT_LAUNCH(VECW, 32 / VECW, 8);
    } else if (P == 4) {
      PCT_LAUNCH(VECW, 0, 4);
    } else {
      PCT_LAUNCH(VECW, 0, 0);
    }
  } else {
    PCT_LAUNCH(1, 0, 0);
  }
#undef PCT_LAUNCH
  return (int)hipGetLastError();
}

template int launch_msda_forward<float>(const void *, const int64_t *, const int64_t *, const void *,
                                        const void *, int, int, int, int, int, int, int, void *, hipStream_t);
template int launch_msda_forward<double>(const void *, const int64_t *, const int64_t *, const void *,
                                         const void *, int, int, int, int, int, int, int, void *, hipStream_t);
template int launch_msda_forward<half_bits>(const void *, const int64_t *, const int64_t *, const void *,
                                            const void *, int, int, int, int, int, int, int, void *,
                                            hipStream_t);
template int launch_msda_forward<bf16_bits>(const void *, const int64_t *, const int64_t *, const void *,
                                            const void *, int, int, int, int, int, int, int, void *,
                                            hipStream_t);

}  // namespace pct
