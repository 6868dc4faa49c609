/*
 * oracle/msda_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference's multi-scale deformable
 * attention op (forward + backward).  It exists only so that tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg have an
 * independent checker / CPU timing baseline.  Nothing under pctrans_amd/
 * may import, link or call it: the product path is the HIP library and
 * fails loudly when that library is missing.
 *
 * What it follows (paths relative to the reference checkout, abbreviations
 * as in SURVEY.md: `cuh` = .../pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh,
 * `cu` = .../ops/src/cuda/ms_deform_attn_cuda.cu):
 *   - bilinear corner fetch + weights ........ cuh:38-89   (forward device fn)
 *   - per-output accumulation over L x P ...... cuh:242-304 (forward kernel)
 *   - sample gate  -1 < h_im < H, -1 < w_im < W  cuh:290-296
 *   - backward of one sample .................. cuh:92-164  (col2im bilinear)
 *   - backward accumulation over channels ..... cuh:306-408 (reduce over D)
 *   - batch / im2col_step precondition ........ cu:55-57
 *
 * Pinning: the forward is pinned against golden vectors produced by the
 * reference's own pure-PyTorch oracle `ms_deform_attn_core_pytorch`
 * (.../ops/functions/ms_deform_attn_func.py:52-72), which the reference's
 * only hot-path test (.../ops/test.py:35-60) asserts its CUDA extension equal
 * to.  Fixtures: the .npz files under tests/golden/, generator: tests/golden/make_golden.py.
 * The backward is pinned against torch autograd through that same reference
 * function (fixtures `*_grad.npz`) and by finite differences.
 *
 * Arithmetic mirrors the reference expression order; build with
 * -ffp-contract=off so that no FMA contraction reorders roundings.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

ORACLE_API int msda_oracle_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

ORACLE_API void msda_oracle_set_threads(int n)
{
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

#define DEFINE_ORACLE(SUFFIX, T, FLOOR)                                                        \
  /* forward: cuh:242-304 + cuh:38-89.  out is [N, Lq, M, D]. */                               \
  ORACLE_API int msda_oracle_forward_##SUFFIX(                                                 \
      const T *value, const int64_t *shapes, const int64_t *starts, const T *loc,              \
      const T *attn, int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,        \
      T *out)                                                                                  \
  {                                                                                            \
    if (N <= 0) return 0;                                                                      \
    int step = N < im2col_step ? N : im2col_step;                                              \
    if (step <= 0 || N % step != 0) return -1; /* cu:55-57 */                                  \
    const int64_t NQ = (int64_t)N * Lq;                                                        \
    _Pragma("omp parallel for schedule(static)")                                               \
    for (int64_t bq = 0; bq < NQ; ++bq) {                                                      \
      const int b = (int)(bq / Lq);                                                            \
      const T *vb = value + (size_t)b * S * M * D;                                             \
      for (int m = 0; m < M; ++m) {                                                            \
        const size_t pair = (size_t)bq * M + m;                                                \
        const T *lp = loc + pair * L * P * 2;                                                  \
        const T *wp = attn + pair * L * P;                                                     \
        T *op = out + pair * D;                                                                \
        for (int c = 0; c < D; ++c) op[c] = 0;                                                 \
        for (int l = 0; l < L; ++l) {                                                          \
          const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                        \
          const T *vl = vb + (size_t)starts[l] * M * D;                                        \
          for (int p = 0; p < P; ++p) {                                                        \
            const T loc_w = lp[(l * P + p) * 2], loc_h = lp[(l * P + p) * 2 + 1];              \
            const T weight = wp[l * P + p];                                                    \
            const T h_im = loc_h * H - (T)0.5, w_im = loc_w * W - (T)0.5;                      \
            if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;                   \
            const int h_low = (int)FLOOR(h_im), w_low = (int)FLOOR(w_im);                      \
            const int h_high = h_low + 1, w_high = w_low + 1;                                  \
            const T lh = h_im - h_low, lw = w_im - w_low;                                      \
            const T hh = 1 - lh, hw = 1 - lw;                                                  \
            const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                    \
            const int ok1 = h_low >= 0 && w_low >= 0;                                          \
            const int ok2 = h_low >= 0 && w_high <= W - 1;                                     \
            const int ok3 = h_high <= H - 1 && w_low >= 0;                                     \
            const int ok4 = h_high <= H - 1 && w_high <= W - 1;                                \
            const ptrdiff_t o1 = ((ptrdiff_t)(h_low * W + w_low) * M + m) * D;                 \
            const ptrdiff_t o2 = o1 + (ptrdiff_t)M * D;                                        \
            const ptrdiff_t o3 = o1 + (ptrdiff_t)W * M * D;                                    \
            const ptrdiff_t o4 = o3 + (ptrdiff_t)M * D;                                        \
            for (int c = 0; c < D; ++c) {                                                      \
              const T v1 = ok1 ? vl[o1 + c] : 0, v2 = ok2 ? vl[o2 + c] : 0;                    \
              const T v3 = ok3 ? vl[o3 + c] : 0, v4 = ok4 ? vl[o4 + c] : 0;                    \
              const T val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                           \
              op[c] += val * weight;                                                           \
            }                                                                                  \
          }                                                                                    \
        }                                                                                      \
      }                                                                                        \
    }                                                                                          \
    return 0;                                                                                  \
  }                                                                                            \
                                                                                               \
  /* backward: cuh:92-164 per sample, channel reduction as cuh:306-408.                        \
   * grad_value accumulates (the reference uses atomicAdd); it is serial here                  \
   * so the summation order is deterministic: (b, q, m, l, p, corner, c). */                   \
  ORACLE_API int msda_oracle_backward_##SUFFIX(                                                \
      const T *value, const int64_t *shapes, const int64_t *starts, const T *loc,              \
      const T *attn, const T *grad_out, int N, int S, int M, int D, int L, int Lq, int P,      \
      int im2col_step, T *grad_value, T *grad_loc, T *grad_attn)                               \
  {                                                                                            \
    if (N <= 0) return 0;                                                                      \
    int step = N < im2col_step ? N : im2col_step;                                              \
    if (step <= 0 || N % step != 0) return -1;                                                 \
    memset(grad_value, 0, sizeof(T) * (size_t)N * S * M * D);                                  \
    memset(grad_loc, 0, sizeof(T) * (size_t)N * Lq * M * L * P * 2);                           \
    memset(grad_attn, 0, sizeof(T) * (size_t)N * Lq * M * L * P);                              \
    /* parallel over images only: grad_value of image b is private to b */                     \
    _Pragma("omp parallel for schedule(dynamic, 1)")                                           \
    for (int b = 0; b < N; ++b) {                                                              \
      const T *vb = value + (size_t)b * S * M * D;                                             \
      T *gvb = grad_value + (size_t)b * S * M * D;                                             \
      for (int q = 0; q < Lq; ++q)                                                             \
        for (int m = 0; m < M; ++m) {                                                          \
          const size_t pair = ((size_t)b * Lq + q) * M + m;                                    \
          const T *lp = loc + pair * L * P * 2;                                                \
          const T *wp = attn + pair * L * P;                                                   \
          const T *go = grad_out + pair * D;                                                   \
          T *glp = grad_loc + pair * L * P * 2;                                                \
          T *gwp = grad_attn + pair * L * P;                                                   \
          for (int l = 0; l < L; ++l) {                                                        \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                      \
            const size_t lvl = (size_t)starts[l] * M * D;                                      \
            for (int p = 0; p < P; ++p) {                                                      \
              const T loc_w = lp[(l * P + p) * 2], loc_h = lp[(l * P + p) * 2 + 1];            \
              const T weight = wp[l * P + p];                                                  \
              const T h_im = loc_h * H - (T)0.5, w_im = loc_w * W - (T)0.5;                    \
              if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;                 \
              const int h_low = (int)FLOOR(h_im), w_low = (int)FLOOR(w_im);                    \
              const int h_high = h_low + 1, w_high = w_low + 1;                                \
              const T lh = h_im - h_low, lw = w_im - w_low;                                    \
              const T hh = 1 - lh, hw = 1 - lw;                                                \
              const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                  \
              const int ok1 = h_low >= 0 && w_low >= 0;                                        \
              const int ok2 = h_low >= 0 && w_high <= W - 1;                                   \
              const int ok3 = h_high <= H - 1 && w_low >= 0;                                   \
              const int ok4 = h_high <= H - 1 && w_high <= W - 1;                              \
              const ptrdiff_t o1 = (ptrdiff_t)lvl + ((ptrdiff_t)(h_low * W + w_low) * M + m) * D; \
              const ptrdiff_t o2 = o1 + (ptrdiff_t)M * D;                                      \
              const ptrdiff_t o3 = o1 + (ptrdiff_t)W * M * D;                                  \
              const ptrdiff_t o4 = o3 + (ptrdiff_t)M * D;                                      \
              T g_w = 0, g_h = 0, g_a = 0;                                                     \
              for (int c = 0; c < D; ++c) {                                                    \
                const T top_grad = go[c];                                                      \
                const T top_grad_value = top_grad * weight;                                    \
                T grad_h_weight = 0, grad_w_weight = 0;                                        \
                T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                              \
                if (ok1) { v1 = vb[o1 + c]; grad_h_weight -= hw * v1; grad_w_weight -= hh * v1; \
                           gvb[o1 + c] += w1 * top_grad_value; }                               \
                if (ok2) { v2 = vb[o2 + c]; grad_h_weight -= lw * v2; grad_w_weight += hh * v2; \
                           gvb[o2 + c] += w2 * top_grad_value; }                               \
                if (ok3) { v3 = vb[o3 + c]; grad_h_weight += hw * v3; grad_w_weight -= lh * v3; \
                           gvb[o3 + c] += w3 * top_grad_value; }                               \
                if (ok4) { v4 = vb[o4 + c]; grad_h_weight += lw * v4; grad_w_weight += lh * v4; \
                           gvb[o4 + c] += w4 * top_grad_value; }                               \
                const T val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                         \
                g_a += top_grad * val;                                                         \
                g_w += W * grad_w_weight * top_grad_value;                                     \
                g_h += H * grad_h_weight * top_grad_value;                                     \
              }                                                                                \
              glp[(l * P + p) * 2] = g_w;                                                      \
              glp[(l * P + p) * 2 + 1] = g_h;                                                  \
              gwp[l * P + p] = g_a;                                                            \
            }                                                                                  \
          }                                                                                    \
        }                                                                                      \
    }                                                                                          \
    return 0;                                                                                  \
  }

DEFINE_ORACLE(f32, float, floorf)
DEFINE_ORACLE(f64, double, floor)
