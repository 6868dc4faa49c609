// extern "C" surface of libpctrans_hip.so -- see include/pctrans_hip.h for the contract and the reference
// interfaces (file:line) each entry point replaces.  Argument validation mirrors the reference host wrappers
// (ops/src/cuda/ms_deform_attn_cuda.cu:33-57, 98-122); device/contiguity checks need tensor metadata and live in
// the Python shim (pctrans_amd/MultiScaleDeformableAttention.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdio.h>

#include "../../include/pctrans_hip.h"
#include "msda_common.hpp"

namespace pct {
template <typename T>
int launch_msda_forward(const void *, const int64_t *, const int64_t *, const void *, const void *, int, int, int,
                        int, int, int, int, void *, hipStream_t);
int launch_linear_k128(const float *x, long long ldx, const float *x2, long long ldx2, long long x2_period, const float *w,
                       const float *bias,
                       long long rows, int n, int epi,
                       float *y, long long ldy, const float *residual, long long ldr, const float *gamma,
                       const float *beta, float eps, hipStream_t stream);
void set_msda_kernel_choice(int v);
int msda_last_kernel();
void set_msda_bwd_kernel_choice(int c);
int msda_bwd_last_kernel();
int launch_linear_k128_split_multi(const float *x, long long ldx, const float *x2, long long ldx2, long long x2_period,
                                   int nseg, const float *const *w, const float *const *bias, const int *n,
                                   const int *use_add, float *const *y, const long long *ldy, long long rows,
                                   hipStream_t stream);
int launch_linear_ln_split(const float *x, long long ldx, const float *w, unsigned short *w_pieces, int K, const float *bias,
                           const float *residual, long long ldr, const float *gamma, const float *beta, float eps,
                           long long rows, float *out, long long ldo, hipStream_t stream);
template <typename A>
int launch_msda_backward(const void *, const int64_t *, const int64_t *, const void *, const void *, const void *,
                         int, int, int, int, int, int, int, void *, void *, void *, hipStream_t);
void set_win_stamp_buffer(void *);
int launch_dyn_mask_head_fused(const float *, const float *, const float *, int, int, int, int, int, int, int, int, int,
                               void *, void *, unsigned char *, hipStream_t);
int launch_dyn_mask_head_mfma(const float *, const float *, const float *, int, int, int, int, int, int, int, int, int,
                              void *, void *, unsigned char *, hipStream_t);
int prepare_win_queue_device();
int prepare_msda_backward_col_device();
const char *msda_forward_col_build_flags();
const char *msda_backward_col_build_flags();
const char *ffn_fused_build_flags();
int prepare_ffn_device();
int launch_msda_forward_planes(const void *, const int64_t *, const int64_t *, const void *, const void *, int, int, int, int,
                               int, int, int, void *, hipStream_t, const float *, long long);
int launch_ffn_fused_split(const float *, long long, const float *, const float *, const float *, const float *, const float *,
                           const float *, float, int, long long, void *, float *, long long, hipStream_t);
int launch_conv1x1_nchw_split(const float *, const float *, const float *, unsigned short *, int, int, int, float *, hipStream_t);
int launch_conv1x1_groupnorm_tokens(const float *, const float *, const float *, unsigned short *, const float *, const float *, float,
                                    int, int, int, float *, float *, float *, long long, hipStream_t);
int launch_cross_attention(const void *, const void *, const void *, const void *, const void *, const unsigned char *,
                           const unsigned char *, int, int, int, int, float, void *, hipStream_t);
int launch_masked_attention(const void *, const void *, const void *, const unsigned char *, int, int, int, int, int,
                            int, float, int, void *, hipStream_t);
int launch_groupnorm_flatten(const float *, const float *, const float *, int, int, int, int, float, float *, float *,
                             long long, long long, hipStream_t);
int launch_lsap(const float *, int, int, int, const int *, int *, int *, hipStream_t);
int launch_add_layernorm(const float *, const float *, const float *, const float *, float, long long, int, float *,
                         hipStream_t);
template <typename T>
int launch_msda_forward_special(const void *, const int64_t *, const int64_t *, const void *, const void *, int, int,
                                int, int, int, int, int, void *, hipStream_t, const float *, long long);
int launch_dyn_mask_head(const float *, const float *, const float *, int, int, int, int, int, int, int, int, int,
                         int, void *, unsigned char *, hipStream_t);
}  // namespace pct

namespace {

int check_common(const void *value, const int64_t *shapes, const int64_t *starts, const void *loc,
                 const void *attn, int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                 size_t elem, size_t loc_elem)
{
  if (N < 0 || S < 0 || Lq < 0 || M <= 0 || D <= 0 || L <= 0 || P <= 0 || im2col_step <= 0) return PCT_ERR_BAD_ARG;
  if (N == 0) return PCT_OK;
  const int step = N < im2col_step ? N : im2col_step;
  if (N % step != 0) return PCT_ERR_IM2COL_STEP;  // cu:57
  if (Lq == 0) return PCT_OK;                     // nothing to sample: empty loc / attn may be null
  if (!value || !shapes || !starts || !loc || !attn) return PCT_ERR_BAD_ARG;
  // per-image offsets are 32-bit in the kernels (as in the reference, cuh:260-283)
  if ((long long)S * M * D >= 0x7fffffffLL) return PCT_ERR_UNSUPPORTED;
  if ((uintptr_t)value % elem || (uintptr_t)loc % loc_elem || (uintptr_t)attn % loc_elem ||
      (uintptr_t)shapes % 8 || (uintptr_t)starts % 8)
    return PCT_ERR_ALIGNMENT;
  return PCT_OK;
}

// largest number of images per launch whose value tensor stays below 2 GiB (at least 1)
int images_per_launch(int N, long long image_bytes)
{
  int chunk = N;
  while (chunk > 1 && (long long)chunk * image_bytes >= 0x7fffffffLL) chunk = (chunk + 1) / 2;
  return chunk;
}

template <typename T, typename LT>
int forward_impl(const void *value, const int64_t *shapes, const int64_t *starts, const LT *loc, const LT *attn,
                 int N, int S, int M, int D, int L, int Lq, int P, int im2col_step, void *out, void *stream)
{
  const int rc = check_common(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P, im2col_step,
                              sizeof(typename pct::Traits<T>::store_t), sizeof(LT));
  if (rc != PCT_OK) return rc;
  if (N == 0 || Lq == 0) return PCT_OK;
  if (!out) return PCT_ERR_BAD_ARG;
  if ((uintptr_t)out % sizeof(typename pct::Traits<T>::store_t)) return PCT_ERR_ALIGNMENT;
  // The specialised kernels address one launch's value tensor with 32-bit byte offsets: a batch whose value tensor
  // reaches 2 GiB goes out in chunks of images, as the reference does with im2col_step (cu:66-80) -- same results.
  using ST = typename pct::Traits<T>::store_t;
  const int chunk = images_per_launch(N, (long long)S * M * D * (long long)sizeof(ST));
  for (int b0 = 0; b0 < N; b0 += chunk) {
    const int n = N - b0 < chunk ? N - b0 : chunk;
    const int rc2 = pct::launch_msda_forward<T>(
        static_cast<const ST *>(value) + (long long)b0 * S * M * D, shapes, starts, loc + (long long)b0 * Lq * M * L * P * 2,
        attn + (long long)b0 * Lq * M * L * P, n, S, M, D, L, Lq, P, static_cast<ST *>(out) + (long long)b0 * Lq * M * D,
        static_cast<hipStream_t>(stream));
    if (rc2 != 0) return rc2;
  }
  return PCT_OK;
}

template <typename A>
int backward_impl(const A *value, const int64_t *shapes, const int64_t *starts, const A *loc, const A *attn,
                  const A *grad_out, int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                  A *grad_value, A *grad_loc, A *grad_attn, void *stream)
{
  const int rc = check_common(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P, im2col_step, sizeof(A),
                              sizeof(A));
  if (rc != PCT_OK) return rc;
  if (N == 0) return PCT_OK;
  if (!grad_value || !grad_loc || !grad_attn || (Lq > 0 && !grad_out)) return PCT_ERR_BAD_ARG;
  if ((uintptr_t)grad_out % sizeof(A) || (uintptr_t)grad_value % sizeof(A) || (uintptr_t)grad_loc % sizeof(A) ||
      (uintptr_t)grad_attn % sizeof(A))
    return PCT_ERR_ALIGNMENT;
  return pct::launch_msda_backward<A>(value, shapes, starts, loc, attn, grad_out, N, S, M, D, L, Lq, P,
                                      grad_value, grad_loc, grad_attn, static_cast<hipStream_t>(stream));
}

}  // namespace

extern "C" {

int pct_abi_version(void) { return PCT_ABI_VERSION; }

const char *pct_error_string(int code)
{
  switch (code) {
    case PCT_OK: return "ok";
    case PCT_ERR_BAD_ARG: return "bad argument (null pointer or non-positive size)";
    case PCT_ERR_IM2COL_STEP: return "batch must be divisible by min(batch, im2col_step)";
    case PCT_ERR_ALIGNMENT: return "pointer not aligned to its element size";
    case PCT_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "unknown error";
}

int pct_ms_deform_attn_forward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                   const float *sampling_loc, const float *attn_weight, int batch,
                                   int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                   int num_point, int im2col_step, float *output, void *stream)
{
  return forward_impl<float, float>(value, spatial_shapes, level_start, sampling_loc, attn_weight, batch,
                                    spatial_size, num_heads, channels, num_levels, num_query, num_point,
                                    im2col_step, output, stream);
}

int pct_ms_deform_attn_forward_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                   const double *sampling_loc, const double *attn_weight, int batch,
                                   int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                   int num_point, int im2col_step, double *output, void *stream)
{
  return forward_impl<double, double>(value, spatial_shapes, level_start, sampling_loc, attn_weight, batch,
                                      spatial_size, num_heads, channels, num_levels, num_query, num_point,
                                      im2col_step, output, stream);
}

int pct_ms_deform_attn_forward_f16(const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                   const float *sampling_loc, const float *attn_weight, int batch,
                                   int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                   int num_point, int im2col_step, void *output, void *stream)
{
  return forward_impl<pct::half_bits, float>(value, spatial_shapes, level_start, sampling_loc, attn_weight, batch,
                                             spatial_size, num_heads, channels, num_levels, num_query, num_point,
                                             im2col_step, output, stream);
}

int pct_ms_deform_attn_forward_bf16(const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                    const float *sampling_loc, const float *attn_weight, int batch,
                                    int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                    int num_point, int im2col_step, void *output, void *stream)
{
  return forward_impl<pct::bf16_bits, float>(value, spatial_shapes, level_start, sampling_loc, attn_weight, batch,
                                             spatial_size, num_heads, channels, num_levels, num_query, num_point,
                                             im2col_step, output, stream);
}

int pct_ms_deform_attn_fused_forward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                         const float *ref_points, long long ref_batch_stride, const float *offsets,
                                         const float *attn_logits, int batch, int spatial_size, int num_heads,
                                         int channels, int num_levels, int num_query, int num_point, float *output,
                                         void *stream)
{
  const int rc = check_common(value, spatial_shapes, level_start, offsets, attn_logits, batch, spatial_size, num_heads,
                              channels, num_levels, num_query, num_point, 1, sizeof(float), sizeof(float));
  if (rc != PCT_OK) return rc;
  if (batch == 0 || num_query == 0) return PCT_OK;
  if (!output || !ref_points || ref_batch_stride < 0) return PCT_ERR_BAD_ARG;
  if (((uintptr_t)output | (uintptr_t)ref_points) & 3u) return PCT_ERR_ALIGNMENT;
  const long long S = spatial_size, M = num_heads, D = channels, L = num_levels, Lq = num_query, P = num_point;
  const int chunk = images_per_launch(batch, S * M * D * 4);               // as forward_impl: < 2 GiB of value per launch
  for (int b0 = 0; b0 < batch; b0 += chunk) {
    const int n = batch - b0 < chunk ? batch - b0 : chunk;
    const int r = pct::launch_msda_forward_special<float>(
        value + b0 * S * M * D, spatial_shapes, level_start, offsets + b0 * Lq * M * L * P * 2, attn_logits + b0 * Lq * M * L * P,
        n, spatial_size, num_heads, channels, num_levels, num_query, num_point, output + b0 * Lq * M * D,
        static_cast<hipStream_t>(stream), ref_points + b0 * ref_batch_stride, ref_batch_stride);
    if (r != 0) return r == -100 ? PCT_ERR_UNSUPPORTED : r;
  }
  return PCT_OK;
}

int pct_ms_deform_attn_forward_planes_f32(const float *value_planes, const int64_t *spatial_shapes, const int64_t *level_start,
                                          const float *loc_planes, const float *attn_planes, const float *ref_points,
                                          long long ref_batch_stride, int batch, int spatial_size, int num_heads, int channels,
                                          int num_levels, int num_query, int num_point, float *output, void *stream)
{
  const int rc = check_common(value_planes, spatial_shapes, level_start, loc_planes, attn_planes, batch, spatial_size, num_heads,
                              channels, num_levels, num_query, num_point, 1, sizeof(float), sizeof(float));
  if (rc != PCT_OK) return rc;
  if (batch == 0 || num_query == 0) return PCT_OK;
  if (!output || ref_batch_stride < 0) return PCT_ERR_BAD_ARG;
  if (((uintptr_t)output | (uintptr_t)ref_points) & 3u) return PCT_ERR_ALIGNMENT;
  if (num_query != spatial_size || channels != 16 || num_point != 4) return PCT_ERR_UNSUPPORTED;
  const long long S = spatial_size, M = num_heads, D = channels, L = num_levels, Lq = num_query, P = num_point;
  const int chunk = images_per_launch(batch, S * M * D * 4);               // as forward_impl: < 2 GiB of value per launch
  for (int b0 = 0; b0 < batch; b0 += chunk) {
    const int n = batch - b0 < chunk ? batch - b0 : chunk;
    const int r = pct::launch_msda_forward_planes(
        value_planes + b0 * S * M * D, spatial_shapes, level_start, loc_planes + b0 * Lq * M * L * P * 2,
        attn_planes + b0 * Lq * M * L * P, n, spatial_size, num_heads, channels, num_levels, num_query, num_point,
        output + b0 * Lq * M * D, static_cast<hipStream_t>(stream), ref_points ? ref_points + b0 * ref_batch_stride : nullptr,
        ref_batch_stride);
    if (r != 0) return r == -100 ? PCT_ERR_UNSUPPORTED : r;
  }
  return PCT_OK;
}

int pct_ms_deform_attn_backward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                    const float *sampling_loc, const float *attn_weight, const float *grad_output,
                                    int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                    int num_query, int num_point, int im2col_step, float *grad_value,
                                    float *grad_sampling_loc, float *grad_attn_weight, void *stream)
{
  return backward_impl<float>(value, spatial_shapes, level_start, sampling_loc, attn_weight, grad_output, batch,
                              spatial_size, num_heads, channels, num_levels, num_query, num_point, im2col_step,
                              grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

int pct_ms_deform_attn_backward_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                    const double *sampling_loc, const double *attn_weight,
                                    const double *grad_output, int batch, int spatial_size, int num_heads,
                                    int channels, int num_levels, int num_query, int num_point, int im2col_step,
                                    double *grad_value, double *grad_sampling_loc, double *grad_attn_weight,
                                    void *stream)
{
  return backward_impl<double>(value, spatial_shapes, level_start, sampling_loc, attn_weight, grad_output, batch,
                               spatial_size, num_heads, channels, num_levels, num_query, num_point, im2col_step,
                               grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

/* diagnostic hook, not part of the product ABI: per-phase cycle stamps of the windowed MSDeformAttn kernel */
__attribute__((visibility("default"))) void pct_debug_set_stamp_buffer(void *p) { pct::set_win_stamp_buffer(p); }

int pct_dynamic_mask_head_forward(const float *mask_feat, const float *ref_points, const float *params, int batch,
                                  int channels, int num_query, int height, int width, int stride, int rel_coord,
                                  int target_h, int target_w, int out_dtype, void *up_logits,
                                  unsigned char *attn_mask, void *stream)
{
  if (batch < 0 || num_query < 0 || channels <= 0 || height <= 0 || width <= 0 || stride <= 0 || target_h <= 0 ||
      target_w <= 0)
    return PCT_ERR_BAD_ARG;
  if (batch == 0 || num_query == 0) return PCT_OK;
  if (!mask_feat || !params || !up_logits || !attn_mask || (rel_coord && !ref_points)) return PCT_ERR_BAD_ARG;
  if (((uintptr_t)mask_feat | (uintptr_t)params | (uintptr_t)ref_points) & 3u) return PCT_ERR_ALIGNMENT;
  if ((uintptr_t)up_logits & 15u) return PCT_ERR_ALIGNMENT;
  if ((long long)height * width >= 0x7fffffffLL / 64) return PCT_ERR_UNSUPPORTED;
  return pct::launch_dyn_mask_head(mask_feat, ref_points, params, batch, channels, num_query, height, width, stride,
                                   rel_coord, target_h, target_w, out_dtype, up_logits, attn_mask,
                                   static_cast<hipStream_t>(stream));
}

int pct_dynamic_mask_head_forward_mfma(const float *mask_feat, const float *ref_points, const float *params, int batch,
                                       int channels, int num_query, int height, int width, int stride, int rel_coord,
                                       int target_h, int target_w, void *scratch, void *up_logits,
                                       unsigned char *attn_mask, void *stream)
{
  if (batch < 0 || num_query < 0 || channels <= 0 || height <= 0 || width <= 0 || stride <= 0 || target_h <= 0 ||
      target_w <= 0)
    return PCT_ERR_BAD_ARG;
  if (batch == 0 || num_query == 0) return PCT_OK;
  if (!mask_feat || !params || !up_logits || !attn_mask || !scratch || (rel_coord && !ref_points)) return PCT_ERR_BAD_ARG;
  if (((uintptr_t)mask_feat | (uintptr_t)params | (uintptr_t)ref_points) & 3u) return PCT_ERR_ALIGNMENT;
  if (((uintptr_t)up_logits & 7u) || ((uintptr_t)scratch & 1u)) return PCT_ERR_ALIGNMENT;
  if ((long long)height * width >= 0x7fffffffLL / 64) return PCT_ERR_UNSUPPORTED;
  return pct::launch_dyn_mask_head_mfma(mask_feat, ref_points, params, batch, channels, num_query, height, width,
                                        stride, rel_coord, target_h, target_w, scratch, up_logits, attn_mask,
                                        static_cast<hipStream_t>(stream));
}

int pct_dynamic_mask_head_forward_fused_bf16(const float *mask_feat, const float *ref_points, const float *params, int batch,
                                             int channels, int num_query, int height, int width, int stride,
                                             int rel_coord, int target_h, int target_w, void *workspace,
                                             void *up_logits, unsigned char *attn_mask, void *stream)
{
  if (batch < 0 || num_query < 0 || channels <= 0 || height <= 0 || width <= 0 || stride <= 0 || target_h <= 0 ||
      target_w <= 0)
    return PCT_ERR_BAD_ARG;
  if (batch == 0 || num_query == 0) return PCT_OK;
  if (!mask_feat || !params || !up_logits || !attn_mask || !workspace || (rel_coord && !ref_points)) return PCT_ERR_BAD_ARG;
  if (((uintptr_t)mask_feat | (uintptr_t)params | (uintptr_t)ref_points) & 3u) return PCT_ERR_ALIGNMENT;
  if (((uintptr_t)up_logits | (uintptr_t)workspace) & 15u) return PCT_ERR_ALIGNMENT;
  if ((long long)height * width >= 0x7fffffffLL / 64) return PCT_ERR_UNSUPPORTED;
  const int r = pct::launch_dyn_mask_head_fused(mask_feat, ref_points, params, batch, channels, num_query, height, width,
                                                stride, rel_coord, target_h, target_w, workspace, up_logits, attn_mask,
                                                static_cast<hipStream_t>(stream));
  return r == -100 ? PCT_ERR_UNSUPPORTED : r;
}

int pct_add_layernorm_f32(const float *x, const float *y, const float *gamma, const float *beta, float eps,
                          long long rows, int cols, float *out, void *stream)
{
  if (rows < 0 || cols <= 0) return PCT_ERR_BAD_ARG;
  if (rows == 0) return PCT_OK;
  if (!x || !gamma || !beta || !out) return PCT_ERR_BAD_ARG;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15u) return PCT_ERR_ALIGNMENT;
  return pct::launch_add_layernorm(x, y, gamma, beta, eps, rows, cols, out, static_cast<hipStream_t>(stream));
}

int pct_linear_k128_f32(const float *x, long long ldx, const float *x_add, long long ld_add, long long add_period,
                        const float *w, const float *bias, long long rows, int n, int act, float *y, long long ldy,
                        void *stream)
{
  if (rows < 0 || n <= 0 || ldx < 128 || ldy < n || (act != 0 && act != 1)) return PCT_ERR_BAD_ARG;
  if (x_add && (ld_add < 128 || (ld_add & 3) || ((uintptr_t)x_add & 15u))) return PCT_ERR_ALIGNMENT;
  if (x_add && (add_period < 32 || add_period * ld_add * 4 > 0x7fffffffLL)) return PCT_ERR_BAD_ARG;
  if (rows == 0) return PCT_OK;
  if (!x || !w || !y) return PCT_ERR_BAD_ARG;
  if (n % 32) return PCT_ERR_UNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)w) & 15u) || (ldx & 3)) return PCT_ERR_ALIGNMENT;
  return pct::launch_linear_k128(x, ldx, x_add, ld_add, add_period, w, bias, rows, n, act, y, ldy, nullptr, 0, nullptr, nullptr, 0.f,
                                 static_cast<hipStream_t>(stream));
}

void pct_msda_set_kernel_choice(int choice) { pct::set_msda_kernel_choice(choice); }

int pct_msda_last_kernel(void) { return pct::msda_last_kernel(); }

void pct_msda_set_bwd_kernel_choice(int choice) { pct::set_msda_bwd_kernel_choice(choice); }

int pct_msda_last_bwd_kernel(void) { return pct::msda_bwd_last_kernel(); }

int pct_linear_k128_multi_f32(const float *x, long long ldx, const float *x_add, long long ld_add, long long add_period,
                              int nseg, const float *const *w, const float *const *bias, const int *n, const int *use_add,
                              float *const *y, const long long *ldy, long long rows, void *stream)
{
  if (rows < 0 || nseg < 1 || nseg > 4 || ldx < 128 || !w || !bias || !n || !use_add || !y || !ldy) return PCT_ERR_BAD_ARG;
  if (x_add && (ld_add < 128 || (ld_add & 3) || ((uintptr_t)x_add & 15u))) return PCT_ERR_ALIGNMENT;
  if (x_add && (add_period < 32 || add_period * ld_add * 4 > 0x7fffffffLL)) return PCT_ERR_BAD_ARG;
  if (rows == 0) return PCT_OK;
  if (!x) return PCT_ERR_BAD_ARG;
  if (((uintptr_t)x & 15u) || (ldx & 3)) return PCT_ERR_ALIGNMENT;
  for (int s = 0; s < nseg; ++s) {
    if (!w[s] || !y[s] || n[s] <= 0 || ldy[s] < n[s]) return PCT_ERR_BAD_ARG;
    if (n[s] % 32) return PCT_ERR_UNSUPPORTED;
  }
  const int rc = pct::launch_linear_k128_split_multi(x, ldx, x_add, ld_add, add_period, nseg, w, bias, n, use_add, y, ldy, rows,
                                                     static_cast<hipStream_t>(stream));
  return rc == -100 ? PCT_ERR_ALIGNMENT : (rc == -4 ? PCT_ERR_UNSUPPORTED : rc);
}

int pct_linear_k128_add_layernorm_f32(const float *x, long long ldx, const float *w, const float *bias,
                                      const float *residual, long long ldr, const float *gamma, const float *beta,
                                      float eps, long long rows, float *out, long long ldo, void *stream)
{
  if (rows < 0 || ldx < 128 || ldr < 128 || ldo < 128) return PCT_ERR_BAD_ARG;
  if (rows == 0) return PCT_OK;
  if (!x || !w || !residual || !gamma || !beta || !out) return PCT_ERR_BAD_ARG;
  if ((((uintptr_t)x | (uintptr_t)w) & 15u) || (ldx & 3)) return PCT_ERR_ALIGNMENT;
  return pct::launch_linear_k128(x, ldx, nullptr, 0, 0, w, bias, rows, 128, 2, out, ldo, residual, ldr, gamma, beta, eps,
                                 static_cast<hipStream_t>(stream));
}

int pct_linear_add_layernorm_f32(const float *x, long long ldx, int k, const float *w, void *w_split_ws, const float *bias,
                                 const float *residual, long long ldr, const float *gamma, const float *beta, float eps,
                                 long long rows, float *out, long long ldo, void *stream)
{
  if (rows < 0 || k <= 0 || ldx < k || ldr < 128 || ldo < 128) return PCT_ERR_BAD_ARG;
  if (rows == 0) return PCT_OK;
  if (!x || !w || !w_split_ws || !residual || !gamma || !beta || !out) return PCT_ERR_BAD_ARG;
  if (k % 32) return PCT_ERR_UNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)w_split_ws | (uintptr_t)residual | (uintptr_t)out | (uintptr_t)bias |
        (uintptr_t)gamma | (uintptr_t)beta) & 15u) || (ldx & 3) || (ldr & 3) || (ldo & 3))
    return PCT_ERR_ALIGNMENT;
  const int rc = pct::launch_linear_ln_split(x, ldx, w, static_cast<unsigned short *>(w_split_ws), k, bias, residual, ldr, gamma,
                                             beta, eps, rows, out, ldo, static_cast<hipStream_t>(stream));
  return rc == -4 ? PCT_ERR_UNSUPPORTED : rc;
}

int pct_groupnorm_flatten_f32(const float *x, const float *gamma, const float *beta, int batch, int channels, int hw,
                              int groups, float eps, float *stats, float *out, long long out_batch_stride,
                              long long out_offset, void *stream)
{
  if (batch < 0 || channels <= 0 || hw < 0 || groups <= 0 || out_batch_stride < 0 || out_offset < 0) return PCT_ERR_BAD_ARG;
  if (batch == 0 || hw == 0) return PCT_OK;
  if (!x || !gamma || !beta || !stats || !out) return PCT_ERR_BAD_ARG;
  if (channels != 128 || channels % groups || (channels / groups) % 4) return PCT_ERR_UNSUPPORTED;
  if ((uintptr_t)x & 15u) return PCT_ERR_ALIGNMENT;
  return pct::launch_groupnorm_flatten(x, gamma, beta, batch, channels, hw, groups, eps, stats, out, out_batch_stride,
                                       out_offset, static_cast<hipStream_t>(stream));
}

int pct_conv1x1_nchw_f32(const float *x, const float *w, const float *bias, void *w_split_ws, int batch, int in_channels,
                         int out_channels, int hw, float *out, void *stream)
{
  if (batch < 0 || in_channels <= 0 || out_channels <= 0 || hw < 0) return PCT_ERR_BAD_ARG;
  if (batch == 0 || hw == 0) return PCT_OK;
  if (!x || !w || !w_split_ws || !out) return PCT_ERR_BAD_ARG;
  if (out_channels != 128) return PCT_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)w_split_ws | (uintptr_t)out | (uintptr_t)bias) & 15u) return PCT_ERR_ALIGNMENT;
  const int rc = pct::launch_conv1x1_nchw_split(x, w, bias, static_cast<unsigned short *>(w_split_ws), batch, in_channels, hw, out,
                                                static_cast<hipStream_t>(stream));
  return rc == -4 ? PCT_ERR_UNSUPPORTED : rc;
}

int pct_conv1x1_groupnorm_tokens_f32(const float *x, const float *w, const float *bias, void *w_split_ws, const float *gamma,
                                     const float *beta, int groups, float eps, int batch, int in_channels, int out_channels, int hw,
                                     float *partial_ws, float *stats, float *out, long long out_batch_stride, long long out_offset,
                                     void *stream)
{
  if (batch < 0 || in_channels <= 0 || out_channels <= 0 || hw < 0 || out_batch_stride < 0 || out_offset < 0) return PCT_ERR_BAD_ARG;
  if (batch == 0 || hw == 0) return PCT_OK;
  if (!x || !w || !w_split_ws || !gamma || !beta || !partial_ws || !stats || !out) return PCT_ERR_BAD_ARG;
  if (out_channels != 128 || groups != 32) return PCT_ERR_UNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)w_split_ws | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)gamma | (uintptr_t)beta |
        (uintptr_t)partial_ws | (uintptr_t)stats) & 15u) || (out_batch_stride & 3) || (out_offset & 3))
    return PCT_ERR_ALIGNMENT;
  const int rc = pct::launch_conv1x1_groupnorm_tokens(x, w, bias, static_cast<unsigned short *>(w_split_ws), gamma, beta, eps, batch,
                                                      in_channels, hw, partial_ws, stats, out + out_offset, out_batch_stride,
                                                      static_cast<hipStream_t>(stream));
  return rc == -4 ? PCT_ERR_UNSUPPORTED : rc;
}

int pct_ffn_layernorm_f32(const float *x, long long ldx, const float *w1, const float *b1, const float *w2, const float *b2,
                          const float *gamma, const float *beta, float eps, int hidden, long long rows, void *w_image_ws, float *out,
                          long long ldo, void *stream)
{
  if (rows < 0 || hidden <= 0 || ldx < 128 || ldo < 128) return PCT_ERR_BAD_ARG;
  if (rows == 0) return PCT_OK;
  if (!x || !w1 || !b1 || !w2 || !gamma || !beta || !w_image_ws || !out) return PCT_ERR_BAD_ARG;
  if (hidden % 32) return PCT_ERR_UNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)b1 | (uintptr_t)w2 | (uintptr_t)b2 | (uintptr_t)gamma | (uintptr_t)beta |
        (uintptr_t)w_image_ws | (uintptr_t)out) & 15u) || (ldx & 3) || (ldo & 3))
    return PCT_ERR_ALIGNMENT;
  const int rc = pct::launch_ffn_fused_split(x, ldx, w1, b1, w2, b2, gamma, beta, eps, hidden, rows, w_image_ws, out, ldo,
                                             static_cast<hipStream_t>(stream));
  return rc == -4 ? PCT_ERR_UNSUPPORTED : rc;
}

int pct_lsap_f32(const float *cost, int batch, int num_query, int ld_target, const int *num_target, int *row_for_target,
                 int *status, void *stream)
{
  if (batch < 0 || num_query <= 0 || ld_target <= 0) return PCT_ERR_BAD_ARG;
  if (batch == 0) return PCT_OK;
  if (!cost || !num_target || !row_for_target || !status) return PCT_ERR_BAD_ARG;
  const int rc = pct::launch_lsap(cost, batch, num_query, ld_target, num_target, row_for_target, status,
                                  static_cast<hipStream_t>(stream));
  return rc == -4 ? PCT_ERR_UNSUPPORTED : rc;
}

int pct_masked_attention_bf16(const void *q, const void *k, const void *vT, const unsigned char *mask, int batch,
                              int heads, int num_query, int num_key, int head_dim, int v_head_dim, float scale,
                              int out_dtype, void *out, void *stream)
{
  if (batch < 0 || heads <= 0 || num_query < 0 || num_key <= 0) return PCT_ERR_BAD_ARG;
  if (batch == 0 || num_query == 0) return PCT_OK;
  if (!q || !k || !vT || !out) return PCT_ERR_BAD_ARG;
  if (((uintptr_t)q | (uintptr_t)k) & 15u) return PCT_ERR_ALIGNMENT;
  if (((uintptr_t)vT | (uintptr_t)out) & 7u) return PCT_ERR_ALIGNMENT;
  return pct::launch_masked_attention(q, k, vT, mask, batch, heads, num_query, num_key, head_dim, v_head_dim, scale,
                                      out_dtype, out, static_cast<hipStream_t>(stream));
}

int pct_prepare_device(void)
{
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)cap;
  const int a = pct::prepare_win_queue_device();
  const int b = pct::prepare_msda_backward_col_device();
  const int c = pct::prepare_ffn_device();
  return a ? a : (b ? b : c);
}

const char *pct_build_info(void)
{
  static char buf[768];
  static const bool once = [] {
#ifdef PCT_EXPERIMENT_BUILD
    const char *exp = "1";
#else
    const char *exp = "0";
#endif
    snprintf(buf, sizeof(buf), "experiment=%s; target=gfx950; %s; %s; %s", exp, pct::msda_forward_col_build_flags(),
             pct::msda_backward_col_build_flags(), pct::ffn_fused_build_flags());
    return true;
  }();
  (void)once;
  return buf;
}

int pct_cross_attention_bf16(const void *q_content, const void *q_pos, const void *k_content, const void *k_pos, const void *v,
                             const unsigned char *mask, const unsigned char *row_open, int batch, int heads, int num_query,
                             int num_key, float scale, void *out, void *stream)
{
  if (batch < 0 || heads <= 0 || num_query < 0 || num_key <= 0) return PCT_ERR_BAD_ARG;
  if (batch == 0 || num_query == 0) return PCT_OK;
  if (!q_content || !q_pos || !k_content || !k_pos || !v || !out) return PCT_ERR_BAD_ARG;
  if (((uintptr_t)q_content | (uintptr_t)q_pos | (uintptr_t)k_content | (uintptr_t)k_pos | (uintptr_t)v) & 15u)
    return PCT_ERR_ALIGNMENT;
  if ((uintptr_t)out & 7u) return PCT_ERR_ALIGNMENT;
  const int r = pct::launch_cross_attention(q_content, q_pos, k_content, k_pos, v, mask, mask ? row_open : nullptr, batch, heads,
                                            num_query, num_key, scale, out, static_cast<hipStream_t>(stream));
  return r == -100 ? PCT_ERR_UNSUPPORTED : r;
}

}  // extern "C"
