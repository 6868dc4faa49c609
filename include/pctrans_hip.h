/*
 * pctrans_hip.h -- C ABI of libpctrans_hip.so, the MI355X (gfx950) drop-in for the native side of
 * PCTrans' Mask2Former-style decoder hot path.
 *
 * Boundary being replaced (paths relative to the reference checkout,
 * OPS = connectomics/model/maskformer_block/pixel_decoder/ops):
 *   OPS/src/vision.cpp:18-21            pybind module `MultiScaleDeformableAttention`
 *   OPS/src/ms_deform_attn.h:25-67      ms_deform_attn_forward / ms_deform_attn_backward dispatch
 *   OPS/src/cuda/ms_deform_attn_cuda.cu:25-85, 88-158   host wrappers (checks, alloc, chunk loop)
 *   OPS/src/cuda/ms_deform_im2col_cuda.cuh:242-304      forward kernel, 306-1331 backward kernels
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / ATen types cross this line.
 *   - every pointer is a DEVICE pointer (HBM) unless its name starts with `host_`.
 *   - tensors are dense row-major ("contiguous") exactly as the reference requires (cu:33-37):
 *       value          [batch, spatial_size, num_heads, channels]
 *       spatial_shapes [num_levels, 2]  int64, (H_l, W_l)
 *       level_start    [num_levels]     int64, prefix sums of H_l*W_l
 *       sampling_loc   [batch, num_query, num_heads, num_levels, num_point, 2]  (x, y) normalised to [0,1]
 *       attn_weight    [batch, num_query, num_heads, num_levels, num_point]
 *       output         [batch, num_query, num_heads*channels]
 *   - inputs are borrowed and never written; outputs are caller-allocated and fully overwritten
 *     (the caller does NOT need to zero them -- the reference's at::zeros, cu:59/126-128, is folded in).
 *   - work is enqueued on `stream` (a hipStream_t, may be NULL = default stream); no host sync.
 *   - return value: 0 on success; >0 = a hipError_t from the launch; <0 = one of PCT_ERR_* below.
 *     Unlike the reference (which only printf()s launch errors, cuh:953-957) errors are returned.
 *   - `im2col_step` keeps the reference's precondition batch % min(batch, im2col_step) == 0 (cu:55-57) and is
 *     otherwise unused: a batch is ONE launch unless its value tensor reaches 2 GiB (32-bit byte offsets inside a
 *     launch); such a batch is sent out in chunks of whole images on the same kernel and stream, as the reference
 *     does with im2col_step (cu:66-80).  Chunking does not change results.
 */
#ifndef PCTRANS_HIP_H_
#define PCTRANS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCT_ABI_VERSION 1

/* exported with default visibility; everything else in the library is hidden */
#define PCT_API __attribute__((visibility("default")))

#define PCT_OK 0
#define PCT_ERR_BAD_ARG (-1)      /* null pointer, non-positive size, sizes that overflow int32 indexing */
#define PCT_ERR_IM2COL_STEP (-2)  /* batch % min(batch, im2col_step) != 0 (cu:57) */
#define PCT_ERR_ALIGNMENT (-3)    /* a pointer is not aligned to its element size */
#define PCT_ERR_UNSUPPORTED (-4)  /* shape outside what the kernel supports (documented per entry point) */

PCT_API int pct_abi_version(void);
/* static string for a code returned by any entry point (hipGetErrorString for >0) */
PCT_API const char *pct_error_string(int code);
/* static string naming what this library was built with: "experiment=0|1; target=gfx950; col: KO=00000 ...; bcol: KO=0 ...".
 * The kernels carry compile-time A/B switches, some of them knock-outs that give WRONG RESULTS by design (timing
 * experiments); those only compile with -DPCT_EXPERIMENT_BUILD, which shows here as experiment=1 (tests/test_abi.py
 * asserts experiment=0 and all knock-outs off for the library the tests run on). */
PCT_API const char *pct_build_info(void);
/* Allocates, for the CURRENT device, the small device-side pools some kernels keep for the life of the process (work-queue
 * counters of the persistent MSDeformAttn kernels: 136 KB; per-item flag buffers of the pyramid-column backward: 64 MB) and sets
 * the fused FFN kernel's dynamic-LDS attribute.
 * Optional -- the first launch that needs a pool allocates it lazily -- but that lazy path allocates and synchronises the
 * device, which must not happen while ANY stream of the process is being captured into a HIP graph (a launch that finds its
 * pool missing under capture runs on another kernel instead: pct_msda_last_kernel / pct_msda_last_bwd_kernel tell).  Call it
 * once per device before capturing; pctrans_amd's Python wrappers do so at their first call on a device.  0 on success. */
PCT_API int pct_prepare_device(void);

/* ---- MSDeformAttn forward: replaces ms_deform_attn_cuda_forward (cu:25-85) ------------------------------- */
/* fp32 / fp64: the two dtypes the reference dispatches (AT_DISPATCH_FLOATING_TYPES, cu:69). */
PCT_API int pct_ms_deform_attn_forward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                   const float *sampling_loc, const float *attn_weight, int batch,
                                   int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                   int num_point, int im2col_step, float *output, void *stream);
PCT_API int pct_ms_deform_attn_forward_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                   const double *sampling_loc, const double *attn_weight, int batch,
                                   int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                   int num_point, int im2col_step, double *output, void *stream);
/* New capability (not in the reference): 16-bit value/output with fp32 locations, weights and accumulation --
 * the dtypes torch autocast produces at this boundary (Linear -> bf16/f16 value; softmax and the
 * reference-point add -> fp32).  `value`/`output` are raw IEEE half / bfloat16 bit patterns. */
PCT_API int pct_ms_deform_attn_forward_f16(const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                   const float *sampling_loc, const float *attn_weight, int batch,
                                   int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                   int num_point, int im2col_step, void *output, void *stream);
PCT_API int pct_ms_deform_attn_forward_bf16(const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                    const float *sampling_loc, const float *attn_weight, int batch,
                                    int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                    int num_point, int im2col_step, void *output, void *stream);

/* ---- MSDeformAttn forward with the module front-end fused in (new entry point, same result) ----------------
 * Replaces, in one launch, ops/modules/ms_deform_attn.py:100-118 + the op above:
 *     attention_weights = softmax(attn_logits over num_levels*num_point)
 *     sampling_locations = ref_points[:, :, None, :, None, :] + offsets / (W_l, H_l)        (2-d reference points)
 *     output = ms_deform_attn_forward(value, ..., sampling_locations, attention_weights)
 *   ref_points  [batch or 1, num_query, num_levels, 2] fp32, (x, y) in [0,1]; ref_batch_stride = elements between
 *               images (0 when one table is shared by the whole batch, as in PCTrans' encoder)
 *   offsets     [batch, num_query, num_heads, num_levels, num_point, 2] fp32 (output of the sampling_offsets Linear)
 *   attn_logits [batch, num_query, num_heads, num_levels*num_point]     fp32 (output of the attention_weights Linear)
 * Supported geometry: channels == 16, num_point in {4, 8}; otherwise PCT_ERR_UNSUPPORTED (callers use the unfused op). */
PCT_API int pct_ms_deform_attn_fused_forward_f32(const float *value, const int64_t *spatial_shapes,
                                                 const int64_t *level_start, const float *ref_points,
                                                 long long ref_batch_stride, const float *offsets,
                                                 const float *attn_logits, int batch, int spatial_size, int num_heads,
                                                 int channels, int num_levels, int num_query, int num_point,
                                                 float *output, void *stream);

/* ---- MSDeformAttn forward on PIECE-PLANE operands (new entry point, same result bit for bit) ------------------
 * An MI355X-native operand layout for callers that control the producers (PCTrans' encoder layer does: the three
 * projections that feed the op, ops/modules/ms_deform_attn.py:96-110, are written by pct_linear_k128_planes_f32 below).
 * A "piece" is four consecutive floats of a (query, head) record; per image a tensor is stored as
 *     planes[num_heads * pieces][spatial_size][4]        plane = head * pieces + piece
 *   value_planes  pieces = channels / 4      (value[n][s][head][4 * piece ..])
 *   loc_planes    pieces = num_levels * num_point * 2 / 4   (sampling_loc[n][q][head] flattened, or with ref_points: offsets)
 *   attn_planes   pieces = num_levels * num_point / 4       (attn_weight[n][q][head] flattened, or with ref_points: logits)
 *   ref_points    NULL: loc / attn are sampling locations and attention weights (the op above);
 *                 else [batch or 1, num_query, num_levels, 2] and loc / attn are raw offsets and logits (the fused entry)
 *   output        [batch, num_query, num_heads * channels], as above
 * so that the 64 lanes of a wavefront read whole 128-byte lines wherever they read (records, window staging) and no
 * register transposition is needed.  Supported: what the pyramid-column kernel covers -- num_query == spatial_size,
 * channels == 16, num_point == 4, 3 <= num_levels <= 5, 16-byte aligned tensors; otherwise PCT_ERR_UNSUPPORTED (callers
 * keep the reference layout and use the entries above). */
PCT_API int pct_ms_deform_attn_forward_planes_f32(const float *value_planes, const int64_t *spatial_shapes,
                                                  const int64_t *level_start, const float *loc_planes,
                                                  const float *attn_planes, const float *ref_points,
                                                  long long ref_batch_stride, int batch, int spatial_size, int num_heads,
                                                  int channels, int num_levels, int num_query, int num_point,
                                                  float *output, void *stream);

/* ---- MSDeformAttn backward: replaces ms_deform_attn_cuda_backward (cu:88-158) ----------------------------- */
/* grad_value [as value], grad_sampling_loc [as sampling_loc], grad_attn_weight [as attn_weight]; all three are
 * fully defined on return (grad_value is zero-filled on `stream` by a kernel of the library before the scatter-add -- not
 * by hipMemsetAsync, whose node did not replay reliably from a torch.cuda.graph capture on ROCm 7.2).  A call is two or three launches on
 * `stream` (zero-fill, the kernel, and for the pyramid-column kernel a second launch for levels whose windows do not fit,
 * which ends at once when there are none); it may be recorded into a HIP graph and replayed.  The pyramid-column kernel keeps
 * one 64 MB block of per-item flag buffers per device for the life of the process (allocated at the first call outside stream
 * capture; a first call under capture, or more than 56 captured calls, run on the windowed kernel instead). */
PCT_API int pct_ms_deform_attn_backward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                    const float *sampling_loc, const float *attn_weight, const float *grad_output,
                                    int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                    int num_query, int num_point, int im2col_step, float *grad_value,
                                    float *grad_sampling_loc, float *grad_attn_weight, void *stream);
PCT_API int pct_ms_deform_attn_backward_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                    const double *sampling_loc, const double *attn_weight,
                                    const double *grad_output, int batch, int spatial_size, int num_heads,
                                    int channels, int num_levels, int num_query, int num_point, int im2col_step,
                                    double *grad_value, double *grad_sampling_loc, double *grad_attn_weight,
                                    void *stream);

/* ---- fused per-query dynamic mask head: replaces dynamic_mask_with_coords + mask_heads_forward ---------------
 * (transformer_decoder/mask2former_transformer_decoder.py:647-719; compute_locations / parse_dynamic_params :929-979)
 *   mask_feat   [batch, channels, height, width] fp32 (channels must be 16 = MODEL.SEM_SEG_HEAD.MASK_DIM)
 *   ref_points  [batch, num_query, 2] fp32, normalised (x, y)
 *   params      [batch, num_query, G] fp32, G = (channels + 2*rel_coord)*8 + 64 + 8 + 8 + 8 + 1, laid out as
 *               parse_dynamic_params splits the controller output: w0 | w1 | w2 | b0 | b1 | b2
 *   up_logits   [batch, num_query, 2*height, 2*width]  bilinear x2 (align_corners = False); out_dtype 0 = fp32,
 *               2 = bfloat16 (logits are then rounded to bf16 before resizing, as the reference's bf16-autocast
 *               convolutions do)
 *   attn_mask   [batch, num_query, target_h*target_w] bytes, 1 where sigmoid(resized logit) < 0.5 (= may not attend);
 *               the reference repeats this per head (:689-691), callers broadcast instead.
 * Returns PCT_ERR_UNSUPPORTED for channels != 16 (callers then use their dense formulation). */
PCT_API int pct_dynamic_mask_head_forward(const float *mask_feat, const float *ref_points, const float *params,
                                          int batch, int channels, int num_query, int height, int width, int stride,
                                          int rel_coord, int target_h, int target_w, int out_dtype, void *up_logits,
                                          unsigned char *attn_mask, void *stream);

/* ---- dynamic mask head, bf16-autocast configuration, on MFMA ----------------------------------------------
 * Same contract as pct_dynamic_mask_head_forward with out_dtype = 2 (bfloat16 outputs), computed as two launches:
 * the per-query 3-layer MLP on v_mfma_f32_16x16x16_bf16 (features / generated weights / hidden activations in bf16,
 * fp32 accumulation, relative-coordinate inputs and biases in fp32) and a streaming x2-upsample + attention-mask pass.
 * `scratch` is a caller-owned workspace of batch*num_query*height*width bfloat16 elements (the logits at feature
 * resolution); nothing is allocated inside the call. */
PCT_API int pct_dynamic_mask_head_forward_mfma(const float *mask_feat, const float *ref_points, const float *params,
                                               int batch, int channels, int num_query, int height, int width,
                                               int stride, int rel_coord, int target_h, int target_w, void *scratch,
                                               void *up_logits, unsigned char *attn_mask, void *stream);

/* ---- dynamic mask head, bf16-autocast configuration, one pass over the pixels ------------------------------------
 * Same contract and bit-identical outputs (for finite logits) as pct_dynamic_mask_head_forward_mfma, without its logits
 * plane: MLP, bf16 rounding, x2 upsample and attention mask in one kernel, so the [batch, num_query, height, width]
 * logits are never written to memory (csrc/dyn_mask_head_fused.hip; a small launch in front of it re-packs the generated
 * parameters).  `workspace`: caller-owned, batch * ceil(num_query / 2) * 5120 bytes, 16-byte aligned (the re-packed
 * parameters; nothing is allocated inside the call).  Geometry: channels == 16, width == 128, height % 8 == 0,
 * (target_h, target_w) = (height, width) / s with s in {2, 4, 8}; anything else returns PCT_ERR_UNSUPPORTED and the
 * caller uses pct_dynamic_mask_head_forward_mfma. */
PCT_API int pct_dynamic_mask_head_forward_fused_bf16(const float *mask_feat, const float *ref_points, const float *params,
                                                     int batch, int channels, int num_query, int height, int width,
                                                     int stride, int rel_coord, int target_h, int target_w,
                                                     void *workspace, void *up_logits, unsigned char *attn_mask,
                                                     void *stream);

/* ---- fused residual add + LayerNorm:  out = LayerNorm(x + y) * gamma + beta  over the last dimension ----------
 * Replaces the `x + dropout(y)` / `nn.LayerNorm` pairs of the encoder and decoder layers in eval mode
 * (pixel_decoder/msdeformattn.py:116-131; transformer_decoder/mask2former_transformer_decoder.py:97-99, 179-181,
 * 216-224).  x, y, out: [rows, cols] fp32 row-major, 16-byte aligned; y may be NULL (plain LayerNorm);
 * cols in {64, 128, 256}, else PCT_ERR_UNSUPPORTED.  out may alias x or y. */
PCT_API int pct_add_layernorm_f32(const float *x, const float *y, const float *gamma, const float *beta, float eps,
                                  long long rows, int cols, float *out, void *stream);

/* ---- GroupNorm of an NCHW map written into the encoder's token-major buffer ------------------------------------
 * out[n, p, c] (at out + n*out_batch_stride + out_offset + p*channels + c) = GroupNorm(x)[n, c, p] * gamma[c] + beta[c].
 * Replaces nn.GroupNorm(32, conv_dim) of the pixel decoder's input projections + `.flatten(2).transpose(1, 2)` +
 * `torch.cat(src_flatten, 1)` (pixel_decoder/msdeformattn.py:75-83, 220-224).  x [batch, channels, hw] fp32 16-byte
 * aligned; channels == 128 and (channels / groups) % 4 == 0, else PCT_ERR_UNSUPPORTED; stats = scratch of
 * batch*groups*2 floats (receives mean, rstd); strides / offsets in elements. */
PCT_API int pct_groupnorm_flatten_f32(const float *x, const float *gamma, const float *beta, int batch, int channels,
                                      int hw, int groups, float eps, float *stats, float *out,
                                      long long out_batch_stride, long long out_offset, void *stream);

/* ---- skinny fp32 projection GEMMs (K = 128) on the fp32 MFMA path ------------------------------------------------
 * pct_linear_k128_f32:  y[rows, n] = act((x + x_add)[rows, 128] . w[n, 128]^T + bias[n]),  act 0 = none, 1 = ReLU;
 *   x_add may be NULL (it is the encoder's `with_pos_embed(src, pos)`, pixel_decoder/msdeformattn.py:112-114); it has
 *   add_period >= 32 rows and repeats: row i of x pairs with row i % add_period (pos is the same for every image).
 *   Replaces nn.Linear for MSDeformAttn's value_proj / sampling_offsets / attention_weights
 *   (ops/modules/ms_deform_attn.py:64-67, 96-103) and the encoder FFN's linear1 + ReLU
 *   (pixel_decoder/msdeformattn.py:103-106, 126).
 * pct_linear_k128_add_layernorm_f32:  out[rows, 128] = LayerNorm(residual + x . w[128, 128]^T + bias) * gamma + beta.
 *   Replaces output_proj followed by `src = src + dropout1(src2); src = norm1(src)` (ops/modules/ms_deform_attn.py:123,
 *   pixel_decoder/msdeformattn.py:116-119), eval mode (dropout = identity).
 * All matrices fp32 row-major; ldx / ldy / ldr = row strides in elements (>= 128 / n / 128); x, w 16-byte aligned with
 * ldx % 4 == 0; n % 32 == 0 (else PCT_ERR_UNSUPPORTED); bias may be NULL in the first form; out may alias residual. */
PCT_API int pct_linear_k128_f32(const float *x, long long ldx, const float *x_add, long long ld_add, long long add_period,
                                const float *w, const float *bias, long long rows, int n, int act, float *y,
                                long long ldy, void *stream);
/* Diagnostics (tests and A/B measurements; not part of the reference's interface).
 * pct_msda_set_kernel_choice: forward kernel selection for this process, overriding PCT_MSDA_KERNEL: 0 = auto (fp32,
 *   Lq == S, 4 points and at least two work items per CU: pyramid-column kernel; else the windowed-LDS kernel when
 *   Lq == S and the problem gives its persistent grid about three items per workgroup; else quad-owner), 1 = windowed,
 *   2 = generic, 3 = quad-owner, 4 = pyramid-column, anything else = follow the environment again.  A forced kernel
 *   that does not cover the call's geometry falls through as in auto.  Results do not depend on the choice.
 * pct_msda_last_kernel: which forward kernel the most recent pct_ms_deform_attn_*forward* call of this process
 *   launched: 1 = windowed, 2 = generic, 3 = quad-owner, 4 = pyramid-column, 0 = none yet.  Both are process-wide
 *   atomics; concurrent callers see each other's values. */
PCT_API void pct_msda_set_kernel_choice(int choice);
PCT_API int pct_msda_last_kernel(void);
/* The same for the backward (pct_ms_deform_attn_backward_f32): choice 0 = auto (fp32, Lq == S, D = 16, 4 points and enough
 *   columns to fill the chip: pyramid-column kernel, msda_backward_col.hip; else the windowed kernel; else generic),
 *   1 = windowed, 2 = generic, 3 = pyramid-column (also below the size threshold), anything else = follow
 *   PCT_MSDA_BWD_KERNEL (auto | win | generic | col) again.  pct_msda_last_bwd_kernel: 1 = windowed, 2 = generic,
 *   3 = pyramid-column, 0 = none yet. */
PCT_API void pct_msda_set_bwd_kernel_choice(int choice);
PCT_API int pct_msda_last_bwd_kernel(void);

/* pct_linear_k128_multi_f32: nseg (1..4) Linear layers over the SAME rows in one launch, y[s] = (x [+ x_add]) . w[s]^T
 *   + bias[s]: MSDeformAttn's value_proj(src), sampling_offsets(src + pos) and attention_weights(src + pos)
 *   (ops/modules/ms_deform_attn.py:96-103) read the rows once instead of three times.  w, bias, n, use_add, y, ldy are
 *   HOST arrays of nseg entries (device pointers / sizes); use_add[s] != 0 adds x_add to the rows of segment s;
 *   n[s] % 32 == 0; w[s], bias[s], y[s] 16-byte aligned, ldy[s] % 4 == 0 (else PCT_ERR_ALIGNMENT). */
PCT_API int pct_linear_k128_multi_f32(const float *x, long long ldx, const float *x_add, long long ld_add,
                                      long long add_period, int nseg, const float *const *w, const float *const *bias,
                                      const int *n, const int *use_add, float *const *y, const long long *ldy,
                                      long long rows, void *stream);
PCT_API int pct_linear_k128_add_layernorm_f32(const float *x, long long ldx, const float *w, const float *bias,
                                              const float *residual, long long ldr, const float *gamma,
                                              const float *beta, float eps, long long rows, float *out,
                                              long long ldo, void *stream);

/* ---- Linear (any K) + residual + LayerNorm(128), fp32 -------------------------------------------------------------
 * pct_linear_add_layernorm_f32:  out[rows, 128] = LayerNorm(residual + x[rows, k] . w[128, k]^T + bias) * gamma + beta.
 *   Replaces the encoder FFN's `src2 = linear2(...)`, `src = src + dropout3(src2)`, `src = norm2(src)`
 *   (pixel_decoder/msdeformattn.py:122-131; dim_feedforward = 1024), eval mode (dropout = identity).
 *   fp32-accurate product on the bf16 matrix cores from exact three-way bf16 splits of both operands (see
 *   pctrans_amd/csrc/linear_ln_split.hip).  k % 32 == 0 (else PCT_ERR_UNSUPPORTED); x, w, residual, out, bias, gamma,
 *   beta 16-byte aligned, ldx / ldr / ldo % 4 == 0; w_split_ws = device workspace of 3 * 128 * k * 2 bytes (16-byte
 *   aligned) that the call overwrites with the split weights; out may alias residual. */
PCT_API int pct_linear_add_layernorm_f32(const float *x, long long ldx, int k, const float *w, void *w_split_ws,
                                         const float *bias, const float *residual, long long ldr, const float *gamma,
                                         const float *beta, float eps, long long rows, float *out, long long ldo,
                                         void *stream);

/* ---- the encoder layer's feed-forward block in one kernel (csrc/ffn_fused_split.hip) -------------------------------------
 * Replaces linear1 -> ReLU -> dropout2 -> linear2 -> `src + dropout3(src2)` -> norm2 (pixel_decoder/msdeformattn.py:122-131) in
 * eval mode (dropout is the identity):   out = LayerNorm(x + W2 . relu(W1 . x + b1) + b2)
 *   x, out [rows, 128] fp32 (row strides ldx, ldo floats); w1 [hidden, 128], b1 [hidden], w2 [128, hidden], b2 [128] or NULL,
 *   gamma / beta [128]; hidden % 32 == 0
 *   w_image_ws: device workspace of (hidden / 32) * 57 344 bytes, refilled on every call (one per stream)
 * fp32-accurate (exact three-way bf16 splits on the matrix cores, as the Linear entries); the [rows, hidden] activations stay
 * in registers between the two products.  Two launches on `stream`. */
PCT_API int pct_ffn_layernorm_f32(const float *x, long long ldx, const float *w1, const float *b1, const float *w2,
                                  const float *b2, const float *gamma, const float *beta, float eps, int hidden,
                                  long long rows, void *w_image_ws, float *out, long long ldo, void *stream);

/* ---- input projection of the pixel decoder in one entry: 1x1 convolution + GroupNorm(32, 128) + flatten into token rows ----
 * Replaces `nn.Sequential(Conv2d(in_channels, 128, kernel_size=1), nn.GroupNorm(32, 128))` followed by
 * `.flatten(2).transpose(1, 2)` and the concat over the levels (pixel_decoder/msdeformattn.py:213-226, 75-83):
 *     out[n][out_offset / 128 + p][c] = GroupNorm(conv(x))[n][c][p]
 *   x [batch, in_channels, hw] fp32, w [128, in_channels], bias [128] or NULL, gamma / beta [128]
 *   out: the [batch, S, 128] token buffer; out_batch_stride = S * 128 floats, out_offset = first row of this level * 128
 *   w_split_ws: 3 * 128 * in_channels * 2 bytes; partial_ws: batch * (hw / 128) * 2 * 64 floats; stats: batch * 64 floats
 * The convolution writes the token rows directly (its lanes hold 4 consecutive channels of a pixel) together with per-tile
 * (mean, centred sum of squares) records per group; the statistics are combined with Chan's formula and the rows normalised in
 * place.  Supported: out_channels == 128, groups == 32, in_channels % 16 == 0, hw % 128 == 0; otherwise PCT_ERR_UNSUPPORTED.
 * Four launches on `stream`. */
PCT_API int pct_conv1x1_groupnorm_tokens_f32(const float *x, const float *w, const float *bias, void *w_split_ws,
                                             const float *gamma, const float *beta, int groups, float eps, int batch,
                                             int in_channels, int out_channels, int hw, float *partial_ws, float *stats,
                                             float *out, long long out_batch_stride, long long out_offset, void *stream);

/* ---- 1x1 convolution on NCHW maps, fp32-accurate on the bf16 matrix cores (csrc/conv1x1_split.hip) -------------------
 * Replaces the `Conv2d(in_channels, conv_dim, kernel_size=1)` of the pixel decoder's input projections and FPN laterals
 * (pixel_decoder/msdeformattn.py:213-226, :262-277) for contiguous fp32 maps:
 *     out[n][co][p] = bias[co] + sum_ci w[co][ci] * x[n][ci][p]
 *   x [batch, in_channels, hw]  w [out_channels, in_channels]  bias [out_channels] or NULL  out [batch, out_channels, hw]
 *   w_split_ws: device workspace of 3 * out_channels * in_channels * 2 bytes, refilled on every call (one per stream)
 * Supported: out_channels == 128, in_channels % 16 == 0, hw % 128 == 0, 16-byte aligned tensors; otherwise
 * PCT_ERR_UNSUPPORTED / PCT_ERR_ALIGNMENT and the caller keeps the library convolution.  Two launches on `stream`. */
PCT_API int pct_conv1x1_nchw_f32(const float *x, const float *w, const float *bias, void *w_split_ws, int batch,
                                 int in_channels, int out_channels, int hw, float *out, void *stream);

/* ---- linear sum assignment on the device ------------------------------------------------------------------------
 * Replaces scipy.optimize.linear_sum_assignment(C.cpu()) of the matcher (connectomics/model/loss/matcher.py:154-165):
 * for every problem b, cost[b] is [num_query, ld_target] fp32 (row = query / prediction, column = target) of which the
 * first num_target[b] <= num_query columns are used; row_for_target[b, j] receives the query assigned to target j
 * (-1 for unused columns) such that the summed cost is minimal.  Same algorithm and fp64 arithmetic as scipy (Crouse's
 * shortest augmenting paths on the transposed problem).  status[b] = 1 when no finite assignment exists (NaN / +inf
 * costs).  num_query <= 1024, ld_target <= 512, else PCT_ERR_UNSUPPORTED.  All pointers are device pointers. */
PCT_API int pct_lsap_f32(const float *cost, int batch, int num_query, int ld_target, const int *num_target,
                         int *row_for_target, int *status, void *stream);

/* ---- fused masked attention core (MFMA, bf16 operands, fp32 accumulate) ---------------------------------------
 * Replaces q*scale -> bmm(q,k^T) -> masked_fill(-inf) -> softmax -> bmm(p,v) of multi_head_attention_forward
 * (transformer_decoder/attention.py:271-387) for the PCTrans decoder under bf16 autocast.
 *   q    [Q, N, heads*head_dim]   bfloat16      k  [S, N, heads*head_dim] bfloat16
 *   vT   [N, heads*16, S]         bfloat16 (v transposed: one row per value channel)
 *   mask [N, Q, S] bytes, nonzero = may not attend, shared by all heads; NULL = no mask
 *   out  [Q, N, heads*16]         out_dtype 0 = fp32, 2 = bfloat16
 * head_dim in {16, 32}, v_head_dim == 16, else PCT_ERR_UNSUPPORTED.  A fully masked row yields NaN (= softmax of
 * all -inf in the reference). */
PCT_API int pct_masked_attention_bf16(const void *q, const void *k, const void *vT, const unsigned char *mask,
                                      int batch, int heads, int num_query, int num_key, int head_dim,
                                      int v_head_dim, float scale, int out_dtype, void *out, void *stream);

/* ---- position-guided masked cross-attention core of the decoder (MFMA, bf16 operands, fp32 accumulate) ------------
 * The CrossAttentionLayer's own operand form (transformer_decoder/mask2former_transformer_decoder.py:130-183): per head
 * the query / key are [content (16) | position (16)] halves from different projections, taken here as separate tensors
 * in the projections' layout, so the per-head torch.cat passes (:160-172) and a transposed value tensor are not needed.
 * Same result, bit for bit, as pct_masked_attention_bf16 on the concatenated operands with head_dim 32.
 *   q_content, q_pos [Q, N, heads*16]   k_content, k_pos, v [S, N, heads*16]   bfloat16, 16-byte aligned
 *   mask [N, Q, S] bytes, nonzero = may not attend, shared by all heads; NULL = no mask
 *   row_open [N, Q] bytes or NULL: nonzero = ignore this query's mask row (the decoder's "a query whose mask rules out every
 *        pixel attends everywhere instead", :561, without rewriting the mask tensor)
 *   out  [Q, N, heads*16] bfloat16
 * Geometry: heads % 4 == 0, num_key % 64 == 0 (mask: 16-byte aligned); anything else returns PCT_ERR_UNSUPPORTED and the
 * caller concatenates and uses pct_masked_attention_bf16.  A fully masked row yields NaN, as there. */
PCT_API int pct_cross_attention_bf16(const void *q_content, const void *q_pos, const void *k_content, const void *k_pos,
                                     const void *v, const unsigned char *mask, const unsigned char *row_open, int batch,
                                     int heads, int num_query, int num_key, float scale, void *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PCTRANS_HIP_H_ */
