"""Drop-in for the reference's compiled extension module `MultiScaleDeformableAttention`.

Replaces (reference checkout, OPS = connectomics/model/maskformer_block/pixel_decoder/ops):
    OPS/src/vision.cpp:18-21                 the pybind module and its two functions
    OPS/src/ms_deform_attn.h:25-67           device dispatch ("Not implemented on the CPU")
    OPS/src/cuda/ms_deform_attn_cuda.cu:25-158   contiguity / device asserts, output allocation

Same positional signatures, same error behaviour: non-contiguous or non-device tensors raise, CPU tensors raise
"Not implemented on the CPU" (callers such as OPS/modules/ms_deform_attn.py:116-121 rely on that exception),
`batch % min(batch, im2col_step) != 0` raises.  The work itself is one launch of the hand-written gfx950 kernel
through the C ABI of libpctrans_hip.so on the caller's current stream; nothing is synchronised.

To let reference code `import MultiScaleDeformableAttention` resolve to this module, see `install_as_extension()`.
"""
import sys

import torch

from . import _lib

_FWD = {
    torch.float32: "pct_ms_deform_attn_forward_f32",
    torch.float64: "pct_ms_deform_attn_forward_f64",
    torch.float16: "pct_ms_deform_attn_forward_f16",
    torch.bfloat16: "pct_ms_deform_attn_forward_bf16",
}
_BWD = {
    torch.float32: "pct_ms_deform_attn_backward_f32",
    torch.float64: "pct_ms_deform_attn_backward_f64",
}


def _check_inputs(named):
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)          # cu:33-37
    if not named[0][1].is_cuda:
        raise RuntimeError("Not implemented on the CPU")                        # ms_deform_attn.h:43
    dev = named[0][1].device
    for name, t in named:
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % name)               # cu:39-43 (device tensor on ROCm)
        if t.device != dev:
            raise RuntimeError("%s is on %s, value is on %s" % (name, t.device, dev))


def _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5 or spatial_shapes.dim() != 2:
        raise RuntimeError("ms_deform_attn: expected value[N,S,M,D], sampling_loc[N,Lq,M,L,P,2], "
                           "attn_weight[N,Lq,M,L,P], spatial_shapes[L,2]")
    N, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    if tuple(sampling_loc.shape) != (N, Lq, M, L, P, 2) or tuple(attn_weight.shape) != (N, Lq, M, L, P) \
            or tuple(spatial_shapes.shape) != (L, 2) or level_start_index.numel() != L:
        raise RuntimeError("ms_deform_attn: inconsistent shapes value=%s loc=%s attn=%s shapes=%s starts=%s" % (
            tuple(value.shape), tuple(sampling_loc.shape), tuple(attn_weight.shape),
            tuple(spatial_shapes.shape), tuple(level_start_index.shape)))
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64")   # kernels read int64_t (cuh:245-246)
    return N, S, M, D, L, Lq, P


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


from ._timing import kernel_timing, timed as _timed  # noqa: E402,F401  (bench.py's roofline leg)

KERNEL_NAMES = {0: "none", 1: "windowed (msda_forward_win.hip)", 2: "generic (msda_forward.hip)",
                3: "quad-owner (msda_forward_dpp.hip)", 4: "pyramid-column (msda_forward_col.hip / msda_forward_col16.hip)"}


def _last_kernel():
    """Kernel id of the forward launch enqueued last BY THE PROCESS (pct_msda_last_kernel is one process-wide atomic word:
    right for the one-thread-per-process launch the path uses, DDP; under threaded launches -- nn.DataParallel -- it may be
    another thread's)."""
    return int(_lib.lib().pct_msda_last_kernel())


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """-> Tensor[N, Lq, M*D].  fp32/fp64 as the reference; fp16/bf16 value with fp32 loc/weights is new capability
    (loc / weights given in the 16-bit dtype are promoted to fp32 first)."""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                   ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)])
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    if value.dtype not in _FWD:
        raise RuntimeError("ms_deform_attn_forward: unsupported dtype %s" % value.dtype)
    aux = value.dtype if value.dtype in (torch.float32, torch.float64) else torch.float32
    if sampling_loc.dtype != aux:
        sampling_loc = sampling_loc.to(aux)
    if attn_weight.dtype != aux:
        attn_weight = attn_weight.to(aux)
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    _lib.prepare_device(value.device)
    with torch.cuda.device(value.device), _timed("forward", value, _last_kernel):
        rc = getattr(_lib.lib(), _FWD[value.dtype])(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
            attn_weight.data_ptr(), N, S, M, D, L, Lq, P, int(im2col_step), out.data_ptr(), _stream(value))
    _lib.check(rc, "ms_deform_attn_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight] (fp32 / fp64, as the reference dispatches)."""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                   ("sampling_loc", sampling_loc), ("attn_weight", attn_weight), ("grad_output", grad_output)])
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    if value.dtype not in _BWD:
        raise RuntimeError("ms_deform_attn_backward: unsupported dtype %s" % value.dtype)
    for name, t in (("sampling_loc", sampling_loc), ("attn_weight", attn_weight), ("grad_output", grad_output)):
        if t.dtype != value.dtype:
            raise RuntimeError("%s dtype %s does not match value dtype %s" % (name, t.dtype, value.dtype))
    if grad_output.numel() != N * Lq * M * D:
        raise RuntimeError("grad_output has %d elements, expected %d" % (grad_output.numel(), N * Lq * M * D))
    grad_value = torch.empty_like(value)
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    _lib.prepare_device(value.device)
    with torch.cuda.device(value.device), _timed("backward", value):
        rc = getattr(_lib.lib(), _BWD[value.dtype])(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
            attn_weight.data_ptr(), grad_output.data_ptr(), N, S, M, D, L, Lq, P, int(im2col_step),
            grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(), _stream(value))
    _lib.check(rc, "ms_deform_attn_backward")
    return [grad_value, grad_loc, grad_attn]


def fused_forward_supported(value, reference_points, sampling_offsets):
    """Geometry the fused front-end kernel covers (PCTrans: fp32, 16 channels per head, 4 or 8 points, 2-d refs)."""
    return (value.is_cuda and value.dtype == torch.float32 and value.dim() == 4 and value.shape[3] == 16
            and sampling_offsets.shape[4] in (4, 8) and reference_points.shape[-1] == 2
            and value.shape[1] * value.shape[2] * value.shape[3] * 4 < 2 ** 31 - 1)      # per image; larger batches are chunked inside the library


def ms_deform_attn_fused_forward(value, spatial_shapes, level_start_index, reference_points, sampling_offsets,
                                 attention_logits):
    """MSDeformAttn.forward's middle section in one launch (OPS/modules/ms_deform_attn.py:100-118):
    softmax(attention_logits over L*P), reference_points + sampling_offsets / (W_l, H_l), then the sampling op.

    value [N,S,M,16] fp32; reference_points [N or 1, Lq, L, 2] (an expanded, stride-0 batch dim is accepted);
    sampling_offsets [N,Lq,M,L,P,2]; attention_logits [N,Lq,M,L*P] (raw Linear outputs) -> [N, Lq, M*16].
    Forward only (no autograd): the module uses it when no gradient is required."""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                   ("sampling_offsets", sampling_offsets), ("attention_logits", attention_logits)])
    N, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq, P = sampling_offsets.shape[1], sampling_offsets.shape[4]
    if tuple(sampling_offsets.shape) != (N, Lq, M, L, P, 2) or attention_logits.numel() != N * Lq * M * L * P:
        raise RuntimeError("ms_deform_attn_fused_forward: inconsistent shapes")
    if not fused_forward_supported(value, reference_points, sampling_offsets):
        raise RuntimeError("ms_deform_attn_fused_forward: unsupported geometry (use the unfused op)")
    if sampling_offsets.dtype != torch.float32 or attention_logits.dtype != torch.float32:
        raise RuntimeError("ms_deform_attn_fused_forward: offsets / logits must be float32")
    ref = reference_points
    if ref.dtype != torch.float32 or not ref.is_cuda:
        raise RuntimeError("reference_points must be a float32 device tensor")
    if tuple(ref.shape[1:]) != (Lq, L, 2) or ref.shape[0] not in (1, N):
        raise RuntimeError("reference_points must be [N or 1, Lq, L, 2]")
    if ref.stride()[1:] != (L * 2, 2, 1):
        ref = ref.contiguous()
    batch_stride = 0 if (ref.shape[0] == 1 or ref.stride(0) == 0) else ref.stride(0)
    if batch_stride not in (0, Lq * L * 2):
        ref = ref.contiguous()
        batch_stride = Lq * L * 2
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    _lib.prepare_device(value.device)
    with torch.cuda.device(value.device), _timed("forward", value, _last_kernel):
        rc = _lib.lib().pct_ms_deform_attn_fused_forward_f32(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), ref.data_ptr(), batch_stride,
            sampling_offsets.data_ptr(), attention_logits.data_ptr(), N, S, M, D, L, Lq, P, out.data_ptr(),
            _stream(value))
    _lib.check(rc, "ms_deform_attn_fused_forward")
    return out


def to_planes(t, num_heads):
    """[N, S, num_heads * C] (or [N, S, num_heads, ...] with C trailing elements per head) -> the piece-plane layout
    [N, num_heads * C / 4, S, 4] of pct_ms_deform_attn_forward_planes_f32 (a copy; the encoder layer's projection GEMM writes
    this layout directly, this helper is for tests and tools)."""
    N, S = t.shape[0], t.shape[1]
    x = t.reshape(N, S, num_heads, -1)
    C = x.shape[3]
    if C % 4:
        raise RuntimeError("to_planes: elements per head must be a multiple of 4")
    return x.reshape(N, S, num_heads * (C // 4), 4).permute(0, 2, 1, 3).contiguous()


def from_planes(p, num_heads):
    """Inverse of to_planes: [N, num_heads * C / 4, S, 4] -> [N, S, num_heads, C]."""
    N, NP, S, _ = p.shape
    return p.permute(0, 2, 1, 3).reshape(N, S, num_heads, (NP // num_heads) * 4).contiguous()


def planes_forward_supported(N, S, M, D, L, Lq, P, dtype=torch.float32):
    """Geometry pct_ms_deform_attn_forward_planes_f32 covers (the pyramid-column kernel's)."""
    return (dtype == torch.float32 and D == 16 and P == 4 and 3 <= L <= 5 and Lq == S and S < (1 << 24)
            and S * M * L * P * 8 < 2 ** 31 - 1 and N * (S + 4096) * M < 2 ** 31 - 1)


def ms_deform_attn_forward_planes(value_planes, spatial_shapes, level_start_index, loc_planes, attn_planes, num_heads,
                                  reference_points=None):
    """The sampling op on PIECE-PLANE operands (include/pctrans_hip.h): value_planes [N, M*4, S, 4], loc_planes
    [N, M*L*P/2, S, 4], attn_planes [N, M*L*P/4, S, 4] fp32 -> [N, S, M*16], bit-identical to ms_deform_attn_forward /
    ms_deform_attn_fused_forward on the same numbers in the reference layout.  `reference_points` None: locations and
    weights; [N or 1, S, L, 2]: raw offsets and logits (the fused front-end).  Forward only."""
    _check_inputs([("value_planes", value_planes), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("loc_planes", loc_planes), ("attn_planes", attn_planes)])
    N, VP, S, four = value_planes.shape
    M, L = int(num_heads), spatial_shapes.shape[0]
    if four != 4 or VP % M or loc_planes.dim() != 4 or attn_planes.dim() != 4:
        raise RuntimeError("ms_deform_attn_forward_planes: operands must be [N, planes, S, 4]")
    D = VP // M * 4
    P = attn_planes.shape[1] * 4 // (M * L)
    if (tuple(loc_planes.shape) != (N, M * L * P // 2, S, 4) or tuple(attn_planes.shape) != (N, M * L * P // 4, S, 4)
            or P * M * L != attn_planes.shape[1] * 4):
        raise RuntimeError("ms_deform_attn_forward_planes: inconsistent shapes")
    if any(t.dtype != torch.float32 for t in (value_planes, loc_planes, attn_planes)):
        raise RuntimeError("ms_deform_attn_forward_planes: float32 operands only")
    if not planes_forward_supported(N, S, M, D, L, S, P):
        raise RuntimeError("ms_deform_attn_forward_planes: unsupported geometry (use the reference layout)")
    ref, batch_stride = None, 0
    if reference_points is not None:
        ref = reference_points
        if ref.dtype != torch.float32 or not ref.is_cuda:
            raise RuntimeError("reference_points must be a float32 device tensor")
        if tuple(ref.shape[1:]) != (S, L, 2) or ref.shape[0] not in (1, N):
            raise RuntimeError("reference_points must be [N or 1, S, L, 2]")
        if ref.stride()[1:] != (L * 2, 2, 1):
            ref = ref.contiguous()
        batch_stride = 0 if (ref.shape[0] == 1 or ref.stride(0) == 0) else ref.stride(0)
        if batch_stride not in (0, S * L * 2):
            ref = ref.contiguous()
            batch_stride = S * L * 2
    out = torch.empty((N, S, M * D), dtype=torch.float32, device=value_planes.device)
    _lib.prepare_device(value_planes.device)
    with torch.cuda.device(value_planes.device), _timed("forward", value_planes, _last_kernel):
        rc = _lib.lib().pct_ms_deform_attn_forward_planes_f32(
            value_planes.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), loc_planes.data_ptr(),
            attn_planes.data_ptr(), ref.data_ptr() if ref is not None else None, batch_stride, N, S, M, D, L, S, P,
            out.data_ptr(), _stream(value_planes))
    _lib.check(rc, "ms_deform_attn_forward_planes")
    return out


def install_as_extension():
    """Register this module under the top-level name the reference imports
    (`import MultiScaleDeformableAttention as MSDA`, OPS/functions/ms_deform_attn_func.py:21-22)."""
    sys.modules["MultiScaleDeformableAttention"] = sys.modules[__name__]
    return sys.modules[__name__]
