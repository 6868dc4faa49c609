"""Python entry points of the small fused HIP kernels (C ABI in include/pctrans_hip.h).  Each function has the same
result as the torch expression in its docstring; callers use them only when no gradient is required (forward-only
kernels) and the tensors are fp32 on the device, otherwise they evaluate the torch expression."""
import torch
from torch.nn import functional as F

from . import _lib


def _fusable(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        return False
    return all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in tensors)


def add_layer_norm(x, y, norm):
    """norm(x + y) for an nn.LayerNorm over the last dim (y may be None -> norm(x))."""
    cols = x.shape[-1]
    if (not _fusable(x, y, norm.weight, norm.bias) or cols not in (64, 128, 256) or norm.weight is None
            or norm.bias is None or tuple(norm.normalized_shape) != (cols,) or (y is not None and y.shape != x.shape)):
        return norm(x if y is None else x + y)
    xc = x.contiguous()
    yc = y.contiguous() if y is not None else None
    out = torch.empty_like(xc)
    rows = xc.numel() // cols
    with torch.cuda.device(x.device):
        rc = _lib.lib().pct_add_layernorm_f32(
            xc.data_ptr(), yc.data_ptr() if yc is not None else None, norm.weight.data_ptr(), norm.bias.data_ptr(),
            float(norm.eps), rows, cols, out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "add_layer_norm")
    return out


def linear_relu(x, linear):
    """relu(linear(x)) with the bias + ReLU applied in the hipBLASLt GEMM epilogue (one pass over the activations)."""
    if not _fusable(x, linear.weight, linear.bias) or linear.bias is None:
        return F.relu(linear(x))
    x2 = x.reshape(-1, x.shape[-1])
    out = torch._addmm_activation(linear.bias, x2, linear.weight.t(), use_gelu=False)
    return out.view(*x.shape[:-1], linear.out_features)
