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


def masked_attention_supported(q, k, v, num_heads, attn_mask, key_padding_mask, dropout_p, training, need_weights):
    """bf16 device tensors, head dims (32|16, 16), boolean mask shared by the heads (or none), forward only."""
    if not (q.is_cuda and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16):
        return False
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        return False
    if need_weights or key_padding_mask is not None or (dropout_p > 0.0 and training):
        return False
    if q.shape[2] // num_heads not in (16, 32) or v.shape[2] // num_heads != 16:
        return False
    if attn_mask is not None:
        if attn_mask.dtype != torch.bool:
            return False
        L, N, S = q.shape[0], q.shape[1], k.shape[0]
        if tuple(attn_mask.shape) not in ((N, 1, L, S), (L, S)):
            return False
    return True


def masked_attention(q, k, v, num_heads, attn_mask=None, v_t=None):
    """softmax(mask(q k^T / sqrt(head_dim))) v per head: q [L,N,E], k [S,N,E], v [S,N,Ev] bf16 -> [L,N,Ev] bf16.
    attn_mask: None | bool [N,1,L,S] | bool [L,S], True = may not attend.  `v_t` [N, Ev, S] may be given instead of v."""
    L, N, E = q.shape
    S = k.shape[0]
    hd = E // num_heads
    qc, kc = q.contiguous(), k.contiguous()
    vT = v_t.contiguous() if v_t is not None else v.permute(1, 2, 0).contiguous()   # [N, Ev, S]
    Ev = vT.shape[1]
    m = None
    if attn_mask is not None:
        m = attn_mask.expand(N, L, S) if attn_mask.dim() == 2 else attn_mask.reshape(N, L, S)
        m = m.contiguous()
    out = torch.empty((L, N, Ev), dtype=torch.bfloat16, device=q.device)
    with torch.cuda.device(q.device):
        rc = _lib.lib().pct_masked_attention_bf16(
            qc.data_ptr(), kc.data_ptr(), vT.data_ptr(), m.data_ptr() if m is not None else None, N, num_heads, L, S,
            hd, Ev // num_heads, float(hd) ** -0.5, 2, out.data_ptr(), torch.cuda.current_stream(q.device).cuda_stream)
    _lib.check(rc, "masked_attention")
    return out
