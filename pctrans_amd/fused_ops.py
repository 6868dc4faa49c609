"""Python entry points of the small fused HIP kernels (C ABI in include/pctrans_hip.h).  Each function has the same
result as the torch expression in its docstring; callers use them only when no gradient is required (forward-only
kernels) and the tensors are fp32 on the device, otherwise they evaluate the torch expression."""
import contextlib

import torch
from torch.nn import functional as F

from . import _lib, _timing


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


def linear_k128_supported(x, weight, bias=None):
    """fp32 device tensors, in_features == 128, out_features % 32 == 0, forward only, outside autocast."""
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.is_cuda):
        return False
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return False
    if torch.is_autocast_enabled():
        return False
    return (x.shape[-1] == 128 and weight.shape[1] == 128 and weight.shape[0] % 32 == 0 and x.numel() >= 128 * 4096
            and weight.is_contiguous() and (bias is None or bias.is_contiguous()))


def _rows_2d(x):
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    return x2


def _x_add_ok(x, x_add):
    """x_add: same trailing shape as x, fp32, and either the same batch or batch 1 (broadcast over the images)."""
    return (x_add.is_cuda and x_add.dtype == torch.float32 and x_add.dim() == x.dim() and x.dim() >= 2
            and x_add.shape[1:] == x.shape[1:] and x_add.shape[0] in (1, x.shape[0])
            and x.numel() // x.shape[0] // 128 >= 32)


def linear_k128(x, weight, bias=None, relu=False, x_add=None):
    """act((x + x_add) @ weight.T + bias) for in_features = 128 on the hand-written fp32 MFMA kernel
    (csrc/linear_k128.hip); x_add (shape of x, or batch 1 = shared by the images, or None) is summed on the way into
    the kernel's LDS tiles."""
    x2 = _rows_2d(x)
    rows, n = x2.shape[0], weight.shape[0]
    a2, period = None, 0
    if x_add is not None:
        if x_add.shape[0] != 1 and x_add.stride(0) == 0:          # an expanded batch-1 tensor
            x_add = x_add[:1]
        a2 = _rows_2d(x_add)
        period = a2.shape[0]
    out = torch.empty((rows, n), dtype=torch.float32, device=x.device)
    timed = _timing.timed("linear_k128 n=1024 relu", x2) if (n == 1024 and relu) else contextlib.nullcontext()
    with torch.cuda.device(x.device), timed:
        rc = _lib.lib().pct_linear_k128_f32(
            x2.data_ptr(), x2.stride(0), a2.data_ptr() if a2 is not None else None,
            a2.stride(0) if a2 is not None else 0, period, weight.data_ptr(),
            bias.data_ptr() if bias is not None else None, rows, n,
            1 if relu else 0, out.data_ptr(), n, torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "linear_k128")
    return out.view(*x.shape[:-1], n)


def conv1x1_from_token_rows_supported(x, conv):
    """A 1x1 convolution of 128 channels-last feature maps with at most 32 output channels, forward only, small batch:
    the decoder's `mask_head` (dec.py:539) on the pixel decoder's finest map, which is a transposed VIEW of the encoder's
    [N, S, 128] token rows."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 128 and x.shape[0] <= 4):
        return False
    if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
        return False
    H, W = x.shape[2], x.shape[3]
    return (x.stride(1) == 1 and x.stride(3) == 128 and x.stride(2) == 128 * W and H * W >= 4096
            and x.data_ptr() % 16 == 0 and (x.stride(0) * 4) % 16 == 0
            and conv.kernel_size == (1, 1) and conv.in_channels == 128 and conv.out_channels <= 32
            and conv.weight.dtype == torch.float32 and getattr(conv, "norm", None) is None
            and getattr(conv, "activation", None) is None)


def conv1x1_from_token_rows(x, conv):
    """-> conv(x) as fp32 [N, Cout, H, W], each image's H*W token rows through the K = 128 MFMA kernel (weights padded to 32
    output channels).  Deterministic run to run, unlike the library's batched bf16 GEMM this replaces at batch 1-4
    (tools/diag_determinism.py), which is what lets a HIP-graph replay of the head reproduce the eager forward bit for
    bit; fp32-accurate where autocast would have rounded the operands to bf16."""
    N, _, H, W = x.shape
    co = conv.out_channels
    # the padded copies live ON the module (a plain attribute, not a buffer: the state dict is unchanged), so they can
    # never be taken for another module's -- a process-wide table keyed on id(conv) could, once a freed module's id is
    # handed to a new one -- and they are rebuilt whenever a parameter was written in place (version counter), replaced
    # or moved (storage pointer, device)
    w, b = conv.weight, conv.bias
    stamp = (w._version, w.data_ptr(), b._version if b is not None else -1, b.data_ptr() if b is not None else 0,
             x.device)
    ent = conv.__dict__.get("_pct_pad32")
    if ent is None or ent[0] != stamp:
        w32 = torch.zeros((32, 128), dtype=torch.float32, device=x.device)
        w32[:co] = w.detach().reshape(co, 128)
        b32 = torch.zeros((32,), dtype=torch.float32, device=x.device)
        if b is not None:
            b32[:co] = b.detach()
        ent = (stamp, w32, b32)
        conv.__dict__["_pct_pad32"] = ent
    outs = [linear_k128(x[n].permute(1, 2, 0).reshape(H * W, 128), ent[1], ent[2]) for n in range(N)]   # views: no copy
    y = torch.stack(outs, 0)                                               # [N, HW, 32]
    return y[..., :co].permute(0, 2, 1).reshape(N, co, H, W).contiguous()


def linear_k128_multi(x, layers, x_add=None):
    """[lin(x + x_add if use_add else x) for (lin, use_add) in layers] in ONE launch of the K = 128 kernel: the rows are
    read from HBM once for all layers (MSDeformAttn: value_proj(src), sampling_offsets(src + pos),
    attention_weights(src + pos), ops/modules/ms_deform_attn.py:96-103).  Every layer must satisfy
    linear_k128_supported; 1 <= len(layers) <= 4."""
    import ctypes
    x2 = _rows_2d(x)
    rows = x2.shape[0]
    a2, period = None, 0
    if x_add is not None:
        if x_add.shape[0] != 1 and x_add.stride(0) == 0:
            x_add = x_add[:1]
        a2 = _rows_2d(x_add)
        period = a2.shape[0]
    k = len(layers)
    outs = [torch.empty((rows, lin.out_features), dtype=torch.float32, device=x.device) for lin, _ in layers]
    vp, ll, ci = ctypes.c_void_p * k, ctypes.c_longlong * k, ctypes.c_int * k
    w = vp(*[lin.weight.data_ptr() for lin, _ in layers])
    b = vp(*[(lin.bias.data_ptr() if lin.bias is not None else None) for lin, _ in layers])
    n = ci(*[lin.out_features for lin, _ in layers])
    add = ci(*[1 if use else 0 for _, use in layers])
    y = vp(*[o.data_ptr() for o in outs])
    ldy = ll(*[o.shape[1] for o in outs])
    with torch.cuda.device(x.device):
        rc = _lib.lib().pct_linear_k128_multi_f32(
            x2.data_ptr(), x2.stride(0), a2.data_ptr() if a2 is not None else None,
            a2.stride(0) if a2 is not None else 0, period, k, w, b, n, add, y, ldy, rows,
            torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "linear_k128_multi")
    return [o.view(*x.shape[:-1], o.shape[1]) for o in outs]


def linear_k128_multi_supported(x, lins, x_add=None):
    return (1 <= len(lins) <= 4 and all(linear_k128_supported(x, l.weight, l.bias) for l in lins)
            and (x_add is None or _x_add_ok(x, x_add))
            and all(l.weight.data_ptr() % 16 == 0 and (l.bias is None or l.bias.data_ptr() % 16 == 0) for l in lins))


def linear(x, lin, relu=False, x_add=None):
    """[relu](lin(x [+ x_add])) for an nn.Linear: the K = 128 MFMA kernel when it applies, else the library GEMM (with
    the bias + ReLU in its epilogue)."""
    if linear_k128_supported(x, lin.weight, lin.bias) and (
            x_add is None or _x_add_ok(x, x_add)):
        return linear_k128(x, lin.weight, lin.bias, relu=relu, x_add=x_add)
    if x_add is not None:
        x = x + x_add
    return linear_relu(x, lin) if relu else lin(x)


def linear_add_layer_norm(x, linear, residual, norm):
    """norm(residual + linear(x)) for a 128 -> 128 Linear and LayerNorm(128): one kernel, one pass over the rows."""
    if not (linear_k128_supported(x, linear.weight, linear.bias) and linear.out_features == 128
            and residual.dtype == torch.float32 and residual.shape[:-1] == x.shape[:-1] and residual.shape[-1] == 128
            and norm.weight is not None and norm.bias is not None and tuple(norm.normalized_shape) == (128,)):
        return add_layer_norm(residual, linear(x), norm)
    x2, r2 = _rows_2d(x), _rows_2d(residual)
    rows = x2.shape[0]
    out = torch.empty((rows, 128), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().pct_linear_k128_add_layernorm_f32(
            x2.data_ptr(), x2.stride(0), linear.weight.data_ptr(),
            linear.bias.data_ptr() if linear.bias is not None else None, r2.data_ptr(), r2.stride(0),
            norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps), rows, out.data_ptr(), 128,
            torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "linear_add_layer_norm")
    return out.view(residual.shape)


_SPLIT_WS = {}                # (device, stream, K) -> workspace the kernel refills with the split weights on every call:
                              # one per stream, so launches on two streams never share it


def linear_layer_norm_supported(x, linear, residual, norm):
    """fp32 device tensors, Linear(K -> 128) with K % 32 == 0 followed by residual add and LayerNorm(128), forward only."""
    w = linear.weight
    return (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and w.device == x.device
            and w.is_contiguous() and linear.out_features == 128 and linear.in_features % 32 == 0
            and x.shape[-1] == linear.in_features and x.numel() > 0 and x.stride(-1) == 1
            and residual.dtype == torch.float32 and residual.device == x.device and residual.shape[-1] == 128
            and residual.shape[:-1] == x.shape[:-1] and residual.stride(-1) == 1
            and isinstance(norm, torch.nn.LayerNorm) and tuple(norm.normalized_shape) == (128,)
            and norm.weight is not None and norm.bias is not None and not torch.is_autocast_enabled()
            and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or residual.requires_grad)))


def linear_layer_norm(x, linear, residual, norm):
    """norm(residual + linear(x)) for Linear(K -> 128) + LayerNorm(128): the encoder FFN's linear2 / dropout3 / norm2
    (pixel_decoder/msdeformattn.py:122-131) as one kernel; the [rows, 128] product never goes to memory."""
    if not linear_layer_norm_supported(x, linear, residual, norm):
        return add_layer_norm(residual, linear(x), norm)
    x2, r2 = _rows_2d(x), _rows_2d(residual)
    rows, k = x2.shape
    aligned = all(t.data_ptr() % 16 == 0 for t in (x2, r2, linear.weight, norm.weight, norm.bias)) and \
        (linear.bias is None or linear.bias.data_ptr() % 16 == 0) and x2.stride(0) % 4 == 0 and r2.stride(0) % 4 == 0
    if not aligned:
        return add_layer_norm(residual, linear(x), norm)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    key = (x.device, stream, k)
    ws = _SPLIT_WS.get(key)
    if ws is None:
        ws = _SPLIT_WS[key] = torch.empty((3, 128, k), dtype=torch.bfloat16, device=x.device)
    out = torch.empty((rows, 128), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timing.timed("linear_ln k=%d" % k, x2):
        rc = _lib.lib().pct_linear_add_layernorm_f32(
            x2.data_ptr(), x2.stride(0), k, linear.weight.data_ptr(), ws.data_ptr(),
            linear.bias.data_ptr() if linear.bias is not None else None, r2.data_ptr(), r2.stride(0),
            norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps), rows, out.data_ptr(), 128, stream)
    _lib.check(rc, "linear_layer_norm")
    return out.view(residual.shape)


_FFN_WS = {}                  # (device, stream, hidden) -> weight-image workspace of ffn_layer_norm (refilled per call)


def ffn_layer_norm_supported(x, linear1, linear2, norm):
    """fp32 device rows of width 128 through Linear(128 -> F) + ReLU + Linear(F -> 128) + residual + LayerNorm(128), F % 32 == 0,
    forward only (csrc/ffn_fused_split.hip)."""
    w1, w2 = linear1.weight, linear2.weight
    return (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 128 and x.stride(-1) == 1 and x.numel() > 0
            and w1.dtype == torch.float32 and w2.dtype == torch.float32 and w1.device == x.device and w2.device == x.device
            and w1.is_contiguous() and w2.is_contiguous() and linear1.in_features == 128 and linear2.out_features == 128
            and linear1.out_features == linear2.in_features and linear1.out_features % 32 == 0
            and linear1.bias is not None and isinstance(norm, torch.nn.LayerNorm) and tuple(norm.normalized_shape) == (128,)
            and norm.weight is not None and norm.bias is not None and not torch.is_autocast_enabled()
            and not (torch.is_grad_enabled() and (x.requires_grad or w1.requires_grad or w2.requires_grad)))


def ffn_layer_norm(x, linear1, linear2, norm):
    """norm(x + linear2(relu(linear1(x)))): the encoder FFN in one kernel; the [rows, F] hidden tensor never goes to memory."""
    x2 = _rows_2d(x)
    rows = x2.shape[0]
    f = linear1.out_features
    ptrs = [x2, linear1.weight, linear1.bias, linear2.weight, norm.weight, norm.bias] + ([linear2.bias] if linear2.bias is not None else [])
    if not (all(t.data_ptr() % 16 == 0 for t in ptrs) and x2.stride(0) % 4 == 0):
        return linear_layer_norm(linear(x, linear1, relu=True), linear2, x, norm)
    _lib.prepare_device(x.device)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    key = (x.device, stream, f)
    ws = _FFN_WS.get(key)
    if ws is None:
        ws = _FFN_WS[key] = torch.empty(((f // 32) * 57344,), dtype=torch.uint8, device=x.device)
    out = torch.empty((rows, 128), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timing.timed("ffn f=%d" % f, x2):
        rc = _lib.lib().pct_ffn_layernorm_f32(
            x2.data_ptr(), x2.stride(0), linear1.weight.data_ptr(), linear1.bias.data_ptr(), linear2.weight.data_ptr(),
            linear2.bias.data_ptr() if linear2.bias is not None else None, norm.weight.data_ptr(), norm.bias.data_ptr(),
            float(norm.eps), f, rows, ws.data_ptr(), out.data_ptr(), 128, stream)
    _lib.check(rc, "ffn_layer_norm")
    return out.view(x.shape)


_CONV_WS = {}                 # (device, stream, in_channels) -> split-weight workspace of conv1x1_nchw (refilled per call)


def conv1x1_nchw_supported(x, conv):
    """fp32 contiguous NCHW device map through a plain 1x1 Conv2d with 128 output channels, forward only
    (csrc/conv1x1_split.hip: in_channels % 16 == 0, H * W % 128 == 0)."""
    w = conv.weight
    return (isinstance(conv, torch.nn.Conv2d) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and w.dtype == torch.float32 and w.device == x.device and tuple(w.shape[2:]) == (1, 1) and w.shape[0] == 128
            and w.shape[1] == x.shape[1] and w.shape[1] % 16 == 0 and (x.shape[2] * x.shape[3]) % 128 == 0
            and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (0, 0) and tuple(conv.dilation) == (1, 1)
            and conv.groups == 1 and getattr(conv, "norm", None) is None and getattr(conv, "activation", None) is None
            and x.numel() > 0 and x.shape[0] * (x.shape[2] * x.shape[3] // 128) < 2 ** 31 - 1
            and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and w.is_contiguous()
            and (conv.bias is None or conv.bias.data_ptr() % 16 == 0) and not torch.is_autocast_enabled()
            and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)))


def conv1x1_nchw(x, conv):
    """conv(x) for a 1x1 Conv2d(in -> 128) on a contiguous fp32 NCHW map: exact split-bf16 products on the matrix cores
    (as accurate as the fp32 convolution it replaces).  Falls back to the module where the kernel does not apply."""
    if not conv1x1_nchw_supported(x, conv):
        return conv(x)
    n, k, h, w_ = x.shape
    stream = torch.cuda.current_stream(x.device).cuda_stream
    key = (x.device, stream, k)
    ws = _CONV_WS.get(key)
    if ws is None:
        ws = _CONV_WS[key] = torch.empty((3, 128, k), dtype=torch.bfloat16, device=x.device)
    out = torch.empty((n, 128, h, w_), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timing.timed("conv1x1 k=%d" % k, x):
        rc = _lib.lib().pct_conv1x1_nchw_f32(
            x.data_ptr(), conv.weight.data_ptr(), conv.bias.data_ptr() if conv.bias is not None else None, ws.data_ptr(),
            n, k, 128, h * w_, out.data_ptr(), stream)
    _lib.check(rc, "conv1x1_nchw")
    return out


def conv1x1_groupnorm_tokens_supported(x, conv, gn):
    """conv1x1_nchw_supported + an affine GroupNorm(32, 128) behind the convolution."""
    return (conv1x1_nchw_supported(x, conv) and isinstance(gn, torch.nn.GroupNorm) and gn.num_groups == 32
            and gn.num_channels == 128 and gn.weight is not None and gn.bias is not None
            and gn.weight.data_ptr() % 16 == 0 and gn.bias.data_ptr() % 16 == 0 and x.shape[0] <= 65535
            and not (torch.is_grad_enabled() and gn.weight.requires_grad))


def conv1x1_groupnorm_tokens_into(x, conv, gn, out, row_offset):
    """out[:, row_offset : row_offset + H*W, :] = gn(conv(x)).flatten(2).transpose(1, 2)  (out: [N, S, 128] contiguous fp32):
    the pixel decoder's input projection of one level in one entry (csrc/conv1x1_split.hip)."""
    n, k, h, w_ = x.shape
    hw = h * w_
    stream = torch.cuda.current_stream(x.device).cuda_stream
    key = (x.device, stream, k)
    ws = _CONV_WS.get(key)
    if ws is None:
        ws = _CONV_WS[key] = torch.empty((3, 128, k), dtype=torch.bfloat16, device=x.device)
    partial = torch.empty((n * (hw // 128) * 2 * 64,), dtype=torch.float32, device=x.device)
    stats = torch.empty((n * 64,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timing.timed("conv1x1_gn k=%d" % k, x):
        rc = _lib.lib().pct_conv1x1_groupnorm_tokens_f32(
            x.data_ptr(), conv.weight.data_ptr(), conv.bias.data_ptr() if conv.bias is not None else None, ws.data_ptr(),
            gn.weight.data_ptr(), gn.bias.data_ptr(), 32, float(gn.eps), n, k, 128, hw, partial.data_ptr(), stats.data_ptr(),
            out.data_ptr(), out.stride(0), row_offset * 128, stream)
    _lib.check(rc, "conv1x1_groupnorm_tokens")


def groupnorm_flatten_supported(x, gn):
    """fp32 NCHW device tensor, 128 channels, groups of a multiple of 4 channels, affine, forward only."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 128 and x.is_contiguous()
            and isinstance(gn, torch.nn.GroupNorm) and gn.num_channels == 128 and (128 // gn.num_groups) % 4 == 0
            and gn.weight is not None and gn.bias is not None and not torch.is_autocast_enabled()
            and not (torch.is_grad_enabled() and (x.requires_grad or gn.weight.requires_grad)))


def groupnorm_flatten_into(x, gn, out, row_offset):
    """out[:, row_offset : row_offset + H*W, :] = gn(x).flatten(2).transpose(1, 2)  (out: [N, S, 128] contiguous)."""
    n, c, h, w = x.shape
    stats = torch.empty((n * gn.num_groups * 2,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().pct_groupnorm_flatten_f32(
            x.data_ptr(), gn.weight.data_ptr(), gn.bias.data_ptr(), n, c, h * w, gn.num_groups, float(gn.eps),
            stats.data_ptr(), out.data_ptr(), out.stride(0), row_offset * c,
            torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "groupnorm_flatten")


def lsap_supported(cost):
    return cost.is_cuda and cost.dim() == 3 and cost.shape[1] <= 1024 and cost.shape[2] <= 512


def lsap(cost, num_target):
    """Batched linear sum assignment on the device.  cost [B, Q, Gmax] (row = query, column = target), num_target [B]
    int32 device tensor with num_target[b] <= Q.  -> (row_for_target [B, Gmax] int32: the query matched to each target,
    -1 past num_target[b]; status [B] int32: 1 where no finite assignment exists).  No host synchronisation."""
    c = cost.detach().float().contiguous()
    B, Q, Gm = c.shape
    nt = num_target.to(device=c.device, dtype=torch.int32).contiguous()
    rows = torch.empty((B, Gm), dtype=torch.int32, device=c.device)
    status = torch.empty((B,), dtype=torch.int32, device=c.device)
    with torch.cuda.device(c.device):
        rc = _lib.lib().pct_lsap_f32(c.data_ptr(), B, Q, Gm, nt.data_ptr(), rows.data_ptr(), status.data_ptr(),
                                     torch.cuda.current_stream(c.device).cuda_stream)
    _lib.check(rc, "lsap")
    return rows, status


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
    with torch.cuda.device(q.device), _timing.timed("masked_attention S=%d" % S, qc):
        rc = _lib.lib().pct_masked_attention_bf16(
            qc.data_ptr(), kc.data_ptr(), vT.data_ptr(), m.data_ptr() if m is not None else None, N, num_heads, L, S,
            hd, Ev // num_heads, float(hd) ** -0.5, 2, out.data_ptr(), torch.cuda.current_stream(q.device).cuda_stream)
    _lib.check(rc, "masked_attention")
    return out


def cross_attention_supported(q_content, k_content, v, num_heads, attn_mask):
    """bf16 device tensors [tokens, N, heads*16], heads % 4 == 0, S % 64 == 0, boolean mask [N,1,L,S] (or none), forward only."""
    if not (q_content.is_cuda and q_content.dtype == torch.bfloat16 and k_content.dtype == torch.bfloat16
            and v.dtype == torch.bfloat16):
        return False
    L, N, C = q_content.shape
    S = k_content.shape[0]
    if C != num_heads * 16 or num_heads % 4 or S % 64 or k_content.shape != (S, N, C) or v.shape != (S, N, C):
        return False
    if attn_mask is not None and (attn_mask.dtype != torch.bool or tuple(attn_mask.shape) != (N, 1, L, S)):
        return False
    return True


def cross_attention(q_content, q_pos, k_content, k_pos, v, num_heads, attn_mask=None, row_open=None):
    """softmax(mask(q k^T / sqrt(32))) v per head with q = [q_content_h | q_pos_h], k = [k_content_h | k_pos_h] (16 + 16
    dims per head): all operands [tokens, N, heads*16] bf16 as the projections write them -> [L, N, heads*16] bf16.
    Bit-identical to masked_attention() on the per-head concatenations (csrc/cross_attention.hip).  `row_open`: bool [N, L] (or
    anything reshapeable to it), True = that query ignores its mask row -- same result as attn_mask & ~row_open[..., None]
    without writing that tensor."""
    L, N, C = q_content.shape
    S = k_content.shape[0]
    qc, qp, kc, kp, vv = (t.contiguous() for t in (q_content, q_pos, k_content, k_pos, v))
    m = attn_mask.reshape(N, L, S).contiguous() if attn_mask is not None else None
    ro = row_open.reshape(N, L).contiguous() if (row_open is not None and m is not None) else None
    if ro is not None and ro.dtype != torch.bool:
        raise RuntimeError("row_open must be a bool tensor")
    out = torch.empty((L, N, C), dtype=torch.bfloat16, device=qc.device)
    with torch.cuda.device(qc.device), _timing.timed("cross_attention S=%d" % S, qc):
        rc = _lib.lib().pct_cross_attention_bf16(
            qc.data_ptr(), qp.data_ptr(), kc.data_ptr(), kp.data_ptr(), vv.data_ptr(),
            m.data_ptr() if m is not None else None, ro.data_ptr() if ro is not None else None, N, num_heads, L, S,
            32.0 ** -0.5, out.data_ptr(),
            torch.cuda.current_stream(qc.device).cuda_stream)
    _lib.check(rc, "cross_attention")
    return out
