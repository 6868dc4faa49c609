"""MSDeformAttn nn.Module on the MI355X kernels.

Mirrors OPS/modules/ms_deform_attn.py:34-125 of the reference (OPS = connectomics/model/maskformer_block/
pixel_decoder/ops): same constructor, parameter names (state-dict keys `sampling_offsets`, `attention_weights`,
`value_proj`, `output_proj`), `_reset_parameters` recipe (:66-80) and forward signature (:82).

Difference by design: the reference wraps the op call in a bare `except:` that silently reroutes ANY failure to the
dense PyTorch formulation (:116-121).  Here device tensors always run the HIP kernel and every failure propagates.
CPU tensors raise "Not implemented on the CPU" exactly like the reference extension (ms_deform_attn.h:43) unless
the caller explicitly opted into the dense formulation with `allow_cpu_reference(True)` (host-logic tests only).
"""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from .... import MultiScaleDeformableAttention as MSDA
from .... import fused_ops
from ..functions import MSDeformAttnFunction, ms_deform_attn_core_pytorch

_ALLOW_CPU_REFERENCE = False


def allow_cpu_reference(flag=True):
    """Opt-in switch for CPU tensors (structural / host-logic tests on machines without a GPU)."""
    global _ALLOW_CPU_REFERENCE
    prev, _ALLOW_CPU_REFERENCE = _ALLOW_CPU_REFERENCE, bool(flag)
    return prev


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        _d_per_head = d_model // n_heads
        if not _is_power_of_2(_d_per_head):
            warnings.warn("MSDeformAttn: a per-head dimension that is a power of 2 (ideally a multiple of 4 fp32 / "
                          "8 half channels) maps best onto the 16-byte lane loads of the gfx950 kernel.")

        self.im2col_step = 128

        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)

        self._reset_parameters()

    def _reset_parameters(self):
        constant_(self.sampling_offsets.weight.data, 0.)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2)
        grid_init = grid_init.repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid_init.view(-1))
        constant_(self.attention_weights.weight.data, 0.)
        constant_(self.attention_weights.bias.data, 0.)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.)

    def _can_fuse(self, value, query, reference_points):
        if not value.is_cuda or (torch.is_grad_enabled() and (value.requires_grad or query.requires_grad)):
            return False
        return (value.dtype == torch.float32 and query.dtype == torch.float32 and reference_points.shape[-1] == 2
                and self.d_model // self.n_heads == 16 and self.n_points in (4, 8) and self.n_heads <= 16
                and value.shape[1] * value.shape[2] * value.shape[3] * 4 < 2 ** 31 - 1)   # per image: batches beyond 2 GiB are chunked in the library

    def forward_add_norm(self, query, reference_points, input_flatten, input_spatial_shapes,
                         input_level_start_index, input_padding_mask, residual, norm, query_pos=None):
        """norm(residual + self(query + query_pos, ...)) -- the encoder layer's `src2 = self_attn(with_pos_embed(src,
        pos), ...); src = norm1(src + dropout1(src2))` in eval mode (pixel_decoder/msdeformattn.py:112-119): when the
        fp32 forward-only path applies, `query + query_pos` is formed inside the projection kernels and output_proj,
        the residual add and the LayerNorm are one kernel."""
        return self.forward(query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                            input_padding_mask, _residual=residual, _norm=norm, _query_pos=query_pos)

    def _project_out(self, output, residual, norm):
        if norm is None:
            return fused_ops.linear(output, self.output_proj)
        return fused_ops.linear_add_layer_norm(output, self.output_proj, residual, norm)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, _residual=None, _norm=None, _query_pos=None):
        """query (N, Lq, C); reference_points (N, Lq, n_levels, 2|4) in [0,1]; input_flatten (N, sum H_l*W_l, C);
        input_spatial_shapes (n_levels, 2) as (H_l, W_l); input_level_start_index (n_levels,);
        input_padding_mask (N, sum H_l*W_l) True = padding.  Returns (N, Lq, C)."""
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        # the reference's check (ops/modules/ms_deform_attn.py:95) compares on the device and therefore blocks the host on
        # every call; a given shapes TENSOR OBJECT is checked once and the verdict is kept on that object together with
        # its version counter: a new tensor (even one that re-uses a freed allocation) or an in-place write is re-checked
        stamp = (input_spatial_shapes._version, Len_in)
        if getattr(input_spatial_shapes, "_pct_shapes_checked", None) != stamp:
            assert (input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum() == Len_in
            try:
                input_spatial_shapes._pct_shapes_checked = stamp
            except (AttributeError, RuntimeError):      # (a tensor subclass without a __dict__: checked every call)
                pass

        # encoder self-attention: the three input projections read the same rows (query is input_flatten): one launch
        merged = None
        if (query is input_flatten and input_padding_mask is None and reference_points.shape[-1] == 2
                and self.d_model // self.n_heads == 16 and self.n_points in (4, 8) and self.n_heads <= 16
                and fused_ops.linear_k128_multi_supported(
                    query, (self.value_proj, self.sampling_offsets, self.attention_weights), _query_pos)):
            merged = fused_ops.linear_k128_multi(
                query, ((self.value_proj, False), (self.sampling_offsets, True), (self.attention_weights, True)),
                x_add=_query_pos)
        value = merged[0] if merged is not None else \
            fused_ops.linear(input_flatten, self.value_proj)         # K = 128 MFMA kernel when forward-only fp32
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)
        if self._can_fuse(value, query, reference_points):
            # one launch: softmax + location math + sampling (no sampling_locations tensor, no softmax output)
            offsets = (merged[1] if merged is not None else
                       fused_ops.linear(query, self.sampling_offsets, x_add=_query_pos)).view(
                N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
            logits = (merged[2] if merged is not None else
                      fused_ops.linear(query, self.attention_weights, x_add=_query_pos)).view(
                N, Len_q, self.n_heads, self.n_levels * self.n_points)
            output = MSDA.ms_deform_attn_fused_forward(
                value.contiguous(), input_spatial_shapes, input_level_start_index, reference_points,
                offsets.contiguous(), logits.contiguous())
            return self._project_out(output, _residual, _norm)
        if _query_pos is not None:
            query = query + _query_pos
        sampling_offsets = self.sampling_offsets(query).view(
            N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
        attention_weights = self.attention_weights(query).view(
            N, Len_q, self.n_heads, self.n_levels * self.n_points)
        attention_weights = F.softmax(attention_weights, -1).view(
            N, Len_q, self.n_heads, self.n_levels, self.n_points)
        if reference_points.shape[-1] == 2:
            offset_normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            sampling_locations = reference_points[:, :, None, :, None, :] \
                + sampling_offsets / offset_normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            sampling_locations = reference_points[:, :, None, :, None, :2] \
                + sampling_offsets / self.n_points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
                reference_points.shape[-1]))
        if value.is_cuda:
            output = MSDeformAttnFunction.apply(
                value.contiguous(), input_spatial_shapes, input_level_start_index,
                sampling_locations.contiguous(), attention_weights.contiguous(), self.im2col_step)
        elif _ALLOW_CPU_REFERENCE:
            output = ms_deform_attn_core_pytorch(value, input_spatial_shapes, sampling_locations, attention_weights)
        else:
            raise RuntimeError("Not implemented on the CPU")
        output = self.output_proj(output)
        if _norm is not None:
            output = _norm(_residual + output)
        return output
