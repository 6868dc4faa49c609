"""Autograd glue for the MI355X MSDeformAttn kernels.

Mirrors OPS/functions/ms_deform_attn_func.py of the reference
(OPS = connectomics/model/maskformer_block/pixel_decoder/ops):
    MSDeformAttnFunction          :32-49   same apply() signature, same saved tensors, same returned grads
    ms_deform_attn_core_pytorch   :52-72   debug/test-only dense formulation (never used by the product path)
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .... import MultiScaleDeformableAttention as MSDA


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        output = MSDA.ms_deform_attn_forward(
            value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
            ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights = \
            ctx.saved_tensors
        if value.dtype in (torch.float16, torch.bfloat16):
            # 16-bit forward is new capability; its backward runs in fp32 (the reference has no 16-bit path at all)
            gv, gl, ga = MSDA.ms_deform_attn_backward(
                value.float(), value_spatial_shapes, value_level_start_index, sampling_locations.float(),
                attention_weights.float(), grad_output.float().contiguous(), ctx.im2col_step)
            return (gv.to(value.dtype), None, None, gl.to(sampling_locations.dtype),
                    ga.to(attention_weights.dtype), None)
        grad_value, grad_sampling_loc, grad_attn_weight = MSDA.ms_deform_attn_backward(
            value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
            grad_output.contiguous(), ctx.im2col_step)
        return grad_value, None, None, grad_sampling_loc, grad_attn_weight, None


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """Dense torch formulation of the op (per level: grid_sample with align_corners=False and zero padding, then the
    attention-weighted sum over levels x points).  Debug / cross-check only, like the reference's function of the
    same name; pctrans_amd never routes device tensors through it."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in value_spatial_shapes]
    per_level = value.split(sizes, dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(value_spatial_shapes):
        h, w = int(h), int(w)
        v = per_level[lvl].permute(0, 2, 3, 1).reshape(N * M, D, h, w)
        g = grids[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(N * M, Lq, P, 2)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    sampled = torch.stack(sampled, dim=-2).flatten(-2)                    # [N*M, D, Lq, L*P]
    w = attention_weights.permute(0, 2, 1, 3, 4).reshape(N * M, 1, Lq, L * P)
    out = (sampled * w).sum(-1).view(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()
