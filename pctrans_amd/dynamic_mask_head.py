"""Python entry of the fused dynamic mask head kernel (C ABI `pct_dynamic_mask_head_forward`,
include/pctrans_hip.h; kernel pctrans_amd/csrc/dyn_mask_head.hip).  Forward only: when gradients are needed the
decoder uses its differentiable batched formulation instead (mask2former_transformer_decoder.py in this package)."""
import os

import torch

from . import _lib, _timing

_OUT = {torch.float32: 0, torch.bfloat16: 2}


def supported(mask_feats, rel_coord=True):
    return mask_feats.is_cuda and mask_feats.dim() == 4 and mask_feats.shape[1] == 16


def fused_geometry(H, W, target_size):
    """The shapes csrc/dyn_mask_head_fused.hip covers (one launch, no logits workspace): 128-pixel-wide maps, a multiple
    of 8 rows, attention-mask target an exact 1/2, 1/4 or 1/8 of the map."""
    th, tw = int(target_size[0]), int(target_size[1])
    if W != 128 or H < 8 or H % 8 or th <= 0 or tw <= 0 or H % th or W % tw or H // th != W // tw:
        return False
    return H // th in (2, 4, 8)


def dynamic_mask_head_forward(mask_feats, ref_xy, params, stride, rel_coord, target_size, out_dtype=torch.float32,
                              feats_f32=None, kernel=None):
    """mask_feats [N, 16, H, W]; ref_xy [N, Q, 2] normalised (x, y); params [N, Q, G] in parse_dynamic_params order.
    -> (logits upsampled x2 [N, Q, 2H, 2W] in `out_dtype`, attention mask bool [N, Q, th*tw], True = may not attend).
    `feats_f32`: optional fp32 contiguous copy of mask_feats made by the caller (the decoder calls this 10 times on the
    same features).  `kernel` (bf16 outputs only; tests and A/B runs): None = the one-launch kernel where its geometry
    applies, else the two-launch MFMA path; "fused" / "mfma" / "valu" force one (PCT_DMH_KERNEL sets the default)."""
    if not mask_feats.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    N, C, H, W = mask_feats.shape
    Q = params.shape[1]
    G = (C + (2 if rel_coord else 0)) * 8 + 64 + 8 + 8 + 8 + 1
    if params.shape != (N, Q, G):
        raise RuntimeError("params must be [N, Q, %d], got %s" % (G, tuple(params.shape)))
    if out_dtype not in _OUT:
        raise RuntimeError("out_dtype must be float32 or bfloat16")
    feats = feats_f32 if feats_f32 is not None else mask_feats.detach().float().contiguous()
    assert feats.shape == mask_feats.shape and feats.dtype == torch.float32 and feats.is_contiguous()
    prm = params.detach().float().contiguous()
    ref = ref_xy.detach().float().contiguous() if rel_coord else None
    if rel_coord and tuple(ref.shape) != (N, Q, 2):
        raise RuntimeError("ref_xy must be [N, Q, 2]")
    th, tw = int(target_size[0]), int(target_size[1])
    up = torch.empty((N, Q, 2 * H, 2 * W), dtype=out_dtype, device=mask_feats.device)
    amask = torch.empty((N, Q, th * tw), dtype=torch.bool, device=mask_feats.device)
    if kernel is None:
        kernel = os.environ.get("PCT_DMH_KERNEL") or None
    if kernel not in (None, "fused", "mfma", "valu"):
        raise RuntimeError("kernel must be None, 'fused', 'mfma' or 'valu'")
    if out_dtype == torch.bfloat16 and kernel in (None, "fused") and (kernel == "fused" or fused_geometry(H, W, (th, tw))):
        # bf16-autocast configuration, one pass over the pixels: the logits plane never goes to memory
        ws = torch.empty((N * ((Q + 1) // 2) * 1280,), dtype=torch.int32, device=mask_feats.device)   # 5120 B per pair
        with torch.cuda.device(mask_feats.device), _timing.timed("mask_head one-pass", feats):
            rc = _lib.lib().pct_dynamic_mask_head_forward_fused_bf16(
                feats.data_ptr(), ref.data_ptr() if rel_coord else None, prm.data_ptr(), N, C, Q, H, W, int(stride),
                1 if rel_coord else 0, th, tw, ws.data_ptr(), up.data_ptr(), amask.data_ptr(),
                torch.cuda.current_stream(mask_feats.device).cuda_stream)
        _lib.check(rc, "dynamic_mask_head_forward_fused_bf16")
        return up, amask
    if out_dtype == torch.bfloat16 and kernel != "valu":
        # bf16-autocast configuration: MLP on MFMA + streaming resize (two launches, bf16 logits workspace)
        scratch = torch.empty((N, Q, H, W), dtype=torch.bfloat16, device=mask_feats.device)
        with torch.cuda.device(mask_feats.device), _timing.timed("mask_head two-launch", feats):
            rc = _lib.lib().pct_dynamic_mask_head_forward_mfma(
                feats.data_ptr(), ref.data_ptr() if rel_coord else None, prm.data_ptr(), N, C, Q, H, W, int(stride),
                1 if rel_coord else 0, th, tw, scratch.data_ptr(), up.data_ptr(), amask.data_ptr(),
                torch.cuda.current_stream(mask_feats.device).cuda_stream)
        _lib.check(rc, "dynamic_mask_head_forward_mfma")
        return up, amask
    with torch.cuda.device(mask_feats.device):
        rc = _lib.lib().pct_dynamic_mask_head_forward(
            feats.data_ptr(), ref.data_ptr() if rel_coord else None, prm.data_ptr(), N, C, Q, H, W, int(stride),
            1 if rel_coord else 0, th, tw, _OUT[out_dtype], up.data_ptr(), amask.data_ptr(),
            torch.cuda.current_stream(mask_feats.device).cuda_stream)
    _lib.check(rc, "dynamic_mask_head_forward")
    return up, amask
