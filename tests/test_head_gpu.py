"""GPU: the whole head (pixel decoder + transformer decoder) -- every fused HIP path (no_grad) against the same modules
evaluated with their torch formulations + the unfused op (grad enabled disables the forward-only kernels).
fp32: tight (every kernel is fp32-exact up to summation order); bf16 autocast: loose (bf16 rounding inside)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _head(levels, Q, seed=0):
    from pctrans_amd.config import get_cfg, resnet_output_shape
    from pctrans_amd.meta_arch.mask_former_head import MaskFormerHead
    torch.manual_seed(seed)
    feats = ("res2", "res3", "res4", "res5")[4 - levels:]
    cfg = get_cfg(num_queries=Q, enc_in_features=feats, norm="BN", sem_norm="BN")
    shapes = resnet_output_shape(18)
    head = MaskFormerHead(**MaskFormerHead.from_config(cfg, shapes)).cuda().eval()
    # leave the all-zero offset/attention init behind so that every query samples differently
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    with torch.no_grad():
        for layer in head.pixel_decoder.transformer.encoder.layers:
            layer.self_attn.sampling_offsets.weight.normal_(0, 0.02, generator=g)
            layer.self_attn.attention_weights.weight.normal_(0, 0.2, generator=g)
    return head, shapes


def _feats(shapes, N, H, W, seed=3):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return {k: torch.randn(N, s.channels, H // s.stride, W // s.stride, device="cuda", generator=g)
            for k, s in shapes.items()}


@pytest.mark.parametrize("levels,H,W", [(3, 256, 256), (4, 128, 160)])
def test_head_fp32_fused_equals_torch_formulation(levels, H, W):
    head, shapes = _head(levels, Q=20)
    feats = _feats(shapes, 2, H, W)
    with torch.no_grad():
        pred_f, mf_f = head(feats)                                   # fused kernels everywhere
    for p in head.parameters():
        p.requires_grad_(True)
    pred_t, mf_t = head(feats)                                       # torch formulations + autograd op
    assert float((mf_f - mf_t).abs().max()) < 1e-3 * max(1.0, float(mf_t.abs().max()))
    a, b = pred_f["pred_masks"], pred_t["pred_masks"].detach()
    scale = max(1.0, float(b.abs().max()))
    # the decoder thresholds masks at logit 0; a rare flipped attention-mask bit changes a query's attention, so
    # compare robustly: almost all elements within 1e-3 of scale
    close = ((a - b).abs() <= 2e-3 * scale).float().mean()
    assert float(close) > 0.995, float(close)
    assert float((pred_f["reference_points"] - pred_t["reference_points"].detach()).abs().max()) < 5e-3
    pred_t["pred_masks"].mean().backward()                            # the torch path is differentiable end to end
    assert head.pixel_decoder.transformer.encoder.layers[0].self_attn.value_proj.weight.grad is not None


def test_head_bf16_autocast_fused_runs_and_is_close_to_fp32():
    head, shapes = _head(4, Q=20)
    feats = _feats(shapes, 2, 128, 128)
    with torch.no_grad():
        pred32, _ = head(feats)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred16, mf16 = head(feats)
    assert mf16.dtype == torch.float32                               # pixel decoder stays fp32 (msdeformattn.py:314)
    assert pred16["pred_masks"].dtype == torch.bfloat16
    assert torch.isfinite(pred16["pred_masks"].float()).all()
    a, b = pred16["pred_masks"].float(), pred32["pred_masks"]
    scale = max(1.0, float(b.abs().max()))
    assert float(((a - b).abs() <= 0.1 * scale).float().mean()) > 0.9
