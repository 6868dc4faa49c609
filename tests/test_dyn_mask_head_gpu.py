"""GPU: the fused dynamic-mask-head kernel (through the C ABI) against the literal restatement of the reference's
formulation (materialised per-query inputs + grouped convs, mask2former_transformer_decoder.py:647-719) built from
torch ops in float64 on the host, at the reference's shapes: 128x128 / 64x64 feature maps, Q = 100 / 300, the three
attention-mask target sizes, non-square BBBC maps.  Tolerance 1e-4 absolute on O(1..10) logits in fp32 (north_star);
mask bits may differ only where the resized logit is within 1e-4 of the threshold."""
import numpy as np
import pytest
import torch
from torch.nn import functional as F

from test_head_cpu import _reference_formulation, _small_decoder

pytestmark = pytest.mark.gpu


def _case(N, Q, H, W, seed, scale=0.3):
    g = torch.Generator().manual_seed(seed)
    mf = torch.randn(N, 16, H, W, generator=g, dtype=torch.float64)
    ref = torch.rand(Q, N, 2, generator=g, dtype=torch.float64)
    prm = torch.randn(Q, N, 233, generator=g, dtype=torch.float64) * scale
    w0 = prm[..., :144].view(Q, N, 8, 18)
    w0[..., :2] *= 0.02           # rel-coord weights see inputs up to +-512
    return mf, ref, prm


@pytest.mark.parametrize("N,Q,H,W,target", [
    (2, 100, 128, 128, (16, 16)), (2, 100, 128, 128, (32, 32)), (1, 100, 128, 128, (64, 64)),
    (2, 300, 65, 87, (17, 22)), (1, 50, 32, 32, (8, 8)), (3, 7, 9, 6, (5, 4)), (1, 10, 200, 128, (25, 16)),
])
def test_fused_mask_head_fp32(N, Q, H, W, target):
    from pctrans_amd import dynamic_mask_head as dmh
    d = _small_decoder(Q=Q).double()
    mf, ref, prm = _case(N, Q, H, W, seed=N * 1000 + Q)
    want = _reference_formulation(d, mf, ref, prm)                                   # [N, Q, H, W] float64
    want_up = F.interpolate(want, size=(2 * H, 2 * W), mode="bilinear", align_corners=False)
    want_rs = F.interpolate(want, size=target, mode="bilinear", align_corners=False).flatten(2)
    up, amask = dmh.dynamic_mask_head_forward(mf.float().cuda(), ref.transpose(0, 1).float().cuda(),
                                              prm.transpose(0, 1).float().cuda(), 4, True, target)
    assert up.shape == (N, Q, 2 * H, 2 * W) and up.dtype == torch.float32
    scale = max(1.0, float(want_up.abs().max()))
    err = float((up.cpu().double() - want_up).abs().max())
    assert err <= 1e-4 * scale, (err, scale)
    got_m = amask.cpu()
    want_m = want_rs.sigmoid() < 0.5
    differ = got_m != want_m
    assert float(want_rs[differ].abs().max() if differ.any() else 0.0) < 1e-4 * scale
    assert differ.float().mean() < 1e-3


def test_fused_mask_head_bf16_output_matches_autocast_semantics():
    """bf16 mode: logits rounded to bf16 (as the reference's autocast convs emit), resized with fp32 math, stored
    as bf16.  Compared with the same pipeline built from torch bf16 ops on the fp32 kernel's logits."""
    from pctrans_amd import dynamic_mask_head as dmh
    N, Q, H, W, target = 2, 100, 64, 64, (16, 16)
    d = _small_decoder(Q=Q).double()
    mf, ref, prm = _case(N, Q, H, W, seed=5)
    want = _reference_formulation(d, mf, ref, prm).float().cuda()
    args = (mf.float().cuda(), ref.transpose(0, 1).float().cuda(), prm.transpose(0, 1).float().cuda(), 4, True, target)
    up, amask = dmh.dynamic_mask_head_forward(*args, out_dtype=torch.bfloat16)
    assert up.dtype == torch.bfloat16
    lb = want.bfloat16()
    want_up = F.interpolate(lb, size=(2 * H, 2 * W), mode="bilinear", align_corners=False)
    # logits within 1 bf16 ulp of a rounding boundary may round differently: allow 2^-7 relative on a tiny fraction
    diff = (up.float() - want_up.float()).abs()
    tol = want_up.float().abs() * 2.0 ** -7 + 1e-3
    assert (diff <= tol).float().mean() > 0.999
    want_m = F.interpolate(lb, size=target, mode="bilinear", align_corners=False).sigmoid().flatten(2) < 0.5
    assert (amask != want_m).float().mean() < 5e-3


def test_decoder_uses_fused_kernel_and_matches_batched_formulation():
    """End to end inside the decoder: eval forward on the GPU (fused kernel) == the differentiable batched torch
    formulation run on the same device (grad enabled -> torch path)."""
    from pctrans_amd.transformer_decoder import mask2former_transformer_decoder as dec
    torch.manual_seed(0)
    d = dec.MultiScaleMaskedTransformerDecoder(
        128, True, hidden_dim=128, num_queries=20, nheads=8, dim_feedforward=256, dec_layers=3, pre_norm=False,
        mask_dim=16, enforce_input_project=False, points_num=1, sem_loss_on=True, norm="BN", rel_coord=True
    ).cuda().eval()
    mf = torch.randn(2, 16, 32, 40, device="cuda")
    ref = torch.rand(20, 2, 2, device="cuda")
    prm = torch.randn(20, 2, 233, device="cuda") * 0.2
    with torch.no_grad():
        up_f, m_f = d.dynamic_mask_with_coords(mf, ref, prm, 4, True, (8, 10))
    up_t, m_t = d.dynamic_mask_with_coords(mf, ref, prm.requires_grad_(), 4, True, (8, 10))
    assert m_f.shape == m_t.shape == (2, 1, 20, 80)
    assert float((up_f - up_t).abs().max()) <= 1e-4 * max(1.0, float(up_t.abs().max()))
    assert (m_f != m_t).float().mean() < 1e-3
