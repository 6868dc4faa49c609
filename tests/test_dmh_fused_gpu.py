"""GPU: the one-launch bf16 dynamic mask head (csrc/dyn_mask_head_fused.hip, C ABI
pct_dynamic_mask_head_forward_fused_bf16) against
  * the two-launch MFMA path it replaces (pct_dynamic_mask_head_forward_mfma): BIT-IDENTICAL upsampled logits and
    attention masks -- both evaluate the expression trees of csrc/dmh_common.hpp -- at the north-star map (128 x 128,
    the three attention-mask targets), odd query counts, one-band maps (top and bottom edge in the same workgroup),
    with and without relative coordinates;
  * the torch emulation of the reference's bf16-autocast pipeline (mask2former_transformer_decoder.py:647-719 under
    torch.autocast: bf16 conv operands, fp32 accumulation), same tolerance as the two-launch path's own test;
and the geometry contract of the entry point (everything else is PCT_ERR_UNSUPPORTED, callers use the two-launch path)."""
import pytest
import torch
from torch.nn import functional as F

from test_dyn_mask_head_gpu import _bf16_chain_reference, _case

pytestmark = pytest.mark.gpu


def _run(kernel, mf, ref, prm, rel, target):
    from pctrans_amd import dynamic_mask_head as dmh
    p = prm.transpose(0, 1)
    if not rel:                                    # parse_dynamic_params order without the two coordinate inputs
        Q, N = prm.shape[0], prm.shape[1]
        w0 = prm[..., :144].reshape(Q, N, 8, 18)[..., 2:].reshape(Q, N, 128)
        p = torch.cat([w0, prm[..., 144:]], dim=2).transpose(0, 1)
    return dmh.dynamic_mask_head_forward(mf, ref.transpose(0, 1), p.contiguous(), 4, rel, target,
                                         out_dtype=torch.bfloat16, kernel=kernel)


@pytest.mark.parametrize("rel", [True, False])
@pytest.mark.parametrize("N,Q,H,target", [
    (2, 100, 128, (16, 16)), (2, 100, 128, (32, 32)), (1, 100, 128, (64, 64)),      # the north-star map, its three targets
    (1, 101, 128, (32, 32)),                                                          # odd query count: a half-empty pair
    (2, 5, 8, (4, 64)), (1, 3, 8, (2, 32)), (3, 2, 8, (1, 16)),                       # one band: both edges in one workgroup
    (1, 9, 24, (12, 64)), (1, 9, 24, (6, 32)), (1, 9, 24, (3, 16)),                   # three bands
])
def test_one_launch_kernel_is_bit_identical_to_the_two_launch_path(rel, N, Q, H, target):
    mf, ref, prm = _case(N, Q, H, 128, seed=17 + Q + H)
    mf, ref, prm = mf.float().cuda(), ref.float().cuda(), prm.float().cuda()
    up_f, m_f = _run("fused", mf, ref, prm, rel, target)
    up_m, m_m = _run("mfma", mf, ref, prm, rel, target)
    assert up_f.dtype == torch.bfloat16 and up_f.shape == (N, Q, 2 * H, 256) and m_f.shape == (N, Q, target[0] * target[1])
    assert torch.isfinite(up_m.float()).all()
    assert torch.equal(up_f.view(torch.int16), up_m.view(torch.int16)), \
        float((up_f.float() - up_m.float()).abs().max())
    assert torch.equal(m_f, m_m), int((m_f != m_m).sum())
    assert 0.02 < float(m_f.float().mean()) < 0.98          # the masks are not trivially constant


def test_one_launch_kernel_matches_the_autocast_semantics():
    N, Q, H, W, target = 1, 101, 128, 128, (64, 64)
    mf, ref, prm = _case(N, Q, H, W, seed=5)
    mf, ref, prm = mf.float().cuda(), ref.float().cuda(), prm.float().cuda()
    want = _bf16_chain_reference(mf, ref, prm)
    up, amask = _run("fused", mf, ref, prm, True, target)
    lb = want.bfloat16()
    want_up = F.interpolate(lb, size=(2 * H, 2 * W), mode="bilinear", align_corners=False)
    diff = (up.float() - want_up.float()).abs()
    tol = want_up.float().abs() * 2.0 ** -6 + 3e-2
    assert (diff <= tol).float().mean() > 0.995, float((diff <= tol).float().mean())
    want_m = F.interpolate(lb, size=target, mode="bilinear", align_corners=False).sigmoid().flatten(2) < 0.5
    assert (amask != want_m).float().mean() < 1e-2


def test_default_route_takes_the_one_launch_kernel_only_where_its_geometry_applies():
    from pctrans_amd import dynamic_mask_head as dmh
    assert dmh.fused_geometry(128, 128, (16, 16)) and dmh.fused_geometry(128, 128, (64, 64))
    assert dmh.fused_geometry(8, 128, (4, 64)) and dmh.fused_geometry(136, 128, (17, 16))
    for H, W, t in ((64, 64, (16, 16)), (128, 128, (128, 128)), (128, 128, (8, 8)), (130, 128, (65, 64)),
                    (128, 128, (32, 64)), (65, 87, (17, 22)), (4, 128, (2, 64))):
        assert not dmh.fused_geometry(H, W, t), (H, W, t)
    # forcing it on a geometry it does not cover is an error from the C ABI, never a silent other kernel
    mf, ref, prm = _case(1, 4, 64, 64, seed=3)
    mf, ref, prm = mf.float().cuda(), ref.float().cuda(), prm.float().cuda()
    with pytest.raises(RuntimeError, match="not supported"):
        _run("fused", mf, ref, prm, True, (16, 16))
    up_d, m_d = _run(None, mf, ref, prm, True, (16, 16))              # default route: the two-launch path
    up_m, m_m = _run("mfma", mf, ref, prm, True, (16, 16))
    assert torch.equal(up_d.view(torch.int16), up_m.view(torch.int16)) and torch.equal(m_d, m_m)
