"""GPU: the MSDeformAttn forward on PIECE-PLANE operands (pct_ms_deform_attn_forward_planes_f32; msda_forward_col.hip, PP)
must give, bit for bit, what the pyramid-column kernel gives on the same numbers in the reference layout
(ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304) -- which tests/test_msda_col_gpu.py holds against the C oracle -- and is
checked against the oracle directly as well; plain op and fused front-end (ops/modules/ms_deform_attn.py:100-118)."""
import numpy as np
import pytest
import torch

from msda_cases import make_case
from oracle import msda_oracle as orc
from test_msda_col_gpu import COL_CASES, K_COL, dev, force

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def MSDA():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from pctrans_amd import MultiScaleDeformableAttention as m
    from pctrans_amd import _lib
    _lib.lib()
    return m


@pytest.fixture(scope="module")
def lib():
    from pctrans_amd import _lib
    return _lib.lib()


def test_to_planes_round_trip(MSDA):
    t = torch.arange(2 * 5 * 3 * 8, dtype=torch.float32, device="cuda").view(2, 5, 3, 8)
    p = MSDA.to_planes(t, 3)
    assert tuple(p.shape) == (2, 6, 5, 4)
    assert float(p[1, 3, 2, 1]) == float(t[1, 2, 1, 5])          # plane 3 = head 1, piece 1 -> elements 4..7
    assert torch.equal(MSDA.from_planes(p, 3), t)


@pytest.mark.parametrize("cid,kw", COL_CASES, ids=[c[0] for c in COL_CASES])
def test_planes_entry_is_bit_identical_to_the_reference_layout(MSDA, lib, cid, kw):
    kw = dict(dict(M=8, D=16, P=4), **kw)
    atol = kw.pop("atol", 1e-4)
    c = make_case(dtype=np.float32, **kw)
    value, shapes, starts, loc, attn = (dev(c[k]) for k in ("value", "shapes", "starts", "loc", "attn"))
    M = value.shape[2]
    with force(lib, K_COL):
        want = MSDA.ms_deform_attn_forward(value, shapes, starts, loc, attn, 64)
        assert lib.pct_msda_last_kernel() == K_COL
    got = MSDA.ms_deform_attn_forward_planes(MSDA.to_planes(value, M), shapes, starts, MSDA.to_planes(loc, M),
                                             MSDA.to_planes(attn, M), M)
    assert lib.pct_msda_last_kernel() == K_COL
    assert torch.equal(got.view(torch.int32), want.view(torch.int32)), float((got - want).abs().max())
    ref = orc.forward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"])
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=atol)


@pytest.mark.parametrize("shapes,N,M,sigma", [
    ([(16, 16), (32, 32), (64, 64), (128, 128)], 2, 8, 2.0), ([(17, 22), (33, 44), (65, 87)], 2, 8, 2.0),
    ([(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], 1, 8, 1.0), ([(16, 20), (40, 50), (9, 9)], 2, 4, 6.0),
])
def test_planes_entry_with_the_fused_front_end(MSDA, lib, shapes, N, M, sigma):
    """Raw offsets + logits + reference points on piece planes == the fused entry on the reference layout, bit for bit."""
    g = torch.Generator(device="cuda").manual_seed(len(shapes) * 100 + N)
    sh = torch.tensor(shapes, dtype=torch.long, device="cuda")
    L, P, D = len(shapes), 4, 16
    S = int(sh.prod(1).sum())
    st = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    value = torch.randn(N, S, M, D, device="cuda", generator=g)
    refs = []
    for h, w in shapes:
        ys, xs = torch.meshgrid((torch.arange(h, device="cuda") + 0.5) / h, (torch.arange(w, device="cuda") + 0.5) / w,
                                indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0)[None, :, None, :].expand(1, S, L, 2).contiguous()
    off = torch.randn(N, S, M, L, P, 2, device="cuda", generator=g) * sigma
    logits = torch.randn(N, S, M, L * P, device="cuda", generator=g) * 2.0
    with force(lib, K_COL):
        want = MSDA.ms_deform_attn_fused_forward(value, sh, st, ref.expand(N, -1, -1, -1), off, logits)
        assert lib.pct_msda_last_kernel() == K_COL
    got = MSDA.ms_deform_attn_forward_planes(MSDA.to_planes(value, M), sh, st, MSDA.to_planes(off, M),
                                             MSDA.to_planes(logits, M), M, reference_points=ref)
    assert torch.isfinite(got).all()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32)), float((got - want).abs().max())


def test_planes_entry_contract(MSDA):
    sh = torch.tensor([(4, 4), (8, 8), (16, 16)], dtype=torch.long, device="cuda")
    st = torch.tensor([0, 16, 80], dtype=torch.long, device="cuda")
    S, M = 336, 8
    v = torch.zeros(1, M * 4, S, 4, device="cuda")
    loc = torch.zeros(1, M * 3 * 4 // 2, S, 4, device="cuda")
    w = torch.zeros(1, M * 3 * 4 // 4, S, 4, device="cuda")
    out = MSDA.ms_deform_attn_forward_planes(v, sh, st, loc, w, M)
    assert tuple(out.shape) == (1, S, M * 16) and float(out.abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="inconsistent"):
        MSDA.ms_deform_attn_forward_planes(v, sh, st, loc[:, :-1].contiguous(), w, M)
    with pytest.raises(RuntimeError, match="float32"):
        MSDA.ms_deform_attn_forward_planes(v.double(), sh, st, loc, w, M)
    with pytest.raises(RuntimeError, match="CPU"):
        MSDA.ms_deform_attn_forward_planes(v.cpu(), sh.cpu(), st.cpu(), loc.cpu(), w.cpu(), M)
