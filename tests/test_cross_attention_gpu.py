"""GPU: the decoder's cross-attention kernel on split operands (csrc/cross_attention.hip, C ABI pct_cross_attention_bf16)
against
  * pct_masked_attention_bf16 on the per-head concatenations the reference builds
    (mask2former_transformer_decoder.py:160-172): BIT-IDENTICAL outputs -- both kernels run the online softmax of
    csrc/attn_common.hpp over the same 32-key steps -- at the decoder's three levels (256 / 1 024 / 4 096 keys), 100 and 300
    queries (a partial last query tile, a half-empty tile group), masked and unmasked, rows whose only live key is the
    first / the last one;
  * the float formulation of attention.py:271-387 (softmax(mask(q k^T / sqrt(32))) v per head) at bf16 tolerance;
and the geometry contract of the entry point."""
import pytest
import torch

from test_fused_ops_gpu import _attn_reference

pytestmark = pytest.mark.gpu


def _operands(L, S, N, heads, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    C = heads * 16
    mk = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
    return mk(L, N, C), mk(L, N, C), mk(S, N, C), mk(S, N, C), mk(S, N, C)


def _cat_heads(a, b, heads):
    T, N, C = a.shape
    return torch.cat([a.view(T, N, heads, 16), b.view(T, N, heads, 16)], dim=3).reshape(T, N, 2 * C)


@pytest.mark.parametrize("L,S,N,heads,masked", [
    (100, 4096, 2, 8, True), (100, 1024, 3, 8, True), (100, 256, 2, 8, True), (300, 1024, 2, 8, True),
    (100, 1024, 2, 8, False), (37, 64, 1, 4, True), (16, 128, 1, 8, False), (65, 192, 2, 12, True),
])
@pytest.mark.parametrize("seed_offset", [0, 2, 18])
def test_split_operand_kernel_is_bit_identical_to_the_concatenated_form(L, S, N, heads, masked, seed_offset):
    from pctrans_amd import fused_ops
    qc, qp, kc, kp, v = _operands(L, S, N, heads, seed=L + S)
    mask = None
    if masked:
        gm = torch.Generator(device="cuda").manual_seed(1000 + L + S + seed_offset)
        mask = torch.rand(N, 1, L, S, device="cuda", generator=gm) < 0.7
        mask[..., 0] = False                               # no fully masked row
        mask[0, 0, 1, :] = True
        mask[0, 0, 1, S - 1] = False                       # a row whose only live key is the very last one
        mask[0, 0, 2, 1:] = True                           # ... and one whose only live key is the first
    assert fused_ops.cross_attention_supported(qc, kc, v, heads, mask)
    got = fused_ops.cross_attention(qc, qp, kc, kp, v, heads, mask)
    q, k = _cat_heads(qc, qp, heads), _cat_heads(kc, kp, heads)
    ref_kernel = fused_ops.masked_attention(q, k, v, heads, mask)
    assert got.shape == (L, N, heads * 16) and got.dtype == torch.bfloat16
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got.view(torch.int16), ref_kernel.view(torch.int16)), float((got.float() - ref_kernel.float()).abs().max())
    want = _attn_reference(q, k, v, heads, mask)
    err = (got.float() - want).abs()
    assert float(err.max()) < 3e-2 and float(err.mean()) < 3e-3, (float(err.max()), float(err.mean()))


def test_fully_masked_rows_give_nan_like_the_softmax_of_all_minus_infinity():
    from pctrans_amd import fused_ops
    L, S, N, heads = 20, 128, 1, 8
    qc, qp, kc, kp, v = _operands(L, S, N, heads, seed=1)
    mask = torch.zeros(N, 1, L, S, dtype=torch.bool, device="cuda")
    mask[0, 0, 5, :] = True
    got = fused_ops.cross_attention(qc, qp, kc, kp, v, heads, mask).float()
    assert torch.isnan(got[5]).all() and torch.isfinite(got[:5]).all() and torch.isfinite(got[6:]).all()


@pytest.mark.parametrize("L,S,N,heads", [(100, 1024, 3, 8), (300, 256, 2, 8), (37, 64, 1, 4)])
def test_row_open_equals_clearing_those_mask_rows(L, S, N, heads):
    """The decoder's rule (:561) as a per-query flag: bit-identical to the mask with those rows cleared, for rows that are
    fully masked (the decoder's use), partly masked, and with no flag set at all."""
    from pctrans_amd import fused_ops
    qc, qp, kc, kp, v = _operands(L, S, N, heads, seed=7 + L)
    gm = torch.Generator(device="cuda").manual_seed(L * S)
    mask = torch.rand(N, 1, L, S, device="cuda", generator=gm) < 0.6
    mask[..., 3] = False
    mask[0, 0, 4, :] = True                                 # fully masked rows: first tile, last (partial) tile
    mask[N - 1, 0, L - 1, :] = True
    row_open = mask.all(dim=-1, keepdim=True)
    row_open[0, 0, 9, 0] = True                             # and a partly masked row opened as well
    assert int(row_open.sum()) == 3
    got = fused_ops.cross_attention(qc, qp, kc, kp, v, heads, mask, row_open=row_open)
    want = fused_ops.cross_attention(qc, qp, kc, kp, v, heads, mask & ~row_open)
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    none = fused_ops.cross_attention(qc, qp, kc, kp, v, heads, mask & ~row_open, row_open=torch.zeros_like(row_open))
    assert torch.equal(none.view(torch.int16), want.view(torch.int16))
    with pytest.raises(RuntimeError, match="bool"):
        fused_ops.cross_attention(qc, qp, kc, kp, v, heads, mask, row_open=row_open.to(torch.uint8))


def test_geometry_contract():
    from pctrans_amd import fused_ops
    qc, qp, kc, kp, v = _operands(10, 96, 1, 8, seed=2)                   # 96 keys: not a multiple of 64
    assert not fused_ops.cross_attention_supported(qc, kc, v, 8, None)
    with pytest.raises(RuntimeError, match="not supported"):
        fused_ops.cross_attention(qc, qp, kc, kp, v, 8, None)
    qc, qp, kc, kp, v = _operands(10, 64, 1, 6, seed=2)                   # 6 heads: not a multiple of 4
    assert not fused_ops.cross_attention_supported(qc, kc, v, 6, None)
