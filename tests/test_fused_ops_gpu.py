"""GPU: small fused kernels against the torch expressions they replace (fp32, tolerance 1e-5 relative to scale)."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

from pctrans_amd import fused_ops  # noqa: E402


@pytest.mark.parametrize("rows,cols", [(348160 // 8, 128), (1600, 128), (7, 128), (1000, 256), (33, 64)])
@pytest.mark.parametrize("with_y", [True, False])
def test_add_layer_norm(rows, cols, with_y):
    from pctrans_amd import fused_ops
    torch.manual_seed(0)
    ln = nn.LayerNorm(cols).cuda()
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.3)
        ln.bias.normal_(0.0, 0.3)
    x = torch.randn(rows, cols, device="cuda") * 3 + 1
    y = torch.randn(rows, cols, device="cuda") if with_y else None
    with torch.no_grad():
        got = fused_ops.add_layer_norm(x, y, ln)
        want = ln(x + y if with_y else x)
        want64 = nn.functional.layer_norm((x + y if with_y else x).double(), (cols,), ln.weight.double(),
                                          ln.bias.double(), ln.eps)
    assert float((got - want).abs().max()) < 2e-5
    assert float((got.double() - want64).abs().max()) <= float((want.double() - want64).abs().max()) + 2e-6
    # grad required -> torch expression (autograd works)
    x.requires_grad_()
    out = fused_ops.add_layer_norm(x, y, ln)
    out.sum().backward()
    assert x.grad is not None


def test_linear_relu_epilogue():
    from pctrans_amd import fused_ops
    torch.manual_seed(1)
    lin = nn.Linear(128, 1024).cuda()
    x = torch.randn(3, 4353, 128, device="cuda")
    with torch.no_grad():
        got = fused_ops.linear_relu(x, lin)
        want = torch.relu(lin(x))
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < 1e-4


def test_encoder_layer_fused_equals_unfused():
    from pctrans_amd.pixel_decoder.msdeformattn import MSDeformAttnTransformerEncoderLayer
    torch.manual_seed(2)
    layer = MSDeformAttnTransformerEncoderLayer(128, 1024, 0.0, "relu", 3, 8, 4).cuda().eval()
    shapes = torch.tensor([[4, 5], [8, 10], [16, 20]], device="cuda")
    starts = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    src = torch.randn(2, S, 128, device="cuda")
    pos = torch.randn(2, S, 128, device="cuda")
    ref = torch.rand(2, S, 3, 2, device="cuda")
    with torch.no_grad():
        a = layer(src, pos, ref, shapes, starts)                      # fused kernels
    b = layer(src.clone().requires_grad_(), pos, ref, shapes, starts)  # torch expressions + unfused op
    assert float((a - b).abs().max()) < 2e-4


# ---- fused masked attention (MFMA) -------------------------------------------------------------------------------
def _attn_reference(q, k, v, heads, mask):
    """fp32 math on the bf16-valued inputs: softmax(mask(q k^T * hd^-0.5)) v per head."""
    L, N, E = q.shape
    S, Ev = k.shape[0], v.shape[2]
    hd, vd = E // heads, Ev // heads
    qh = q.float().reshape(L, N, heads, hd).permute(1, 2, 0, 3)
    kh = k.float().reshape(S, N, heads, hd).permute(1, 2, 3, 0)
    vh = v.float().reshape(S, N, heads, vd).permute(1, 2, 0, 3)
    s = torch.matmul(qh, kh) * hd ** -0.5
    if mask is not None:
        s = s.masked_fill(mask, float("-inf"))
    return torch.matmul(torch.softmax(s, -1), vh).permute(2, 0, 1, 3).reshape(L, N, Ev)


@pytest.mark.parametrize("L,S,N,E,masked", [
    (100, 4096, 2, 256, True), (100, 1024, 3, 256, True), (100, 256, 2, 256, True), (300, 374, 2, 256, True),
    (100, 100, 4, 128, False), (37, 50, 1, 256, True), (16, 32, 1, 128, False),
])
def test_masked_attention_mfma(L, S, N, E, masked):
    from pctrans_amd import fused_ops
    torch.manual_seed(L + S)
    heads = 8
    q = torch.randn(L, N, E, device="cuda").bfloat16()
    k = torch.randn(S, N, E, device="cuda").bfloat16()
    v = torch.randn(S, N, 128, device="cuda").bfloat16()
    mask = None
    if masked:
        mask = torch.rand(N, 1, L, S, device="cuda") < 0.7
        mask[..., 0] = False                               # no fully masked row
        mask[0, 0, 1, :] = True
        mask[0, 0, 1, S - 1] = False                       # a row whose only live key is the very last one
    assert fused_ops.masked_attention_supported(q, k, v, heads, mask, None, 0.0, False, False)
    got = fused_ops.masked_attention(q, k, v, heads, mask).float()
    want = _attn_reference(q, k, v, heads, mask)
    # P is rounded to bf16 before P.V (as under autocast) and the output is bf16: 2^-8 relative + accumulation
    err = (got - want).abs()
    assert float(err.max()) < 3e-2 and float(err.mean()) < 3e-3, (float(err.max()), float(err.mean()))


def test_attention_core_dispatches_to_mfma_kernel_and_matches_torch_path():
    from pctrans_amd.transformer_decoder.attention import attention_core
    torch.manual_seed(3)
    L, S, N = 100, 1024, 2
    q = torch.randn(L, N, 256, device="cuda").bfloat16()
    k = torch.randn(S, N, 256, device="cuda").bfloat16()
    v = torch.randn(S, N, 128, device="cuda").bfloat16()
    mask = torch.rand(N, 1, L, S, device="cuda") < 0.5
    mask[..., 0] = False
    with torch.no_grad():
        a, _ = attention_core(q, k, v, 8, attn_mask=mask)                 # MFMA kernel
    b, _ = attention_core(q.float(), k.float(), v.float(), 8, attn_mask=mask)   # torch path (fp32 inputs)
    assert a.dtype == torch.bfloat16
    assert float((a.float() - b).abs().max()) < 3e-2
    # fully masked row -> NaN in both
    mask2 = mask.clone()
    mask2[1, 0, 5, :] = True
    with torch.no_grad():
        c, _ = attention_core(q, k, v, 8, attn_mask=mask2)
    assert torch.isnan(c[5, 1]).all() and torch.isfinite(c[4, 1]).all()


@pytest.mark.parametrize("rows,n,relu,bias", [(4096, 128, False, True), (5000, 256, True, True), (33, 128, False, False),
                                              (21760 * 2, 384, False, True), (4097, 1024, True, True), (5376, 192, False, True),
                                              (5376, 96, False, True), (3000, 640, False, True)])
def test_linear_k128_matches_fp64_matmul(rows, n, relu, bias):
    """Hand-written fp32 MFMA GEMM (k-permuted operands, accumulator-layout epilogue) against an fp64 product; the
    tolerance is the fp32 accumulation noise of a 128-term dot product."""
    g = torch.Generator(device="cuda").manual_seed(rows + n)
    x = torch.randn(rows, 128, device="cuda", generator=g)
    w = torch.randn(n, 128, device="cuda", generator=g) * 0.1
    b = torch.randn(n, device="cuda", generator=g) if bias else None
    got = fused_ops.linear_k128(x, w, b, relu=relu)
    want = x.double() @ w.double().t()
    if bias:
        want = want + b.double()
    if relu:
        want = want.relu()
    assert got.shape == (rows, n)
    torch.testing.assert_close(got.double(), want, rtol=0, atol=2e-5)
    # and it is at least as accurate as the library GEMM it replaces
    lib = torch.nn.functional.linear(x, w, b)
    if relu:
        lib = lib.relu()
    assert (got.double() - want).abs().max() <= 2.0 * (lib.double() - want).abs().max() + 1e-6


def test_linear_k128_split_products_keep_fp32_accuracy_over_the_exponent_range():
    """The default kernel evaluates x·Wᵀ on the bf16 matrix cores from the exact three-way bf16 split of every fp32
    operand (six leading partial products, fp32 accumulation).  Per-row / per-column scales from 1e-12 to 1e12 and
    heavy cancellation: the error relative to sum_k |x_k w_k| must stay at the level of an fp32 dot product
    (128 terms: a few 2^-24), far below what any single bf16 rounding (2^-9) would leave."""
    g = torch.Generator(device="cuda").manual_seed(5)
    rows, n = 8192, 256
    x = torch.randn(rows, 128, device="cuda", generator=g)
    w = torch.randn(n, 128, device="cuda", generator=g)
    x = x * torch.logspace(-12, 12, rows, device="cuda")[torch.randperm(rows, device="cuda", generator=g)][:, None]
    w = w * torch.logspace(-12, 12, n, device="cuda")[:, None]
    x[:, 64:] = -x[:, :64] * (1 + 1e-3 * torch.randn(rows, 64, device="cuda", generator=g))     # cancellation
    w[:, 64:] = w[:, :64]
    got = fused_ops.linear_k128(x, w)
    want = x.double() @ w.double().t()
    scale = x.double().abs() @ w.double().abs().t()
    err = ((got.double() - want).abs() / scale).max().item()
    lib = ((torch.nn.functional.linear(x, w).double() - want).abs() / scale).max().item()
    assert err < 4e-7, err                                  # fp32: unit roundoff 6e-8, 128-term accumulation
    assert err <= 2.0 * lib + 1e-8, (err, lib)


@pytest.mark.parametrize("S,shared", [(5376, True), (4099, False)])
def test_linear_k128_multi_equals_the_separate_launches(S, shared):
    """value_proj(src), sampling_offsets(src + pos), attention_weights(src + pos) in one launch (the rows are read once):
    same kernel arithmetic per output column, so the results are bit-identical to three launches."""
    torch.manual_seed(S)
    x = torch.randn(3, S, 128, device="cuda")
    pos = torch.randn(1 if shared else 3, S, 128, device="cuda")
    lv, lo, la = (torch.nn.Linear(128, n).cuda() for n in (128, 256, 128))
    with torch.no_grad():
        assert fused_ops.linear_k128_multi_supported(x, (lv, lo, la), pos)
        got = fused_ops.linear_k128_multi(x, ((lv, False), (lo, True), (la, True)), x_add=pos)
        want = [fused_ops.linear_k128(x, lv.weight, lv.bias),
                fused_ops.linear_k128(x, lo.weight, lo.bias, x_add=pos),
                fused_ops.linear_k128(x, la.weight, la.bias, x_add=pos)]
        ref64 = (x.double() + pos.double()) @ lo.weight.double().t() + lo.bias.double()
    for g, w in zip(got, want):
        assert g.shape == w.shape and torch.equal(g, w)
    torch.testing.assert_close(got[1].double(), ref64, rtol=0, atol=3e-5)


def test_linear_k128_unaligned_bias_takes_the_fp32_mfma_kernel():
    """The split kernel's epilogue is dwordx4: a bias that is only 4-byte aligned makes the launcher fall back to the
    fp32-MFMA kernel (dword epilogue) instead of faulting or refusing."""
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(4100, 128, device="cuda", generator=g)
    w = torch.randn(256, 128, device="cuda", generator=g) * 0.1
    bias = torch.randn(257, device="cuda", generator=g)[1:]         # data_ptr % 16 == 4
    assert bias.data_ptr() % 16 != 0
    got = fused_ops.linear_k128(x, w, bias, relu=True)
    want = (x.double() @ w.double().t() + bias.double()).relu()
    torch.testing.assert_close(got.double(), want, rtol=0, atol=2e-5)


def test_linear_k128_fp32_mfma_variant_still_matches(tmp_path):
    """PCT_LIN_KERNEL=f32 selects the fp32-MFMA kernel (kept for A/B); the switch is read once per process."""
    import subprocess, sys, os
    code = (
        "import torch, sys; sys.path.insert(0, %r)\n"
        "from pctrans_amd import fused_ops\n"
        "torch.manual_seed(0)\n"
        "x = torch.randn(5000, 128, device='cuda'); lin = torch.nn.Linear(128, 384).cuda()\n"
        "res = torch.randn(5000, 128, device='cuda'); norm = torch.nn.LayerNorm(128).cuda(); l2 = torch.nn.Linear(128, 128).cuda()\n"
        "with torch.no_grad():\n"
        "    got = fused_ops.linear_k128(x, lin.weight, lin.bias, relu=True)\n"
        "    want = (x.double() @ lin.weight.double().t() + lin.bias.double()).relu()\n"
        "    assert (got.double() - want).abs().max().item() < 2e-5\n"
        "    got = fused_ops.linear_add_layer_norm(x, l2, res, norm)\n"
        "    want = torch.nn.functional.layer_norm(res.double() + x.double() @ l2.weight.double().t() + l2.bias.double(), (128,), norm.weight.double(), norm.bias.double(), norm.eps)\n"
        "    assert (got.double() - want).abs().max().item() < 2e-5\n"
        "print('ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, PCT_LIN_KERNEL="f32")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("S,shared", [(5000, False), (5376, True), (7481, True), (33, True)])
def test_linear_k128_adds_the_positional_operand_on_the_fly(S, shared):
    """x_add = the encoder's positional term: per image, or one [1, S, 128] tensor shared by the batch (the kernel
    wraps its row index; S need not be a multiple of the 32-row tile)."""
    x = torch.randn(3, S, 128, device="cuda")
    pos = torch.randn(1 if shared else 3, S, 128, device="cuda")
    lin = torch.nn.Linear(128, 256).cuda()
    with torch.no_grad():
        got = fused_ops.linear_k128(x, lin.weight, lin.bias, x_add=pos)
        got_expanded = fused_ops.linear_k128(x, lin.weight, lin.bias, x_add=pos.expand(3, -1, -1))
        want = ((x.double() + pos.double()) @ lin.weight.double().t() + lin.bias.double())
    torch.testing.assert_close(got.double(), want, rtol=0, atol=3e-5)
    torch.testing.assert_close(got_expanded, got, rtol=0, atol=0)


def test_linear_k128_strided_rows_and_view_shapes():
    x_full = torch.randn(3, 1000, 256, device="cuda")
    x = x_full[..., :128]                                   # row stride 256, still 16-byte aligned
    w = torch.randn(128, 128, device="cuda") * 0.1
    got = fused_ops.linear_k128(x, w)
    assert got.shape == (3, 1000, 128)
    torch.testing.assert_close(got, torch.nn.functional.linear(x, w), rtol=0, atol=2e-5)


@pytest.mark.parametrize("rows", [21760, 4099])
def test_linear_add_layer_norm_equals_unfused(rows):
    torch.manual_seed(rows)
    lin = torch.nn.Linear(128, 128).cuda()
    norm = torch.nn.LayerNorm(128).cuda()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
        x = torch.randn(2, rows, 128, device="cuda")
        res = torch.randn(2, rows, 128, device="cuda") * 3 + 1.5
        got = fused_ops.linear_add_layer_norm(x, lin, res, norm)
        want = torch.nn.functional.layer_norm((res.double() + x.double() @ lin.weight.double().t() + lin.bias.double()),
                                              (128,), norm.weight.double(), norm.bias.double(), norm.eps)
    assert got.shape == res.shape
    torch.testing.assert_close(got.double(), want, rtol=0, atol=2e-5)


@pytest.mark.parametrize("rows,k", [(21760, 1024), (4099, 1024), (130, 256), (128, 32), (7, 2048)])
def test_linear_layer_norm_any_k_matches_fp64(rows, k):
    """Encoder FFN tail: norm2(src + linear2(h)) as one tiled GEMM on split-bf16 operands (linear_ln_split.hip),
    ragged last tile, several K; against fp64 at the fp32 noise level of a K-term dot product + LayerNorm."""
    torch.manual_seed(rows + k)
    lin = torch.nn.Linear(k, 128).cuda()
    norm = torch.nn.LayerNorm(128).cuda()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
        x = torch.randn(2, rows, k, device="cuda").relu_()
        res = torch.randn(2, rows, 128, device="cuda") * 3 + 1.5
        assert fused_ops.linear_layer_norm_supported(x, lin, res, norm)
        got = fused_ops.linear_layer_norm(x, lin, res, norm)
        want = torch.nn.functional.layer_norm(res.double() + x.double() @ lin.weight.double().t() + lin.bias.double(),
                                              (128,), norm.weight.double(), norm.bias.double(), norm.eps)
        lib = fused_ops.add_layer_norm(res, lin(x), norm)
    assert got.shape == res.shape
    torch.testing.assert_close(got.double(), want, rtol=0, atol=2e-5)
    assert (got.double() - want).abs().max() <= 2.0 * (lib.double() - want).abs().max() + 1e-6


def test_linear_layer_norm_pre_norm_sum_keeps_fp32_accuracy_over_the_exponent_range():
    """The product inside the fused kernel, isolated: gamma = 1, beta = 0 and a residual that dominates the row, so
    that LayerNorm is (nearly) affine in the product; operands scaled from 1e-6 to 1e6 per column of x."""
    torch.manual_seed(3)
    k, rows = 1024, 4096
    lin = torch.nn.Linear(k, 128, bias=False).cuda()
    norm = torch.nn.LayerNorm(128).cuda()
    with torch.no_grad():
        x = torch.randn(rows, k, device="cuda") * torch.logspace(-6, 6, k, device="cuda")[torch.randperm(k, device="cuda")]
        lin.weight.mul_(1.0 / torch.logspace(-6, 6, k, device="cuda")[torch.randperm(k, device="cuda")])
        res = torch.zeros(rows, 128, device="cuda")
        got = fused_ops.linear_layer_norm(x, lin, res, norm)
        prod = x.double() @ lin.weight.double().t()
        want = torch.nn.functional.layer_norm(prod, (128,), None, None, norm.eps)
        lib = torch.nn.functional.layer_norm(lin(x), (128,), None, None, norm.eps)
    err, lib_err = (got.double() - want).abs().max().item(), (lib.double() - want).abs().max().item()
    assert err <= 2.0 * lib_err + 1e-6, (err, lib_err)


def test_linear_layer_norm_falls_back_when_unsupported():
    lin = torch.nn.Linear(100, 128).cuda()              # K % 32 != 0
    norm = torch.nn.LayerNorm(128).cuda()
    x, res = torch.randn(50, 100, device="cuda"), torch.randn(50, 128, device="cuda")
    with torch.no_grad():
        assert not fused_ops.linear_layer_norm_supported(x, lin, res, norm)
        got = fused_ops.linear_layer_norm(x, lin, res, norm)
        torch.testing.assert_close(got, norm(res + lin(x)), rtol=1e-5, atol=1e-5)


def test_cached_linear_equals_autocast_linear_and_follows_weight_updates():
    from pctrans_amd.layers import CachedLinear
    torch.manual_seed(3)
    ref = nn.Linear(128, 64).cuda()
    lin = CachedLinear(128, 64).cuda()
    lin.load_state_dict(ref.state_dict())
    assert sorted(lin.state_dict()) == ["bias", "weight"]
    x = torch.randn(7, 100, 128, device="cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        assert torch.equal(lin(x), ref(x))
        assert torch.equal(lin(x, relu=True), torch.relu(ref(x)))
        ref.weight.mul_(1.5)
        lin.weight.mul_(1.5)                          # in-place update bumps the version -> the bf16 copy is rebuilt
        torch.clear_autocast_cache()                  # (autocast itself would keep serving ref's stale bf16 weight)
        assert torch.equal(lin(x), ref(x))
        lin.load_state_dict({k: v * 0.5 for k, v in ref.state_dict().items()})
        ref.load_state_dict(lin.state_dict())
        torch.clear_autocast_cache()
        assert torch.equal(lin(x), ref(x))
    with torch.no_grad():                               # outside autocast: plain fp32 nn.Linear
        assert torch.equal(lin(x), ref(x))
    with torch.autocast("cuda", dtype=torch.bfloat16):  # autograd needed: falls back to the differentiable path
        y = lin(x)
        assert y.requires_grad
        y.float().sum().backward()
        assert lin.weight.grad is not None


@pytest.mark.parametrize("cin,cout,hw,bias", [(256, 128, (64, 64), True), (2048, 128, (17, 22), True), (128, 16, (130, 174), False)])
def test_pointwise_conv2d_is_a_batched_gemm_with_conv_semantics(cin, cout, hw, bias):
    from pctrans_amd.layers import Conv2d
    torch.manual_seed(cin)
    conv = Conv2d(cin, cout, kernel_size=1, bias=bias, norm=nn.GroupNorm(8, cout), activation=torch.relu).cuda()
    x = torch.randn(3, cin, *hw, device="cuda")
    with torch.no_grad():
        got = conv(x)
        want = torch.relu(conv.norm(torch.nn.functional.conv2d(x.double(), conv.weight.double(),
                                                              None if conv.bias is None else conv.bias.double()).float()))
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)
        # channels-last input takes the library convolution; 3x3 kernels are untouched
        cl = x.contiguous(memory_format=torch.channels_last)
        torch.testing.assert_close(conv(cl), want, rtol=1e-4, atol=1e-4)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert conv(x).dtype == torch.float32 or conv(x).dtype == torch.bfloat16
            y16 = Conv2d(cin, cout, kernel_size=1, bias=bias).cuda()(x)
            assert y16.dtype == torch.bfloat16          # same autocast policy as conv2d


@pytest.mark.parametrize("hw", [(16, 16), (17, 22), (128, 128), (65, 87)])
def test_groupnorm_flatten_into_equals_groupnorm_then_transpose(hw):
    torch.manual_seed(hw[0])
    gn = nn.GroupNorm(32, 128).cuda()
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)
        x = torch.randn(3, 128, *hw, device="cuda") * 2 + 0.3
        S = hw[0] * hw[1] + 11
        out = torch.full((3, S, 128), float("nan"), device="cuda")
        assert fused_ops.groupnorm_flatten_supported(x, gn)
        fused_ops.groupnorm_flatten_into(x, gn, out, 7)
        want = gn(x.double().float()).flatten(2).transpose(1, 2)
    torch.testing.assert_close(out[:, 7:7 + hw[0] * hw[1]], want, rtol=0, atol=2e-5)
    assert torch.isnan(out[:, :7]).all() and torch.isnan(out[:, 7 + hw[0] * hw[1]:]).all()    # nothing else touched


@pytest.mark.parametrize("tag,masked", [("ca", True), ("ca", False), ("sa", False)])
def test_mfma_attention_kernel_against_the_reference_class_fixture(golden, tag, masked):
    """The bf16 MFMA attention kernel (C ABI) followed by out_proj, against the fp32 output of the reference's own
    MultiheadAttention (tests/golden/make_golden_decoder.py); tolerance = bf16 operand rounding."""
    import numpy as np
    g = golden("dec_attention_" + tag)
    heads = int(g["heads"])
    q, k, v = (torch.from_numpy(g[n]).cuda() for n in ("q", "k", "v"))
    if q.shape[2] // heads not in (16, 32) or v.shape[2] // heads != 16:
        pytest.skip("head geometry outside the MFMA kernel")
    L, N, S = q.shape[0], q.shape[1], k.shape[0]
    mask = None
    if masked:
        mask = torch.from_numpy(g["bool_mask"]).view(N, heads, L, S)[:, :1].contiguous().cuda()
    core = fused_ops.masked_attention(q.bfloat16(), k.bfloat16(), v.bfloat16(), heads, mask).float()
    out = core @ torch.from_numpy(g["out_w"]).cuda().t() + torch.from_numpy(g["out_b"]).cuda()
    want = g["out_bool" if masked else "out_plain"]
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=3e-2 * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("Q,counts", [(100, [17, 0, 50, 100, 1]), (300, [120, 299, 3]), (12, [12, 5]), (1000, [400])])
def test_device_lsap_equals_scipy(Q, counts):
    """Hungarian matching on the device (csrc/lsap.hip) against scipy.optimize.linear_sum_assignment, the solver the
    reference's matcher calls: same assignment on generic (tie-free) costs, hence the same total cost."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    rng = np.random.RandomState(Q + len(counts))
    gmax = max(1, max(counts))
    cost = (rng.standard_normal((len(counts), Q, gmax)) * 3 + rng.random_sample((len(counts), Q, 1))).astype(np.float32)
    rows, status = fused_ops.lsap(torch.from_numpy(cost).cuda(), torch.tensor(counts, dtype=torch.int32))
    rows, status = rows.cpu().numpy(), status.cpu().numpy()
    assert not status.any()
    for b, g in enumerate(counts):
        assert (rows[b, g:] == -1).all()
        if g == 0:
            continue
        ri, ci = linear_sum_assignment(cost[b, :, :g].astype(np.float64))
        want = np.empty(g, dtype=np.int64)
        want[ci] = ri
        np.testing.assert_array_equal(rows[b, :g], want)


def test_device_lsap_ties_and_infeasible_costs():
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    # heavy ties (integer costs): any optimal assignment is acceptable -> compare the total cost, and validity
    rng = np.random.RandomState(3)
    cost = rng.randint(0, 4, size=(2, 40, 25)).astype(np.float32)
    rows, status = fused_ops.lsap(torch.from_numpy(cost).cuda(), torch.tensor([25, 25], dtype=torch.int32))
    rows = rows.cpu().numpy()
    for b in range(2):
        assert len(set(rows[b].tolist())) == 25 and rows[b].min() >= 0
        ri, ci = linear_sum_assignment(cost[b].astype(np.float64))
        assert cost[b][rows[b], np.arange(25)].sum() == cost[b][ri, ci].sum()
    bad = torch.randn(1, 10, 4).cuda()
    bad[0, :, 2] = float("nan")
    _, status = fused_ops.lsap(bad, torch.tensor([4], dtype=torch.int32))
    assert int(status[0]) == 1


def test_matcher_on_device_equals_the_scipy_path():
    """Point_HungarianMatcher with device assignment vs the reference's host path (scipy) on the same sampled points."""
    from pctrans_amd.loss.matcher import Point_HungarianMatcher
    torch.manual_seed(0)
    pred = torch.randn(3, 50, 32, 32, device="cuda") * 2
    targets = [{"masks": (torch.rand(g, 64, 64, device="cuda") > 0.5).float()} for g in (7, 0, 20)]
    m = Point_HungarianMatcher(cost_mask=5.0, cost_dice=5.0, num_points=256)
    torch.manual_seed(1)
    on_dev = m({"pred_masks": pred}, targets)
    m.check()
    m.device_lsap = False
    torch.manual_seed(1)                              # same random point coordinates
    on_host = m({"pred_masks": pred}, targets)
    for (i, j), (ih, jh) in zip(on_dev, on_host):
        assert i.is_cuda and i.dtype == torch.int64
        assert torch.equal(i.cpu(), ih) and torch.equal(j.cpu(), jh)


def test_matcher_forward_many_is_one_launch_with_the_per_call_results():
    """Point_HungarianMatcher.forward_many: ten predictions' assignment problems in ONE device launch = ten calls of forward
    with the same random stream (ragged target counts, an image without targets), and the host path when the device solver
    is switched off."""
    from pctrans_amd import fused_ops
    from pctrans_amd.loss.matcher import Point_HungarianMatcher
    torch.manual_seed(0)
    preds = [torch.randn(3, 50, 32, 32, device="cuda") * 2 for _ in range(10)]
    targets = [{"masks": (torch.rand(g, 64, 64, device="cuda") > 0.5).float()} for g in (7, 0, 20)]
    m = Point_HungarianMatcher(cost_mask=5.0, cost_dice=5.0, num_points=256)
    torch.manual_seed(1)
    ref = [m({"pred_masks": p}, targets) for p in preds]
    calls = []
    real = fused_ops.lsap
    fused_ops.lsap = lambda c, n: (calls.append(tuple(c.shape)), real(c, n))[1]
    try:
        torch.manual_seed(1)
        many = m.forward_many([{"pred_masks": p} for p in preds], targets)
    finally:
        fused_ops.lsap = real
    m.check()
    assert calls == [(30, 50, 20)]
    for a, b in zip(many, ref):
        for (i, j), (k, l) in zip(a, b):
            assert i.is_cuda and i.dtype == torch.int64 and torch.equal(i, k) and torch.equal(j, l)
    m.device_lsap = False
    torch.manual_seed(1)
    host = m.forward_many([{"pred_masks": p} for p in preds], targets)
    for a, b in zip(host, ref):
        for (i, j), (k, l) in zip(a, b):
            assert torch.equal(i, k.cpu()) and torch.equal(j, l.cpu())


def test_device_lsap_flags_what_scipy_rejects():
    """scipy.optimize.linear_sum_assignment raises on a NaN or -inf entry anywhere in the matrix and on more columns than
    rows in the transposed call the reference makes (matcher.py:154-165); the device solver reports those problems
    through status instead of returning an assignment (and never indexes past its LDS state)."""
    cost = torch.randn(4, 12, 6).cuda()
    cost[1, 3, 2] = float("nan")                 # one NaN among valid entries
    cost[2, 0, 5] = float("-inf")
    cost[3, :, 1] = float("inf")                 # a target nobody can take: infeasible, as before
    rows, status = fused_ops.lsap(cost, torch.tensor([6, 6, 6, 6], dtype=torch.int32))
    assert status.cpu().tolist() == [0, 1, 1, 1]
    assert (rows[0] >= 0).all() and len(set(rows[0].cpu().tolist())) == 6
    # NaN / -inf in columns past num_target are not part of the problem
    cost2 = torch.randn(1, 12, 6).cuda()
    cost2[0, :, 4:] = float("nan")
    rows2, status2 = fused_ops.lsap(cost2, torch.tensor([4], dtype=torch.int32))
    assert int(status2[0]) == 0 and (rows2[0, :4] >= 0).all() and (rows2[0, 4:] == -1).all()
    # more targets than queries, more targets than columns: flagged, nothing written out of bounds
    _, status3 = fused_ops.lsap(torch.randn(2, 5, 8).cuda(), torch.tensor([6, 9], dtype=torch.int32))
    assert status3.cpu().tolist() == [1, 1]


def test_cross_attention_trains_its_projections_when_the_inputs_are_detached():
    """bf16 autocast, grad enabled, trainable ca_* weights but detached inputs (a frozen pixel decoder / query
    embedding): the layer must take the differentiable path -- the forward-only MFMA attention kernel would leave the
    projections without gradients, silently."""
    from pctrans_amd.transformer_decoder.mask2former_transformer_decoder import CrossAttentionLayer
    torch.manual_seed(0)
    ca = CrossAttentionLayer(128, 8).cuda()
    Q, N, HW = 10, 2, 64
    tgt, mem, pos = torch.randn(Q, N, 128).cuda(), torch.randn(HW, N, 128).cuda(), torch.randn(HW, N, 128).cuda()
    qpos, qsine = torch.randn(Q, N, 128).cuda(), torch.randn(Q, N, 256).cuda()
    mask = torch.zeros(N, 1, Q, HW, dtype=torch.bool).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = ca(tgt, mem, memory_mask=mask, pos=pos, query_pos=qpos, query_sine_embed=qsine, is_first=True)
    out.float().square().mean().backward()
    for name in ("ca_qcontent_proj", "ca_kcontent_proj", "ca_kpos_proj", "ca_v_proj", "ca_qpos_sine_proj", "ca_qpos_proj"):
        g = getattr(ca, name).weight.grad
        assert g is not None and float(g.abs().max()) > 0, name
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):          # and forward-only still fuses
        out2 = ca(tgt, mem, memory_mask=mask, pos=pos, query_pos=qpos, query_sine_embed=qsine, is_first=True)
    assert float((out2.float() - out.float()).abs().max()) < 0.1


def test_loss_and_postprocessing_definitions_on_the_device(golden):
    """SURVEY 8 f-3 / f-4 on device tensors: dice_loss, sigmoid_ce_loss, calculate_uncertainty
    (maskformer_criterion.py:23-115), batch_dice_loss, batch_sigmoid_ce_loss (matcher.py:15-62), dice_for, mask_post,
    comput_mmi (arch/maskformer.py:349-431) against the vectors the reference's own functions produced."""
    import numpy as np
    from pctrans_amd.arch import maskformer as mfm
    from pctrans_amd.loss import maskformer_criterion as crit
    from pctrans_amd.loss import matcher
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    g = golden("loss_functions")
    logits, tgt, tgt2, nm = t(g["logits"]), t(g["targets"]), t(g["targets2"]), float(g["num_masks"])
    np.testing.assert_allclose(crit.dice_loss(logits, tgt, nm).cpu().numpy(), g["dice_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(crit.sigmoid_ce_loss(logits, tgt, nm).cpu().numpy(), g["sigmoid_ce_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(crit.calculate_uncertainty(logits[:, None, :]).cpu().numpy(), g["uncertainty"], rtol=0, atol=0)
    np.testing.assert_allclose(matcher.batch_dice_loss(logits, tgt2).cpu().numpy(), g["batch_dice"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(matcher.batch_sigmoid_ce_loss(logits, tgt2).cpu().numpy(), g["batch_ce"], rtol=1e-5, atol=2e-5)
    g = golden("arch_mask_post")
    inst = t(g["inst_masks"])
    np.testing.assert_allclose(mfm.dice_for(inst).cpu().numpy(), g["dice"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mfm.mask_post(inst, thres1=0.5, thres2=0.6, bd_flag=False).cpu().numpy(), g["post_soft"],
                               rtol=0, atol=1e-6)
    np.testing.assert_array_equal(mfm.mask_post(inst, thres1=0.5, thres2=0.6, bd_flag=True).cpu().numpy(), g["post_hard"])
    np.testing.assert_allclose(mfm.mask_post(inst, thres1=0.15, thres2=0.25).cpu().numpy(), g["post_bbbc"], rtol=0, atol=1e-6)
    for (a, b, c), want in zip(g["mmi_in"], g["mmi_out"]):
        got = float(mfm.comput_mmi(torch.tensor(float(a)).cuda(), torch.tensor(float(b)).cuda(), torch.tensor(float(c)).cuda()))
        assert abs(got - float(want)) <= 1e-6 * max(1.0, abs(float(want)))
    g = golden("dec_query_contrast")            # query-contrast selection (dec.py:800-900) on device tensors
    from pctrans_amd.transformer_decoder import query_contrast as qc
    pos_indices = [(t(g["pos_src_%d" % b]), t(g["pos_tgt_%d" % b])) for b in range(2)]
    items_q = qc.select_pos_neg_query(t(g["query"]), t(g["emb_dist"]), pos_indices)
    items_m = qc.select_pos_neg_mask(t(g["masks"]), t(g["emb_dist"]), pos_indices)
    assert len(items_q) == int(g["n_items_q"]) and len(items_m) == int(g["n_items_m"])
    # (negatives come in set-iteration order in the reference, ascending here: compared per group up to order, as the loss --
    # a logsumexp over each group -- does; see tests/test_decoder_golden.py)
    for prefix, items in (("q", items_q), ("m", items_m)):
        for i, it in enumerate(items):
            want_c, want_l = g["%s%d_contrast" % (prefix, i)].ravel(), g["%s%d_label" % (prefix, i)]
            got_c, got_l = it["contrast"].cpu().numpy().ravel(), it["label"].cpu().numpy()
            np.testing.assert_array_equal(got_l, want_l)
            for lab in (0, 1):
                np.testing.assert_allclose(np.sort(got_c[got_l == lab]), np.sort(want_c[want_l == lab]), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_token_row_conv_never_serves_another_modules_weights():
    """The 1x1 `mask_head` projection at batch <= 4 keeps padded copies of its weights.  They live on the module, so a
    second module (evaluation sweeps, a new checkpoint) built after the first one was freed -- CPython may hand it the
    same id(), and init bumps the version counters identically -- gets its own; in-place writes and replaced parameters
    are followed."""
    import gc
    from pctrans_amd import fused_ops
    from pctrans_amd.layers import Conv2d
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(11)
    rows = torch.randn(2, 64 * 64, 128, generator=g).to(dev)
    x = rows.view(2, 64, 64, 128).permute(0, 3, 1, 2)                      # the encoder's token rows seen as NCHW
    outs = []
    for seed in (1, 2, 3, 4, 5):
        torch.manual_seed(seed)
        conv = Conv2d(128, 16, kernel_size=1).to(dev)
        with torch.no_grad():
            assert fused_ops.conv1x1_from_token_rows_supported(x, conv)
            got = fused_ops.conv1x1_from_token_rows(x, conv)
            want = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double())
        assert float((got.double() - want).abs().max()) < 2e-5, "seed %d: stale weights?" % seed
        outs.append(got)
        del conv
        gc.collect()
    assert not torch.equal(outs[0], outs[1])
    # in-place update and parameter replacement on ONE module
    torch.manual_seed(9)
    conv = Conv2d(128, 16, kernel_size=1).to(dev)
    with torch.no_grad():
        a = fused_ops.conv1x1_from_token_rows(x, conv)
        conv.weight.mul_(2.0)
        b = fused_ops.conv1x1_from_token_rows(x, conv)
        conv.weight = torch.nn.Parameter(conv.weight.detach().clone() * 0.5)
        c = fused_ops.conv1x1_from_token_rows(x, conv)
        conv.weight.data.mul_(3.0)            # a .data write does not bump the version: documented limit of every cache here
    bias = conv.bias.view(1, -1, 1, 1)
    assert float(((b - bias) - 2.0 * (a - bias)).abs().max()) < 1e-4
    assert float((c - a).abs().max()) < 1e-4


@pytest.mark.parametrize("N,K,H,W,bias", [(2, 256, 128, 128, True), (3, 512, 68, 64, True), (1, 2048, 16, 16, True),
                                          (2, 32, 8, 16, False), (5, 1024, 32, 32, True), (1, 16, 16, 8, True)])
def test_conv1x1_nchw_is_as_accurate_as_fp32(N, K, H, W, bias):
    """The pixel decoder's 1x1 input projections on the split-bf16 MFMA kernel (csrc/conv1x1_split.hip) against fp64:
    error no larger than the fp32 convolution's own (pixel_decoder/msdeformattn.py:213-226)."""
    from pctrans_amd.layers import Conv2d
    torch.manual_seed(K + H)
    conv = Conv2d(K, 128, kernel_size=1, bias=bias).cuda()
    with torch.no_grad():
        conv.weight.mul_(3.0)
    x = torch.randn(N, K, H, W, device="cuda") * torch.logspace(-2, 2, K, device="cuda").view(1, K, 1, 1)
    with torch.no_grad():
        assert fused_ops.conv1x1_nchw_supported(x, conv)
        got = fused_ops.conv1x1_nchw(x, conv)
        ref64 = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double() if bias else None)
        ref32 = conv(x)
    scale = float(ref64.abs().max())
    err = float((got.double() - ref64).abs().max()) / scale
    err32 = float((ref32.double() - ref64).abs().max()) / scale
    assert got.shape == ref32.shape and torch.isfinite(got).all()
    assert err <= max(2.0 * err32, 2e-7), (err, err32)


def test_conv1x1_nchw_falls_back_where_the_kernel_does_not_apply():
    from pctrans_amd.layers import Conv2d
    conv = Conv2d(64, 128, kernel_size=1).cuda()
    x = torch.randn(1, 64, 10, 10, device="cuda")                    # 100 pixels: not a multiple of 128
    assert not fused_ops.conv1x1_nchw_supported(x, conv)
    with torch.no_grad():
        assert torch.equal(fused_ops.conv1x1_nchw(x, conv), conv(x))
    conv3 = Conv2d(64, 128, kernel_size=3, padding=1).cuda()
    assert not fused_ops.conv1x1_nchw_supported(torch.randn(1, 64, 16, 8, device="cuda"), conv3)
    assert not fused_ops.conv1x1_nchw_supported(torch.randn(1, 64, 16, 8, device="cuda").requires_grad_(), conv)


@pytest.mark.parametrize("N,K,H,W", [(2, 256, 32, 32), (3, 512, 16, 24), (1, 2048, 16, 16), (2, 32, 8, 16)])
def test_conv1x1_groupnorm_tokens_equals_the_module_chain(N, K, H, W):
    """Input projection in one entry (csrc/conv1x1_split.hip, token epilogue + per-tile GroupNorm records + in-place
    normalisation) against nn.Sequential(Conv2d(K, 128, 1), GroupNorm(32, 128)) + flatten(2).transpose(1, 2) in fp64, written at
    a row offset of a larger token buffer whose other rows must stay untouched (pixel_decoder/msdeformattn.py:213-226, 75-83)."""
    from pctrans_amd.layers import Conv2d
    torch.manual_seed(N * K + W)
    conv = Conv2d(K, 128, kernel_size=1).cuda()
    gn = torch.nn.GroupNorm(32, 128).cuda()
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)
        conv.bias.uniform_(-2.0, 2.0)                           # group means far from zero: the variance must not cancel
    x = torch.randn(N, K, H, W, device="cuda") + 0.5
    S, off = H * W + 40, 24
    out = torch.full((N, S, 128), 7.0, device="cuda")
    with torch.no_grad():
        assert fused_ops.conv1x1_groupnorm_tokens_supported(x, conv, gn)
        fused_ops.conv1x1_groupnorm_tokens_into(x, conv, gn, out, off)
        ref = torch.nn.functional.group_norm(
            torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double()), 32, gn.weight.double(),
            gn.bias.double(), gn.eps).flatten(2).transpose(1, 2)
    assert float((out[:, :off] - 7.0).abs().max()) == 0.0 and float((out[:, off + H * W:] - 7.0).abs().max()) == 0.0
    err = float((out[:, off:off + H * W].double() - ref).abs().max())
    assert err <= 2e-5, err


@pytest.mark.parametrize("rows,F", [(128, 1024), (4000, 1024), (37, 64), (257, 2048), (129, 32), (1000, 96), (70001, 1024)])
def test_fused_ffn_is_as_accurate_as_fp32(rows, F):
    """The encoder layer's FFN in one kernel (csrc/ffn_fused_split.hip: the hidden activations stay in registers between the two
    split-bf16 products) against fp64 and against the fp32 module chain norm2(x + linear2(relu(linear1(x))))
    (pixel_decoder/msdeformattn.py:122-131): error no larger than twice the fp32 chain's own."""
    torch.manual_seed(rows + F)
    l1, l2 = torch.nn.Linear(128, F).cuda(), torch.nn.Linear(F, 128).cuda()
    norm = torch.nn.LayerNorm(128).cuda()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
        l1.weight.mul_(2.0)
    x = torch.randn(rows, 128, device="cuda") * torch.logspace(-1, 1, 128, device="cuda")
    with torch.no_grad():
        assert fused_ops.ffn_layer_norm_supported(x, l1, l2, norm)
        got = fused_ops.ffn_layer_norm(x, l1, l2, norm)
        xd = x.double()
        hid = torch.relu(xd @ l1.weight.double().t() + l1.bias.double())
        ref = torch.nn.functional.layer_norm(xd + hid @ l2.weight.double().t() + l2.bias.double(), (128,), norm.weight.double(),
                                             norm.bias.double(), norm.eps)
        chain = norm(x + l2(torch.relu(l1(x))))
    err = float((got.double() - ref).abs().max())
    err32 = float((chain.double() - ref).abs().max())
    assert torch.isfinite(got).all() and got.shape == x.shape
    assert err <= max(2.0 * err32, 5e-6), (err, err32)


def test_fused_ffn_strided_rows_no_bias_and_repeat_calls():
    """Rows that are a column slice of a wider tensor (row stride 160 floats), linear2 without a bias, and the same workspace
    reused by calls with different weights (the weight images are refilled per call); the persistent loop over several tiles
    per workgroup against the two-kernel path."""
    torch.manual_seed(5)
    norm = torch.nn.LayerNorm(128).cuda()
    wide = torch.randn(40000, 160, device="cuda")
    x = wide[:, 16:144]
    assert x.stride(0) == 160 and x.data_ptr() % 16 == 0
    with torch.no_grad():
        for seed in (1, 2):
            torch.manual_seed(seed)
            l1, l2 = torch.nn.Linear(128, 256).cuda(), torch.nn.Linear(256, 128, bias=seed == 1).cuda()
            assert fused_ops.ffn_layer_norm_supported(x, l1, l2, norm)
            got = fused_ops.ffn_layer_norm(x, l1, l2, norm)
            two = fused_ops.linear_layer_norm(fused_ops.linear(x.contiguous(), l1, relu=True), l2, x.contiguous(), norm)
            ref = torch.nn.functional.layer_norm(
                x.double() + torch.relu(x.double() @ l1.weight.double().t() + l1.bias.double()) @ l2.weight.double().t()
                + (l2.bias.double() if l2.bias is not None else 0.0), (128,), norm.weight.double(), norm.bias.double(), norm.eps)
            assert float((got.double() - ref).abs().max()) <= 5e-6
            assert float((got - two).abs().max()) <= 5e-6
    assert float((wide[:, :16] - wide[:, :16]).abs().max()) == 0.0


def test_fused_ffn_refuses_what_it_does_not_cover():
    l1, l2, norm = torch.nn.Linear(128, 1000).cuda(), torch.nn.Linear(1000, 128).cuda(), torch.nn.LayerNorm(128).cuda()
    x = torch.randn(64, 128, device="cuda")
    with torch.no_grad():
        assert not fused_ops.ffn_layer_norm_supported(x, l1, l2, norm)                      # hidden % 32
        assert not fused_ops.ffn_layer_norm_supported(x.half(), l1, l2, norm)
    l1b = torch.nn.Linear(128, 1024).cuda()
    assert not fused_ops.ffn_layer_norm_supported(x.requires_grad_(True), l1b, torch.nn.Linear(1024, 128).cuda(), norm)   # autograd
