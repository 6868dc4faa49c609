"""GPU: small fused kernels against the torch expressions they replace (fp32, tolerance 1e-5 relative to scale)."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


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
