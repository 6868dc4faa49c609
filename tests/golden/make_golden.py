#!/usr/bin/env python3
"""Generate golden vectors from the reference's own importable pieces.  RUN IN THE DEV CONTAINER ONLY.

This is the only place that touches /root/reference (which does not exist on the GPU box):
it imports, from the reference checkout, the pure-PyTorch pieces that run on CPU
    * ms_deform_attn_core_pytorch   (.../pixel_decoder/ops/functions/ms_deform_attn_func.py:52-72)
    * MSDeformAttn (nn.Module)      (.../pixel_decoder/ops/modules/ms_deform_attn.py:34-125)
    * PositionEmbeddingSine         (.../transformer_decoder/position_encoding.py:12-52)
feeds them seeded inputs and stores inputs + outputs as small .npz fixtures next to this file.
Nothing of the reference's source is copied; only data is written.

    python tests/golden/make_golden.py [--ref /root/reference]
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OPS = "connectomics/model/maskformer_block/pixel_decoder/ops"
TDEC = "connectomics/model/maskformer_block/transformer_decoder"


def load_reference(ref_root):
    """Import the three torch-only reference pieces without the (absent) CUDA extension."""
    # the import guard at ms_deform_attn_func.py:21-29 wants a module of this name to exist
    sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
    pkg = types.ModuleType("refops")
    pkg.__path__ = [os.path.join(ref_root, OPS)]
    sys.modules["refops"] = pkg
    import refops.functions.ms_deform_attn_func as func  # noqa: E402
    import refops.modules.ms_deform_attn as mod  # noqa: E402

    spec = importlib.util.spec_from_file_location(
        "ref_position_encoding", os.path.join(ref_root, TDEC, "position_encoding.py"))
    pe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pe)
    return func, mod, pe


def starts_of(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def make_inputs(seed, N, M, D, Lq, P, shapes, dtype, lo=0.0, hi=1.0, value_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor(shapes, dtype=torch.long)
    L = shapes.shape[0]
    S = int(shapes.prod(1).sum())
    value = (torch.randn(N, S, M, D, generator=g, dtype=torch.float64) * value_scale).to(dtype)
    loc = (torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=torch.float64) * (hi - lo) + lo).to(dtype)
    w = torch.rand(N, Lq, M, L, P, generator=g, dtype=torch.float64) + 1e-5
    w = (w / w.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype)
    return value, shapes, starts_of(shapes), loc, w


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def fwd_case(func, name, with_grad=False, **kw):
    value, shapes, starts, loc, w = make_inputs(**kw)
    if with_grad:
        value.requires_grad_(True)
        loc.requires_grad_(True)
        w.requires_grad_(True)
    out = func.ms_deform_attn_core_pytorch(value, shapes, loc, w)
    arrs = dict(value=value, shapes=shapes, starts=starts, loc=loc, attn=w, out=out)
    if with_grad:
        gg = torch.Generator().manual_seed(kw["seed"] + 1000)
        grad_out = torch.randn(out.shape, generator=gg, dtype=torch.float64).to(out.dtype)
        gv, gl, ga = torch.autograd.grad(out, (value, loc, w), grad_out)
        arrs.update(grad_out=grad_out, grad_value=gv, grad_loc=gl, grad_attn=ga)
    save(name, **arrs)


def kat_test_py(func):
    """The reference's own test recipe: .../ops/test.py:24-59 (seed 3; 1st draw f64 check, 2nd draw f32 check)."""
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    arrs = {"shapes": shapes, "starts": starts_of(shapes)}
    for tag in ("d1", "d2"):
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        w = torch.rand(N, Lq, M, L, P) + 1e-5
        w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
        arrs[tag + "_value"], arrs[tag + "_loc"], arrs[tag + "_attn"] = value, loc, w
        arrs[tag + "_out_f32"] = func.ms_deform_attn_core_pytorch(value, shapes, loc, w)
        arrs[tag + "_out_f64"] = func.ms_deform_attn_core_pytorch(value.double(), shapes, loc.double(), w.double())
    save("kat_test_py", **arrs)


def module_case(mod, name, seed, d_model, n_levels, n_heads, n_points, shapes, N, with_mask):
    """MSDeformAttn module (ops/modules/ms_deform_attn.py:82-125); its bare `except:` routes CPU tensors to the
    pure-PyTorch core, so this pins offsets/softmax/location math + the 4 Linears for a seeded state-dict."""
    torch.manual_seed(seed)
    m = mod.MSDeformAttn(d_model=d_model, n_levels=n_levels, n_heads=n_heads, n_points=n_points)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():  # leave the zero-weight init (every query samples the same offsets) behind
        m.sampling_offsets.weight.copy_(torch.randn(m.sampling_offsets.weight.shape, generator=g) * 0.05)
        m.attention_weights.weight.copy_(torch.randn(m.attention_weights.weight.shape, generator=g) * 0.2)
        m.attention_weights.bias.copy_(torch.randn(m.attention_weights.bias.shape, generator=g) * 0.2)
    shapes = torch.as_tensor(shapes, dtype=torch.long)
    S = int(shapes.prod(1).sum())
    query = torch.randn(N, S, d_model, generator=g)
    src = torch.randn(N, S, d_model, generator=g)
    ref = torch.rand(N, S, n_levels, 2, generator=g)
    mask = (torch.rand(N, S, generator=g) < 0.15) if with_mask else None
    with torch.no_grad():
        out = m(query, ref, src, shapes, starts_of(shapes), mask)
    arrs = {("sd." + k): v for k, v in m.state_dict().items()}
    arrs.update(query=query, ref=ref, src=src, shapes=shapes, starts=starts_of(shapes), out=out)
    if mask is not None:
        arrs["mask"] = mask
    save(name, **arrs)


def pe_case(pe):
    """PositionEmbeddingSine(N_steps, normalize=True) as the pixel decoder / decoder build it
    (msdeformattn.py:210-211, mask2former_transformer_decoder.py:341-342)."""
    enc = pe.PositionEmbeddingSine(64, normalize=True)
    arrs = {}
    for (h, w) in ((4, 4), (5, 7), (16, 16)):
        x = torch.zeros(2, 3, h, w)
        arrs["pe_%dx%d" % (h, w)] = enc(x)
    save("position_encoding_sine", **arrs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    func, mod, pe = load_reference(args.ref)
    torch.set_num_threads(4)

    kat_test_py(func)
    # power-of-two pyramid, PCTrans head geometry (M=8, D=16, P=4, L=3), query subset
    fwd_case(func, "fwd_pow2_L3_f32", seed=11, N=2, M=8, D=16, Lq=96, P=4,
             shapes=[(4, 4), (8, 8), (16, 16)], dtype=torch.float32)
    # BBBC-like non-square / non-power-of-two levels, locations spill outside [0,1] (edge + gate cases)
    fwd_case(func, "fwd_nonpow2_edges_f32", seed=12, N=1, M=8, D=16, Lq=100, P=4,
             shapes=[(5, 7), (9, 11), (17, 22)], dtype=torch.float32, lo=-0.25, hi=1.25)
    # 4 levels, 8 points, wider heads, batch 3
    fwd_case(func, "fwd_L4_P8_D32_f32", seed=13, N=3, M=4, D=32, Lq=40, P=8,
             shapes=[(3, 3), (5, 4), (8, 8), (12, 10)], dtype=torch.float32)
    # channel count that is not a multiple of 4 (scalar lane path)
    fwd_case(func, "fwd_oddD_f64", seed=14, N=2, M=3, D=5, Lq=17, P=3,
             shapes=[(6, 4), (3, 2)], dtype=torch.float64, lo=-0.1, hi=1.1, with_grad=True)
    # gradients (autograd through the reference function), f64 and f32
    fwd_case(func, "grad_small_f64", seed=15, N=2, M=2, D=4, Lq=9, P=2,
             shapes=[(6, 4), (3, 2)], dtype=torch.float64, lo=-0.1, hi=1.1, with_grad=True)
    fwd_case(func, "grad_head_geom_f32", seed=16, N=1, M=8, D=16, Lq=48, P=4,
             shapes=[(4, 4), (8, 8), (16, 16)], dtype=torch.float32, with_grad=True)

    module_case(mod, "module_L3_d128", seed=21, d_model=128, n_levels=3, n_heads=8, n_points=4,
                shapes=[(2, 3), (4, 6), (8, 12)], N=2, with_mask=False)
    module_case(mod, "module_L2_d64_mask", seed=22, d_model=64, n_levels=2, n_heads=4, n_points=2,
                shapes=[(3, 3), (6, 5)], N=1, with_mask=True)
    pe_case(pe)


if __name__ == "__main__":
    main()
