#!/usr/bin/env python3
"""Timing of the K = 128 fp32 projection kernel against the library path it replaces (development tool)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pctrans_amd import fused_ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rows = 64 * 21760
x = torch.randn(rows, 128, device="cuda")
res = torch.randn(rows, 128, device="cuda")
norm = torch.nn.LayerNorm(128).cuda()
with torch.no_grad():
    for n, relu in ((128, False), (256, False), (384, False), (1024, True)):
        lin = torch.nn.Linear(128, n).cuda()
        flops = 2.0 * rows * 128 * n
        t_lib = timed(lambda: (torch._addmm_activation(lin.bias, x, lin.weight.t(), use_gelu=False) if relu
                               else torch.nn.functional.linear(x, lin.weight, lin.bias)))
        t_own = timed(lambda: fused_ops.linear_k128(x, lin.weight, lin.bias, relu=relu))
        print("n=%4d relu=%d  library %.3f ms (%.0f TF/s)   linear_k128 %.3f ms (%.0f TF/s)" % (
            n, relu, t_lib, flops / t_lib / 1e9, t_own, flops / t_own / 1e9))
    lin = torch.nn.Linear(128, 128).cuda()
    t_lib = timed(lambda: fused_ops.add_layer_norm(res, torch.nn.functional.linear(x, lin.weight, lin.bias), norm))
    t_own = timed(lambda: fused_ops.linear_add_layer_norm(x, lin, res, norm))
    print("output_proj + residual + LayerNorm: library GEMM + add_layernorm %.3f ms   fused %.3f ms" % (t_lib, t_own))
    pos = torch.randn(rows, 128, device="cuda")
    for n in (128, 256):
        lin = torch.nn.Linear(128, n).cuda()
        t_sep = timed(lambda: fused_ops.linear_k128(x + pos, lin.weight, lin.bias))
        t_fused = timed(lambda: fused_ops.linear_k128(x, lin.weight, lin.bias, x_add=pos))
        print("n=%4d (x + pos): add kernel + linear_k128 %.3f ms   x_add operand %.3f ms" % (n, t_sep, t_fused))
    lin2 = torch.nn.Linear(1024, 128).cuda()
    hdn = torch.randn(rows, 1024, device="cuda").relu_()
    t_lib = timed(lambda: fused_ops.add_layer_norm(res, lin2(hdn), norm))
    t_own = timed(lambda: fused_ops.linear_layer_norm(hdn, lin2, res, norm))
    fl = 2.0 * rows * 1024 * 128
    print("linear2 (K=1024) + residual + LayerNorm: library GEMM + add_layernorm %.3f ms   fused split-bf16 %.3f ms (%.0f TF/s fp32-equivalent)" % (
        t_lib, t_own, fl / t_own / 1e9))
