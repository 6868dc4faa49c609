#!/usr/bin/env python3
"""The encoder FFN at the bench's row count: linear1 + ReLU and linear2 + residual + LayerNorm as two kernels (the hidden tensor
through HBM) against the fused kernel (csrc/ffn_fused_split.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pctrans_amd import fused_ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 21760


def timed(fn, n=10):
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


with torch.no_grad():
    x = torch.randn(rows, 128, device="cuda")
    lin1 = torch.nn.Linear(128, 1024).cuda()
    lin2 = torch.nn.Linear(1024, 128).cuda()
    norm = torch.nn.LayerNorm(128).cuda()
    two = lambda: fused_ops.linear_layer_norm(fused_ops.linear_k128(x, lin1.weight, lin1.bias, relu=True), lin2, x, norm)
    one = lambda: fused_ops.ffn_layer_norm(x, lin1, lin2, norm)
    a, b = two(), one()
    print("max |two kernels - fused| = %.2e" % float((a - b).abs().max()))
    fl = 4.0 * rows * 128 * 1024
    for _ in range(2):
        t2, t1 = timed(two), timed(one)
        print("rows=%d  two kernels %.3f ms (%.0f TF/s fp32-equiv)   fused %.3f ms (%.0f TF/s)" % (rows, t2, fl / t2 / 1e9, t1, fl / t1 / 1e9))
