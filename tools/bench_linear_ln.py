#!/usr/bin/env python3
"""Time the encoder FFN tail kernel (linear2 + residual + LayerNorm, K = 1024 -> 128; csrc/linear_ln_split.hip) and linear1 at
the bench's row count (batch 128: 2 785 280 rows)."""
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
    h = fused_ops.linear_k128(x, lin1.weight, lin1.bias, relu=True)
    t1 = timed(lambda: fused_ops.linear_k128(x, lin1.weight, lin1.bias, relu=True))
    t2 = timed(lambda: fused_ops.linear_layer_norm(h, lin2, x, norm))
    fl = 2.0 * rows * 128 * 1024
    print("rows=%d  linear1+ReLU %.3f ms (%.0f TF/s fp32-equiv)   linear2+res+LN %.3f ms (%.0f TF/s)" % (rows, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9))
