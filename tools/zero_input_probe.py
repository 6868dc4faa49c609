#!/usr/bin/env python3
"""Is a kernel limited by power (DVFS) or by its own schedule?  Same launch on random and on all-zero operands: the
instruction stream is identical, only the switching activity differs (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
with torch.no_grad():
    lin1, lin2, norm = torch.nn.Linear(128, 1024).cuda(), torch.nn.Linear(1024, 128).cuda(), torch.nn.LayerNorm(128).cuda()
    for name, scale in (("random", 1.0), ("zeros", 0.0)):
        x = torch.randn(rows, 128, device="cuda") * scale
        h = torch.randn(rows, 1024, device="cuda").relu_() * scale
        res = torch.randn(rows, 128, device="cuda") * scale
        if scale == 0.0:
            for m in (lin1, lin2):
                m.weight.zero_()
                m.bias.zero_()
        t1 = timed(lambda: fused_ops.linear_k128(x, lin1.weight, lin1.bias, relu=True))
        t2 = timed(lambda: fused_ops.linear_layer_norm(h, lin2, res, norm))
        print("%-6s operands: linear1 %.3f ms   linear2+LN %.3f ms" % (name, t1, t2))
