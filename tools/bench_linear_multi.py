#!/usr/bin/env python3
"""value_proj + sampling_offsets + attention_weights: three launches of the K = 128 kernel vs one merged launch."""
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


S, N = 21760, 64
x = torch.randn(N, S, 128, device="cuda")
pos = torch.randn(1, S, 128, device="cuda")
lv, lo, la = (torch.nn.Linear(128, n).cuda() for n in (128, 256, 128))
with torch.no_grad():
    t3 = timed(lambda: (fused_ops.linear_k128(x, lv.weight, lv.bias),
                        fused_ops.linear_k128(x, lo.weight, lo.bias, x_add=pos),
                        fused_ops.linear_k128(x, la.weight, la.bias, x_add=pos)))
    t1 = timed(lambda: fused_ops.linear_k128_multi(x, ((lv, False), (lo, True), (la, True)), x_add=pos))
print("three launches %.3f ms   merged %.3f ms" % (t3, t1))
