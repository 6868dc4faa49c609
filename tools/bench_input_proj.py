#!/usr/bin/env python3
"""The pixel decoder's four input projections at the bench shape (batch 128): conv1x1 kernel + GroupNorm statistics + apply /
transpose (three kernels, NCHW intermediate) against the one-entry form (token epilogue + per-tile records + in-place apply)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pctrans_amd import fused_ops
from pctrans_amd.layers import Conv2d

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


levels = ((2048, 16), (1024, 32), (512, 64), (256, 128))
with torch.no_grad():
    mods = [(Conv2d(K, 128, kernel_size=1).cuda(), torch.nn.GroupNorm(32, 128).cuda(), torch.randn(N, K, S, S, device="cuda"))
            for K, S in levels]
    total = sum(S * S for _, S in levels)
    out = torch.empty(N, total, 128, device="cuda")

    def old():
        start = 0
        for conv, gn, x in mods:
            c = fused_ops.conv1x1_nchw(x, conv)
            fused_ops.groupnorm_flatten_into(c, gn, out, start)
            start += x.shape[2] * x.shape[3]

    def new():
        start = 0
        for conv, gn, x in mods:
            fused_ops.conv1x1_groupnorm_tokens_into(x, conv, gn, out, start)
            start += x.shape[2] * x.shape[3]

    old()
    a = out.clone()
    new()
    print("max |old - new| = %.2e" % float((a - out).abs().max()))
    for _ in range(2):
        print("three kernels + NCHW intermediate %.3f ms   one entry %.3f ms" % (timeit(old), timeit(new)))
