#!/usr/bin/env python3
"""A/B of the pixel decoder's four 1x1 input projections at the bench shape (batch 128, 512^2 image, ResNet-50 channels):
layers.Conv2d (strided-batched fp32 GEMM through rocBLAS / hipBLASLt) against csrc/conv1x1_split.hip."""
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


tot_a = tot_b = 0.0
with torch.no_grad():
    for K, S in ((256, 128), (512, 64), (1024, 32), (2048, 16)):
        conv = Conv2d(K, 128, kernel_size=1).cuda()
        x = torch.randn(N, K, S, S, device="cuda")
        a = timeit(lambda: conv(x))
        b = timeit(lambda: fused_ops.conv1x1_nchw(x, conv))
        gb = (x.numel() + N * 128 * S * S) * 4 / 1e9
        print("K=%4d %3dx%-3d  library %.3f ms (%.2f TB/s)   split kernel %.3f ms (%.2f TB/s)" % (K, S, S, a, gb / a, b, gb / b))
        tot_a += a
        tot_b += b
print("sum: library %.3f ms, split kernel %.3f ms" % (tot_a, tot_b))
