#!/usr/bin/env python3
"""Stand-alone timing of the dynamic mask head kernels (development tool)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pctrans_amd import dynamic_mask_head as dmh  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Q, H, W = 100, 128, 128
g = torch.Generator(device="cuda").manual_seed(0)
mf = torch.randn(N, 16, H, W, device="cuda", generator=g)
ref = torch.rand(N, Q, 2, device="cuda", generator=g)
prm = torch.randn(N, Q, 233, device="cuda", generator=g) * 0.2


def timeit(fn, iters=10):
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


for tgt in ((16, 16), (32, 32), (64, 64)):
    for kern in ("fused", "mfma", "fused", "mfma", "valu"):
        t = timeit(lambda: dmh.dynamic_mask_head_forward(mf, ref, prm, 4, True, tgt, out_dtype=torch.bfloat16, kernel=kern))
        print("N=%d target=%s kernel=%s  %.3f ms" % (N, tgt, kern, t))
t = timeit(lambda: dmh.dynamic_mask_head_forward(mf, ref, prm, 4, True, (32, 32), out_dtype=torch.float32))
print("N=%d fp32 fused kernel %.3f ms" % (N, t))
x = torch.empty(N, Q, 2 * H, 2 * W, dtype=torch.bfloat16, device="cuda")
print("plain fill of the output tensor: %.3f ms" % timeit(lambda: x.fill_(1.0)))
y = torch.empty(N, Q, H, W, dtype=torch.bfloat16, device="cuda")
print("torch bf16 interpolate x2: %.3f ms" % timeit(
    lambda: torch.nn.functional.interpolate(y, size=(2 * H, 2 * W), mode="bilinear", align_corners=False)))
