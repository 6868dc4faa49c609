#!/usr/bin/env python3
"""Op-level A/B: fused front-end entry vs unfused op on the same 'I' (init-bias) sampling pattern."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pctrans_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [(16, 16), (32, 32), (64, 64), (128, 128)]
M, D, P, L = 8, 16, 4, 4
sh = torch.tensor(shapes, dtype=torch.long, device="cuda")
S = int(sh.prod(1).sum())
st = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
g = torch.Generator(device="cuda").manual_seed(0)
value = torch.randn(N, S, M, D, device="cuda", generator=g)
ref = []
for h, w in shapes:
    ys, xs = torch.meshgrid((torch.arange(h, device="cuda") + 0.5) / h, (torch.arange(w, device="cuda") + 0.5) / w,
                            indexing="ij")
    ref.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
ref = torch.cat(ref, 0)[None, :, None, :].expand(1, S, L, 2).contiguous()          # [1, S, L, 2]
th = torch.arange(M, device="cuda", dtype=torch.float32) * (2 * math.pi / M)
gi = torch.stack([th.cos(), th.sin()], -1)
gi = gi / gi.abs().max(-1, keepdim=True)[0]
off = (gi.view(1, 1, M, 1, 1, 2) * torch.arange(1, P + 1, device="cuda").view(1, 1, 1, 1, P, 1)).expand(
    N, S, M, L, P, 2).contiguous()
logits = torch.zeros(N, S, M, L * P, device="cuda")
norm = torch.stack([sh[:, 1], sh[:, 0]], -1).float()
loc = (ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]).contiguous()
w = torch.softmax(logits, -1).view(N, S, M, L, P).contiguous()


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


a = MSDA.ms_deform_attn_forward(value, sh, st, loc, w, 128)
b = MSDA.ms_deform_attn_fused_forward(value, sh, st, ref.expand(N, -1, -1, -1), off, logits)
print("max |fused - unfused| = %.2e" % float((a - b).abs().max()))
print("N=%d unfused %.3f ms   fused %.3f ms" % (N, timeit(lambda: MSDA.ms_deform_attn_forward(value, sh, st, loc, w, 128)),
                                               timeit(lambda: MSDA.ms_deform_attn_fused_forward(
                                                   value, sh, st, ref.expand(N, -1, -1, -1), off, logits))))
