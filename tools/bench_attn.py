#!/usr/bin/env python3
"""Stand-alone timing of the decoder's cross-attention core: split-operand kernel (csrc/cross_attention.hip) against the
generic kernel on concatenated operands (csrc/masked_attention.hip) incl. the two torch.cat passes it needs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pctrans_amd import fused_ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L, heads, C = 100, 8, 128


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


for S in (4096, 1024, 256):
    g = torch.Generator(device="cuda").manual_seed(0)
    mk = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
    qc, qp, kc, kp, v = mk(L, N, C), mk(L, N, C), mk(S, N, C), mk(S, N, C), mk(S, N, C)
    mask = torch.rand(N, 1, L, S, device="cuda", generator=g) < 0.7
    mask[..., 0] = False
    q = torch.cat([qc.view(L, N, heads, 16), qp.view(L, N, heads, 16)], 3).reshape(L, N, 2 * C)
    k = torch.cat([kc.view(S, N, heads, 16), kp.view(S, N, heads, 16)], 3).reshape(S, N, 2 * C)
    vt = v.permute(1, 2, 0).contiguous()
    t_new = timeit(lambda: fused_ops.cross_attention(qc, qp, kc, kp, v, heads, mask))
    t_old = timeit(lambda: fused_ops.masked_attention(q, k, None, heads, mask, v_t=vt))
    t_cat = timeit(lambda: torch.cat([kc.view(S, N, heads, 16), kp.view(S, N, heads, 16)], 3))
    flops = 2.0 * N * heads * L * S * (32 + 16)
    byts = 2.0 * (3 * S * N * C + 3 * L * N * C) + N * L * S
    for name, t in (("split-operand kernel", t_new), ("generic kernel (operands pre-concatenated)", t_old)):
        print("N=%d S=%4d %-44s %.3f ms  %6.1f TFLOP/s useful  %5.2f TB/s of operand bytes" % (N, S, name, t, flops / t / 1e9, byts / t / 1e9))
    print("N=%d S=%4d torch.cat of the key halves: %.3f ms" % (N, S, t_cat))
