#!/usr/bin/env python3
"""Soak test of the windowed MSDeformAttn kernels with the dynamic item queue: many launches per shape, the forward must
be bit-identical every time, the backward must stay within its float-atomics noise (development tool)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_cases import make_case  # noqa: E402
from pctrans_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cases = [
    dict(N=8, shapes=[(16, 16), (32, 32), (64, 64), (128, 128)]),
    dict(N=3, shapes=[(16, 16), (32, 32), (64, 64)]),
    dict(N=5, shapes=[(17, 22), (33, 44), (65, 87)]),
    dict(N=1, shapes=[(16, 16), (32, 32), (64, 64), (128, 128)]),
]
d = lambda a: torch.from_numpy(a).cuda()
bad = 0
for i, cs in enumerate(cases):
    S = sum(h * w for h, w in cs["shapes"])
    c = make_case(seed=10 + i, M=8, D=16, P=4, Lq=S, model_like=True, **cs)
    args = [d(c[k]) for k in ("value", "shapes", "starts", "loc", "attn")]
    first = MSDA.ms_deform_attn_forward(*args, 64)
    go = torch.randn_like(first)
    gv0, gl0, ga0 = MSDA.ms_deform_attn_backward(*args, go, 64)
    scale = float(gv0.abs().max())
    for r in range(reps):
        out = MSDA.ms_deform_attn_forward(*args, 64)
        if not torch.equal(out, first):
            bad += 1
            print("case", i, "rep", r, "forward differs: max", float((out - first).abs().max()), flush=True)
        if r % 10 == 0:
            gv, gl, ga = MSDA.ms_deform_attn_backward(*args, go, 64)
            if not (torch.equal(gl, gl0) and torch.equal(ga, ga0)) or float((gv - gv0).abs().max()) > 1e-5 * max(scale, 1.0):
                bad += 1
                print("case", i, "rep", r, "backward differs", float((gv - gv0).abs().max()), flush=True)
    torch.cuda.synchronize()
    print("case", i, "done", flush=True)
print("mismatches:", bad)
