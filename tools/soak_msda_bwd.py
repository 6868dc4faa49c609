#!/usr/bin/env python3
"""Soak of the pyramid-column MSDeformAttn backward (development tool): many launches of mixed sizes back to back, each
compared with a reference launch of the same inputs -- grad_loc / grad_attn bitwise (integer LDS sums), grad_value to the float
atomics' noise -- so that a rare race, a stale queue counter or flag word shows up."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_msda_op import SHAPES, make  # noqa: E402
from pctrans_amd import MultiScaleDeformableAttention as MSDA, _lib  # noqa: E402

lib = _lib.lib()
lib.pct_msda_set_bwd_kernel_choice(3)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cases = []
for sname, N, dist in (("P2", 2, "M"), ("P2", 8, "I"), ("P1", 8, "M"), ("P2", 3, "U"), ("P2", 16, "M"), ("P1", 1, "I")):
    shapes, P = SHAPES[sname]
    v, sh, st, loc, w = make(shapes, P, N, dist, torch.float32)
    go = torch.randn(N, v.shape[1], v.shape[2] * v.shape[3], device="cuda")
    ref = MSDA.ms_deform_attn_backward(v, sh, st, loc, w, go, 128)
    torch.cuda.synchronize()
    assert lib.pct_msda_last_bwd_kernel() == 3
    cases.append(((v, sh, st, loc, w, go, 128), ref, "%s N=%d %s" % (sname, N, dist)))
bad = 0
for r in range(rounds):
    outs = [MSDA.ms_deform_attn_backward(*a) for a, _, _ in cases]           # back to back, no synchronisation in between
    torch.cuda.synchronize()
    for (a, ref, name), o in zip(cases, outs):
        ok = torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2])
        e = float((o[0] - ref[0]).abs().max()) / max(1e-30, float(ref[0].abs().max()))
        if not ok or not (e <= 1e-5):
            bad += 1
            print("round %d %s: grad_loc/attn bitwise %s, grad_value rel err %.2e" % (r, name, ok, e), flush=True)
print("%d rounds x %d cases, %d mismatches" % (rounds, len(cases), bad))
sys.exit(1 if bad else 0)
