#!/usr/bin/env python3
"""Op-level timing of the MSDeformAttn backward kernel (development tool)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_msda_op import SHAPES, make  # noqa: E402
from pctrans_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402

for sname, N in (("P1", 8), ("P2", 2), ("P2", 8)):
    shapes, P = SHAPES[sname]
    for dist in ("I", "M", "U"):
        v, sh, st, loc, w = make(shapes, P, N, dist, torch.float32)
        go = torch.randn(N, v.shape[1], 128, device="cuda")
        for _ in range(3):
            MSDA.ms_deform_attn_backward(v, sh, st, loc, w, go, 128)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            MSDA.ms_deform_attn_backward(v, sh, st, loc, w, go, 128)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        atom_bytes = N * v.shape[1] * 8 * len(shapes) * P * 4 * 64
        print("%s N=%d dist=%s  backward %.3f ms  (atomic bytes %.2f GB -> %.2f TB/s)" % (
            sname, N, dist, ms, atom_bytes / 1e9, atom_bytes / ms / 1e9))
