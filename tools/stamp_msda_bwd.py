#!/usr/bin/env python3
"""Diagnostic: per-part cycle shares of the pyramid-column MSDeformAttn backward (in-kernel s_memtime stamps, wave 0 of each
workgroup; needs a library built with -DPCT_BCOL_STAMP=1: tools/variant_file.sh bc_stamp msda_backward_col -DPCT_BCOL_STAMP=1).
Read SHARES, not absolute time: the stamped build forbids overlaps the real kernel has."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_msda_op import SHAPES, make  # noqa: E402
from pctrans_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402
from pctrans_amd import _lib  # noqa: E402

dist = sys.argv[1] if len(sys.argv) > 1 else "I"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
shapes, P = SHAPES["P2"]
v, sh, st, loc, w = make(shapes, P, N, dist, torch.float32)
go = torch.randn(N, v.shape[1], v.shape[2] * v.shape[3], device="cuda")
buf = torch.zeros(1024 * 16, dtype=torch.int64, device="cuda")
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.pct_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
MSDA.ms_deform_attn_backward(v, sh, st, loc, w, go, 128)
torch.cuda.synchronize()
lib.pct_debug_set_stamp_buffer(buf.data_ptr())
MSDA.ms_deform_attn_backward(v, sh, st, loc, w, go, 128)
torch.cuda.synchronize()
lib.pct_debug_set_stamp_buffer(None)
t = buf.view(1024, 16).double().cpu()
t = t[t.sum(1) > 0]
order = [(0, "records transposed, boxes, weight / grad_out loads issued"), (1, "barrier A"),
         (2, "planning, staging issued, next item decoded, front end"), (3, "staging landed + barrier B1"),
         (9, "later phases: barrier, staging, barrier"), (4, "dots pass"), (5, "barrier B2, zeroing, barrier B3, grad_out rotated"),
         (6, "scatter pass"), (7, "barrier B4"), (8, "flush"), (10, "flags, hand-over")]
tot = t.sum(1).mean().item()
print("dist=%s N=%d  workgroups %d  mean cycles per WG %.0f" % (dist, N, t.shape[0], tot))
for i, n in order:
    print("  %-62s %6.1f %%   %10.0f cyc/WG" % (n, 100 * t[:, i].mean().item() / tot, t[:, i].mean().item()))
