#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the windowed MSDeformAttn kernel (in-kernel s_memtime stamps, wave 0 of each
workgroup).  Read SHARES, not absolute time: the stamped build forbids overlaps the real kernel has."""
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
buf = torch.zeros(1024 * 16, dtype=torch.int64, device="cuda")
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.pct_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
MSDA.ms_deform_attn_forward(v, sh, st, loc, w, 128)
torch.cuda.synchronize()
lib.pct_debug_set_stamp_buffer(buf.data_ptr())
MSDA.ms_deform_attn_forward(v, sh, st, loc, w, 128)
torch.cuda.synchronize()
lib.pct_debug_set_stamp_buffer(None)
kern = _lib.lib().pct_msda_last_kernel()
if kern == 4:   # pyramid-column kernel: 16 slots per workgroup, in program order
    t = buf.view(1024, 16).double().cpu()
    t = t[t.sum(1) > 0]
    order = [(7, "records waited for + transposed, pixel coordinates"), (0, "queue + box pre-pass + weight loads issued"),
             (1, "barrier A (boxes)"), (2, "windows / phases planned"), (8, "first phase: LDS-DMA issued"),
             (3, "next item decoded"), (4, "staging barrier (DMA landed)"), (9, "weights transposed (fused: soft-max)"),
             (10, "first level gathered"), (5, "other levels (incl. later phases)"), (6, "store + hand-over")]
else:
    t = buf.view(2048, 8)[:1024].double().cpu()
    order = list(enumerate(["first prepass", "barrier A (boxes, pool free)", "windows", "staging issue", "barrier B (staged)",
                            "gather+store", "next item's prepass"]))
tot = t.sum(1).mean().item()
print("dist=%s N=%d  mean cycles per WG %.0f" % (dist, N, tot))
for i, n in order:
    print("  %-52s %6.1f %%   %10.0f cyc/WG" % (n, 100 * t[:, i].mean().item() / tot, t[:, i].mean().item()))
