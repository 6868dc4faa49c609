#!/usr/bin/env python3
"""Profiler workload: the pyramid-column kernel on the reference layout and on piece planes (plain op + fused front-end),
3 launches each, P2 at the given batch / distribution.  Run under rocprofv3 (tools/pmc_planes.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_msda_op import SHAPES, make  # noqa: E402
from pctrans_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402

dist = sys.argv[1] if len(sys.argv) > 1 else "M"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
shapes, P = SHAPES["P2"]
v, sh, st, loc, w = make(shapes, P, N, dist, torch.float32)
M = v.shape[2]
vp, lp, wp = MSDA.to_planes(v, M), MSDA.to_planes(loc, M), MSDA.to_planes(w, M)
for _ in range(3):
    MSDA.ms_deform_attn_forward(v, sh, st, loc, w, 128)
    MSDA.ms_deform_attn_forward_planes(vp, sh, st, lp, wp, M)
torch.cuda.synchronize()
