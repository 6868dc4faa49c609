#!/usr/bin/env python3
"""Run ONE MSDeformAttn forward configuration a few times (for rocprofv3 --pmc / --kernel-trace runs)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_msda_op import SHAPES, make  # noqa: E402
from pctrans_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="P2")
ap.add_argument("--dist", default="I")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--dtype", default="f32")
a = ap.parse_args()
dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
shapes, P = SHAPES[a.shape]
v, sh, st, loc, w = make(shapes, P, a.batch, a.dist, dt)
for _ in range(a.iters):
    MSDA.ms_deform_attn_forward(v, sh, st, loc, w, 128)
torch.cuda.synchronize()
print("done")
