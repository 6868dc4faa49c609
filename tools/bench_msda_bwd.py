#!/usr/bin/env python3
"""Op-level timing of the MSDeformAttn backward (development tool; the record kept under profiles/ comes from here).
Algorithmic bytes per (query, all heads), each operand once: read value, locations, weights, grad_output; write grad_value,
grad_sampling_loc, grad_attn_weight = 2*(2*M*D) + 2*(3*M*L*P) elements -> P2 fp32: 4 608 B per query."""
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
ap.add_argument("--cases", default="P1:8,P2:2,P2:8,P2:32")
ap.add_argument("--dists", default="I,M,U")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
for case in a.cases.split(","):
    sname, N = case.split(":")
    N = int(N)
    shapes, P = SHAPES[sname]
    for dist in a.dists.split(","):
        v, sh, st, loc, w = make(shapes, P, N, dist, torch.float32)
        S, M, D, L = v.shape[1], v.shape[2], v.shape[3], len(shapes)
        go = torch.randn(N, S, M * D, device="cuda")
        for _ in range(3):
            MSDA.ms_deform_attn_backward(v, sh, st, loc, w, go, 128)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            MSDA.ms_deform_attn_backward(v, sh, st, loc, w, go, 128)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        alg = N * S * 4 * (2 * (2 * M * D) + 2 * (3 * M * L * P))
        print("%s f32 N=%-3d dist=%s  backward %.3f ms  alg %.1f MB  %.1f GB/s  frac %.3f of 8 TB/s" % (
            sname, N, dist, ms, alg / 1e6, alg / ms / 1e6, alg / ms / 1e6 / 8000.0), flush=True)
