#!/usr/bin/env python3
"""Full-size cross-check of the column kernels against the generic kernel (same inputs, two code paths): the shapes the CPU
oracle no longer finishes quickly.  Development tool; the judged parity tests are tests/test_msda*_gpu.py."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_msda_op import SHAPES, make  # noqa: E402
from pctrans_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402
from pctrans_amd import _lib  # noqa: E402

lib = _lib.lib()
worst = 0.0
for sname, dt, N, tol in (("P2", torch.float32, 3, 2e-5), ("P4", torch.float32, 5, 2e-5), ("P1", torch.float32, 9, 2e-5),
                          ("P3", torch.float16, 2, 2e-3), ("P3", torch.bfloat16, 1, 1.6e-2), ("P2", torch.bfloat16, 3, 1.6e-2),
                          ("P2", torch.float16, 2, 2e-3)):
    shapes, P = SHAPES[sname]
    for dist in ("I", "M", "U"):
        for seed in (0, 1):
            v, sh, st, loc, a = make(shapes, P, N, dist, dt, seed=seed)
            if dist == "U":            # out-of-map samples too
                loc = loc * 1.3 - 0.15
            out = {}
            for k in (4, 2):
                lib.pct_msda_set_kernel_choice(k)
                out[k] = MSDA.ms_deform_attn_forward(v, sh, st, loc, a, 128).float()
                assert lib.pct_msda_last_kernel() == k, (sname, dt, k, lib.pct_msda_last_kernel())
            lib.pct_msda_set_kernel_choice(-1)
            err = (out[4] - out[2]).abs().max().item()
            ref = out[2].abs().max().item()
            worst = max(worst, err / tol)
            print("%s %-8s N=%d dist=%s seed=%d  max|col - generic| = %.3e (max|out| %.2f)  %s" % (
                sname, str(dt).split(".")[1], N, dist, seed, err, ref, "ok" if err <= tol * max(1.0, ref) else "MISMATCH"),
                flush=True)
            assert err <= tol * max(1.0, ref)
print("all within tolerance; worst err/tol = %.2f" % worst)
