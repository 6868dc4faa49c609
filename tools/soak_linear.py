#!/usr/bin/env python3
"""Soak test of the split-bf16 GEMM kernels: repeated launches must be bit-identical (no atomics anywhere)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pctrans_amd import fused_ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(0)
bad = 0
with torch.no_grad():
    for rows in (21760 * 8, 4099 * 3 + 5):
        x = torch.randn(rows, 128, device="cuda")
        pos = torch.randn(rows, 128, device="cuda")
        res = torch.randn(rows, 128, device="cuda")
        h = torch.randn(rows, 1024, device="cuda").relu_()
        l1, l2 = torch.nn.Linear(128, 1024).cuda(), torch.nn.Linear(1024, 128).cuda()
        lv, lo, la, lp = (torch.nn.Linear(128, n).cuda() for n in (128, 256, 128, 128))
        norm = torch.nn.LayerNorm(128).cuda()
        fns = {
            "linear1+relu": lambda: fused_ops.linear_k128(x, l1.weight, l1.bias, relu=True),
            "multi": lambda: torch.cat(fused_ops.linear_k128_multi(x, ((lv, False), (lo, True), (la, True)), x_add=pos), -1),
            "out_proj+ln": lambda: fused_ops.linear_add_layer_norm(x, lp, res, norm),
            "linear2+ln": lambda: fused_ops.linear_layer_norm(h, l2, res, norm),
        }
        for name, fn in fns.items():
            first = fn()
            for r in range(reps):
                if not torch.equal(fn(), first):
                    bad += 1
                    print(rows, name, "rep", r, "differs", flush=True)
            torch.cuda.synchronize()
            print(rows, name, "ok", flush=True)
print("mismatches:", bad)
