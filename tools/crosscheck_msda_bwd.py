#!/usr/bin/env python3
"""Cross-check of the pyramid-column MSDeformAttn backward against the generic kernel on random geometries (development
tool): ragged pyramids of 3-5 levels, 1-9 heads, several batch sizes and location statistics, both kernels forced through
the C ABI's diagnostic switch.  Tolerance 3e-5 of each gradient's magnitude (both kernels sum in different orders)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_cases import make_case  # noqa: E402
from pctrans_amd import MultiScaleDeformableAttention as MSDA, _lib  # noqa: E402

lib = _lib.lib()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
worst = 0.0
for i in range(ncase):
    L = int(rng.choice([3, 4, 5]))
    base_h, base_w = int(rng.randint(1, 12)), int(rng.randint(1, 14))
    shapes = []
    for l in range(L):
        f = 2 ** l
        shapes.append((max(1, base_h * f + int(rng.randint(-1, 2))), max(1, base_w * f + int(rng.randint(-1, 2)))))
    if rng.rand() < 0.3:
        shapes = shapes[::-1]
    M = int(rng.choice([1, 2, 3, 4, 8, 9]))
    N = int(rng.choice([1, 2, 3, 5]))
    S = sum(h * w for h, w in shapes)
    kind = rng.choice(["M", "I", "U", "E", "W"])
    kw = dict(M=dict(model_like=True, px_sigma=float(rng.choice([0.5, 2.0, 5.0]))), I=dict(init_like=True),
              U=dict(), E=dict(lo=-0.4, hi=1.4), W=dict(model_like=True, px_sigma=15.0))[kind]
    c = make_case(seed=1000 + i, N=N, M=M, D=16, Lq=S, P=4, shapes=shapes, **kw)
    go = rng.standard_normal((N, S, M * 16)).astype(np.float32)
    args = [dev(c[k]) for k in ("value", "shapes", "starts", "loc", "attn")] + [dev(go), 64]
    res = {}
    for name, ch in (("col", 3), ("generic", 2)):
        lib.pct_msda_set_bwd_kernel_choice(ch)
        g = MSDA.ms_deform_attn_backward(*args)
        torch.cuda.synchronize()
        assert lib.pct_msda_last_bwd_kernel() == ch, (name, lib.pct_msda_last_bwd_kernel())
        res[name] = [t.cpu().numpy() for t in g]
    lib.pct_msda_set_bwd_kernel_choice(-1)
    wh = np.stack([c["shapes"][:, 1], c["shapes"][:, 0]], -1).astype(np.float64)
    pix = c["loc"].astype(np.float64) * wh[None, None, None, :, None, :] - 0.5
    edge = (np.abs(pix - np.round(pix)) < 1e-3).any(-1, keepdims=True)
    errs = []
    for a, b, nm in zip(res["col"], res["generic"], ("grad_value", "grad_loc", "grad_attn")):
        if nm == "grad_loc":
            a, b = np.where(edge, 0, a), np.where(edge, 0, b)
        e = float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
        errs.append(e)
    worst = max(worst, max(errs))
    print("case %2d  L=%d shapes=%s M=%d N=%d kind=%s  rel err gv %.1e gl %.1e ga %.1e %s" % (
        i, L, shapes, M, N, kind, errs[0], errs[1], errs[2], "" if max(errs) <= 3e-5 else "  <-- MISMATCH"), flush=True)
print("worst %.2e  %s" % (worst, "OK" if worst <= 3e-5 else "FAILED"))
sys.exit(0 if worst <= 3e-5 else 1)
