#!/usr/bin/env python3
"""Op-level A/B: the pyramid-column kernel on the reference layout against the same kernel on piece-plane operands
(pct_ms_deform_attn_forward_planes_f32), plain op and fused front-end, P2 at the given batch, distributions M and I.
HIP events on the launch stream; prints ms per launch and the roofline fraction (algorithmic bytes / 8 TB/s)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_msda_op import SHAPES, make  # noqa: E402
from pctrans_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
shapes, P = SHAPES["P2"]


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for dist in ("M", "I"):
    v, sh, st, loc, w = make(shapes, P, N, dist, torch.float32)
    Nn, S, M, D = v.shape
    L = sh.shape[0]
    alg = Nn * S * (2 * M * D * 4 + 3 * M * L * P * 4)
    vp, lp, wp = MSDA.to_planes(v, M), MSDA.to_planes(loc, M), MSDA.to_planes(w, M)
    a = MSDA.ms_deform_attn_forward(v, sh, st, loc, w, 128)
    b = MSDA.ms_deform_attn_forward_planes(vp, sh, st, lp, wp, M)
    same = torch.equal(a.view(torch.int32), b.view(torch.int32))
    for rep in range(REPS):                                # alternate: A B A B ...
        ta = timeit(lambda: MSDA.ms_deform_attn_forward(v, sh, st, loc, w, 128))
        tb = timeit(lambda: MSDA.ms_deform_attn_forward_planes(vp, sh, st, lp, wp, M))
        print("plain  dist=%s N=%d  reference layout %.3f ms (%.3f)   piece planes %.3f ms (%.3f)   bit-identical=%s" % (
            dist, N, ta, alg / ta / 1e-3 / 8e12, tb, alg / tb / 1e-3 / 8e12, same), flush=True)
    # fused front-end: offsets and logits that reproduce the same locations / weights
    ref = []
    for h, ww in shapes:
        ys, xs = torch.meshgrid((torch.arange(h, device="cuda") + 0.5) / h, (torch.arange(ww, device="cuda") + 0.5) / ww,
                                indexing="ij")
        ref.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(ref, 0)[None, :, None, :].expand(1, S, L, 2).contiguous()
    norm = torch.stack([sh[:, 1], sh[:, 0]], -1).float()
    off = ((loc - ref[:, :, None, :, None, :]) * norm[None, None, None, :, None, :]).contiguous()
    logits = torch.log(w.clamp_min(1e-30)).reshape(Nn, S, M, L * P).contiguous()
    del loc, w, lp, wp
    op, gp = MSDA.to_planes(off, M), MSDA.to_planes(logits, M)
    a = MSDA.ms_deform_attn_fused_forward(v, sh, st, ref.expand(Nn, -1, -1, -1), off, logits)
    b = MSDA.ms_deform_attn_forward_planes(vp, sh, st, op, gp, M, reference_points=ref)
    same = torch.equal(a.view(torch.int32), b.view(torch.int32))
    for rep in range(REPS):
        ta = timeit(lambda: MSDA.ms_deform_attn_fused_forward(v, sh, st, ref.expand(Nn, -1, -1, -1), off, logits))
        tb = timeit(lambda: MSDA.ms_deform_attn_forward_planes(vp, sh, st, op, gp, M, reference_points=ref))
        print("fused  dist=%s N=%d  reference layout %.3f ms (%.3f)   piece planes %.3f ms (%.3f)   bit-identical=%s" % (
            dist, N, ta, alg / ta / 1e-3 / 8e12, tb, alg / tb / 1e-3 / 8e12, same), flush=True)
    del v, vp, off, logits, op, gp, a, b
    torch.cuda.empty_cache()
