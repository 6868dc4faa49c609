#!/usr/bin/env python3
"""Op-level sweep of the MSDeformAttn forward kernel: shapes P1..P4 x location distributions U / M.
Prints one line per case: time per launch (HIP events on the launch stream), algorithmic GB/s and roofline fraction.
Development tool; the judged numbers come from bench.py."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pctrans_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402

PEAK = 8.0e12

SHAPES = {
    "P1": ([(16, 16), (32, 32), (64, 64)], 4),
    "P2": ([(16, 16), (32, 32), (64, 64), (128, 128)], 4),
    "P3": ([(16, 16), (32, 32), (64, 64), (128, 128), (256, 256)], 8),
    "P4": ([(17, 22), (33, 44), (65, 87)], 4),
}


def make(shapes, P, N, dist, dtype, M=8, D=16, sigma=2.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    sh = torch.tensor(shapes, dtype=torch.long, device="cuda")
    L = sh.shape[0]
    S = int(sh.prod(1).sum())
    starts = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    value = torch.randn(N, S, M, D, device="cuda", generator=g).to(dtype)
    a = torch.softmax(torch.randn(N, S, M, L * P, device="cuda", generator=g), -1).view(N, S, M, L, P)
    if dist == "U":
        loc = torch.rand(N, S, M, L, P, 2, device="cuda", generator=g)
    else:
        ref = []
        for h, w in shapes:
            ys, xs = torch.meshgrid((torch.arange(h, device="cuda") + 0.5) / h,
                                    (torch.arange(w, device="cuda") + 0.5) / w, indexing="ij")
            ref.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
        ref = torch.cat(ref, 0)
        norm = torch.stack([sh[:, 1], sh[:, 0]], -1).float()
        if dist == "M":      # gaussian offsets, sigma px on the sampled level
            off = torch.randn(N, S, M, L, P, 2, device="cuda", generator=g) * sigma
        else:                # "I": the module's init bias -- head-directional offsets of 1..P px
            import math
            th = torch.arange(M, device="cuda", dtype=torch.float32) * (2 * math.pi / M)
            gi = torch.stack([th.cos(), th.sin()], -1)
            gi = gi / gi.abs().max(-1, keepdim=True)[0]
            off = gi.view(1, 1, M, 1, 1, 2) * torch.arange(1, P + 1, device="cuda").view(1, 1, 1, 1, P, 1)
            off = off.expand(N, S, M, L, P, 2)
        loc = ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
        loc = loc.contiguous()
    if dtype in (torch.float16, torch.bfloat16):
        loc, a = loc.float(), a.float()
    else:
        loc, a = loc.to(dtype), a.to(dtype)
    return value, sh, starts, loc, a.contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="P1,P2,P4")
    ap.add_argument("--dists", default="U,M,I")
    ap.add_argument("--batches", default="1,8,32")
    ap.add_argument("--dtypes", default="f32")
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dts = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16, "f64": torch.float64}
    for sname in args.shapes.split(","):
        shapes, P = SHAPES[sname]
        for dn in args.dtypes.split(","):
            dt = dts[dn]
            for N in [int(x) for x in args.batches.split(",")]:
                for dist in args.dists.split(","):
                    v, sh, st, loc, a = make(shapes, P, N, dist, dt)
                    S, M, D, L = v.shape[1], v.shape[2], v.shape[3], sh.shape[0]
                    e = v.element_size()
                    le = loc.element_size()
                    bytes_alg = N * S * (2 * M * D * e + 3 * M * L * P * le)
                    for _ in range(5):
                        MSDA.ms_deform_attn_forward(v, sh, st, loc, a, 128)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        MSDA.ms_deform_attn_forward(v, sh, st, loc, a, 128)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / args.iters
                    gbs = bytes_alg / (ms * 1e-3) / 1e9
                    from pctrans_amd import _lib
                    kern = {0: "-", 1: "win", 2: "generic", 3: "dpp", 4: "col"}[_lib.lib().pct_msda_last_kernel()]
                    print("%s %-4s N=%-3d dist=%s  %8.3f ms  alg %7.1f MB  %8.1f GB/s  frac %.3f  [%s]" % (
                        sname, dn, N, dist, ms, bytes_alg / 1e6, gbs, gbs * 1e9 / PEAK, kern), flush=True)
                    del v, loc, a


if __name__ == "__main__":
    main()
