#!/usr/bin/env python3
"""Diagnostic: per-part cycle sums of one wave of the one-launch mask head (library built with -DFZ_STAMP; its attention
mask is overwritten by the stamps, timing only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pctrans_amd import dynamic_mask_head as dmh
N, Q, H, W = 128, 100, 128, 128
g = torch.Generator(device="cuda").manual_seed(0)
mf = torch.randn(N, 16, H, W, device="cuda", generator=g)
ref = torch.rand(N, Q, 2, device="cuda", generator=g)
prm = torch.randn(N, Q, 233, device="cuda", generator=g) * 0.2
for _ in range(3):
    up, am = dmh.dynamic_mask_head_forward(mf, ref, prm, 4, True, (32, 32), out_dtype=torch.bfloat16, kernel="fused")
torch.cuda.synchronize()
st = am.view(torch.uint8).flatten()[:64].clone().view(torch.int64).tolist()
names = ["prepare + next fetch", "tiles (MLP)", "pack + LDS write", "barrier", "LDS read + window", "blend + store", "mask", "-"]
tot = sum(st[:7])
for n_, v in zip(names, st):
    print("%-22s %10d cycles  %5.1f %%  (%d per pair)" % (n_, v, 100.0 * v / max(tot, 1), v // 50))
