#!/usr/bin/env python3
"""Training-step timing of the head alone (development tool, not the judged metric): forward + backward of
MaskFormerHead on synthetic ResNet-50-shaped features at the reference's per-GPU training batch (2 x 448^2 crops,
CVPPP config 3: 3 encoder levels) with a surrogate scalar loss, fp32.  Shows where the HIP backward kernel stands:
PCT_MSDA_BWD_KERNEL=generic reproduces the per-sample global-atomic scatter the reference's CUDA kernel uses."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pctrans_amd.config import get_cfg, resnet_output_shape  # noqa: E402
from pctrans_amd.meta_arch.mask_former_head import MaskFormerHead  # noqa: E402
from pctrans_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402

N = int(os.environ.get("N", "2"))
IMG = int(os.environ.get("IMG", "448"))
LEVELS = int(os.environ.get("LEVELS", "3"))
feats_names = ("res2", "res3", "res4", "res5")[4 - LEVELS:]
cfg = get_cfg(num_queries=100, enc_in_features=feats_names)
shapes = resnet_output_shape(50)
torch.manual_seed(0)
head = MaskFormerHead(**MaskFormerHead.from_config(cfg, shapes)).cuda().train()
g = torch.Generator(device="cuda").manual_seed(1)
feats = {k: torch.randn(N, s.channels, IMG // s.stride, IMG // s.stride, device="cuda", generator=g, requires_grad=True)
         for k, s in shapes.items()}


def step():
    pred, mf = head(feats)
    loss = pred["pred_masks"].float().square().mean() + sum(a["pred_masks"].float().mean() for a in pred["aux_outputs"])
    loss.backward()
    head.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
MSDA.kernel_timing(True)
t0 = time.perf_counter()
K = 10
for _ in range(K):
    step()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / K
rec = MSDA.kernel_timing(False)
fwd = [ms for n, ms in rec if n == "forward"]
bwd = [ms for n, ms in rec if n == "backward"]
print("train step N=%d %dx%d L=%d: %.1f ms  | MSDeformAttn forward %.3f ms x %d, backward %.3f ms x %d per step (%s)" % (
    N, IMG, IMG, LEVELS, el * 1e3, sum(fwd) / max(1, len(fwd)), len(fwd) // K, sum(bwd) / max(1, len(bwd)), len(bwd) // K,
    os.environ.get("PCT_MSDA_BWD_KERNEL", "auto")))
