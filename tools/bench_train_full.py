#!/usr/bin/env python3
"""Full training step (backbone + head + matcher + criterion + backward) with the matcher's assignment on the device
(csrc/lsap.hip) vs the reference's host path (scipy): development tool for SURVEY 8 f-3, not the judged metric."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pctrans_amd.arch import maskformer as mfm  # noqa: E402
from pctrans_amd.arch.resnet import ResNet  # noqa: E402
from pctrans_amd.config import get_cfg  # noqa: E402
from test_arch_cpu import _blob  # noqa: E402

N, H, W, Q, G = 2, 448, 448, 100, 24


def build():
    """-> (model, step): MaskFormer (ResNet-50) in train mode on synthetic data and a closure running one SGD step."""
    torch.manual_seed(0)
    cfg = get_cfg(num_queries=Q, norm="BN", sem_norm="BN", dataset="CVPPP")
    model = mfm.MaskFormer(**mfm.MaskFormer.from_config(cfg, ResNet(50, 3, norm="BN"))).cuda().train()
    vol = torch.randn(N, 3, H, W, device="cuda")
    g = torch.Generator().manual_seed(1)
    targets = []
    for b in range(N):
        cy, cx = torch.randint(30, H - 30, (G,), generator=g), torch.randint(30, W - 30, (G,), generator=g)
        masks = torch.stack([_blob(H, W, int(y), int(x), 12) for y, x in zip(cy, cx)]).cuda()
        centers = torch.stack([cx.float() / W, cy.float() / H], -1).view(G, 1, 2).cuda()
        targets.append({"masks": masks, "labels": torch.ones(G, dtype=torch.long, device="cuda"),
                        "fg_masks": (masks.sum(0) > 0).float(), "center_points": centers})
    opt = torch.optim.SGD(model.parameters(), lr=1e-4)

    def step():
        losses = model(vol, targets, True)
        total = sum(v for v in losses.values() if torch.is_tensor(v))
        opt.zero_grad(set_to_none=True)
        total.backward()
        opt.step()
    return model, step


if __name__ == "__main__":
    model, step = build()
    for mode in (True, False, True, False):
        model.criterion.matcher.device_lsap = mode
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        print("train step N=%d %dx%d R50 Q=%d G=%d  matcher on %s: %.1f ms" % (
            N, H, W, Q, G, "device" if mode else "host (scipy)", (time.perf_counter() - t0) / 5 * 1e3))
