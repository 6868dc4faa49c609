#!/usr/bin/env python3
"""Where do the launches of a TRAINING step come from?  Runs one step (forward + losses + backward) on the CPU under a
dispatch-mode counter and attributes every ATen call of the forward to the innermost pctrans_amd frame that issued it (a
device op is one or more kernel launches, so the ranking carries over to the GPU).  Development tool.
    python tools/count_ops.py [--queries 100] [--instances 24] [--size 128]"""
import argparse
import collections
import os
import random
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pctrans_amd.arch import maskformer as mfm  # noqa: E402
from pctrans_amd.arch.resnet import ResNet  # noqa: E402
from pctrans_amd.config import get_cfg  # noqa: E402
from pctrans_amd.pixel_decoder.ops.modules import ms_deform_attn as msda_mod  # noqa: E402
from test_arch_cpu import _blob  # noqa: E402


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_site = collections.Counter()
        self.total = 0

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.total += 1
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if "pctrans_amd" in fr.filename:
                site = "%s:%s" % (os.path.relpath(fr.filename, ROOT), fr.name)
                break
        self.by_site[site] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=100)
    ap.add_argument("--instances", type=int, default=24)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    msda_mod.allow_cpu_reference(True)
    torch.manual_seed(0)
    random.seed(0)
    cfg = get_cfg(num_queries=a.queries, norm="BN", sem_norm="BN", dataset="CVPPP")
    model = mfm.MaskFormer(**mfm.MaskFormer.from_config(cfg, ResNet(18, 3, norm="BN"))).train()
    H = W = a.size
    g = torch.Generator().manual_seed(1)
    vol = torch.randn(a.batch, 3, H, W, generator=g)
    targets = []
    for b in range(a.batch):
        cy = torch.randint(10, H - 10, (a.instances,), generator=g)
        cx = torch.randint(10, W - 10, (a.instances,), generator=g)
        masks = torch.stack([_blob(H, W, int(y), int(x), 5) for y, x in zip(cy, cx)])
        centers = torch.stack([cx.float() / W, cy.float() / H], -1).view(a.instances, 1, 2)
        targets.append({"masks": masks, "labels": torch.ones(a.instances, dtype=torch.long),
                        "fg_masks": (masks.sum(0) > 0).float(), "center_points": centers})
    # one uncounted step first: per-shape tables (position encodings, coordinate grids) are cached on the first call
    warm = model(vol, targets, True)
    sum(v for v in warm.values() if torch.is_tensor(v)).backward()
    model.zero_grad(set_to_none=True)
    with Counter() as c:
        losses = model(vol, targets, True)
        total = sum(v for v in losses.values() if torch.is_tensor(v))
    fwd = c.total
    with Counter() as cb:
        total.backward()
    print("forward + losses: %d ATen calls; backward: %d" % (fwd, cb.total))
    for site, n in c.by_site.most_common(40):
        print("  %6d  %s" % (n, site))


if __name__ == "__main__":
    main()
