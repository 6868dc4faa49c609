#!/usr/bin/env python3
"""Top aten ops of one head forward by device time, grouped by input shape (development tool)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class A:
    batch, image, queries, levels, dtype = 64, 512, 100, 4, "bf16"


args = A()
dev = torch.device("cuda", 0)
head, shapes = bench.build_head(args, dev)
feats = bench.synth_features(shapes, args.batch, args.image, dev, 1)


def fwd():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return head(feats)


for _ in range(3):
    fwd()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    fwd()
    torch.cuda.synchronize()
rows = [(e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:100])
        for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 60]
rows.sort(reverse=True)
for r in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 40]:
    print("%9.0f us  x%-4d %-34s %s" % r)
