#!/usr/bin/env python3
"""Which torch operator (and input shapes) launches each of the heaviest kernels of the bench step: torch profiler with
shapes on ONE step after warm-up (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import ProfilerActivity, profile


class A:
    image, queries, levels, dtype, batch = 512, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 4, "bf16", 128


dev = torch.device("cuda", 0)
head, shapes = bench.build_head(A, dev)
feats = bench.synth_features(shapes, A.batch, A.image, dev, 1234)
bench.model_like_offsets(head, feats)
amp = torch.autocast("cuda", dtype=torch.bfloat16)


def step():
    with torch.no_grad(), amp:
        return head(feats)


for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "device_time_total", None)
    if dt is None:
        dt = e.cuda_time_total
    sdt = getattr(e, "self_device_time_total", None)
    if sdt is None:
        sdt = e.self_cuda_time_total
    if sdt > 0:
        rows.append((sdt, e.key, e.count, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("sum of self device time: %.2f ms" % (tot / 1e3))
for sdt, key, cnt, shp in rows[:28]:
    print("%7.3f ms  x%-4d %-60s %s" % (sdt / 1e3, cnt, key[:60], shp))
