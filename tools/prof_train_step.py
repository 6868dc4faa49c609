#!/usr/bin/env python3
"""Host-side hot spots of one full training step (torch profiler; development tool)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_train_full import build  # noqa: E402

model, step = build()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages()
for e in sorted(ka, key=lambda e: -e.self_cpu_time_total)[:22]:
    print("%9.0f us cpu  %9.0f us dev  x%-5d %s" % (e.self_cpu_time_total, e.self_device_time_total, e.count, e.key[:70]))
print("total device time %.1f ms; launches %d" % (
    sum(e.self_device_time_total for e in ka) / 1e3,
    sum(e.count for e in ka if e.key.startswith(("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel")))))
