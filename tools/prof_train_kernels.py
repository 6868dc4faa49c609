#!/usr/bin/env python3
"""Full kernel names of one training step of configs[3] (torch profiler; development tool): which element-wise functors
the 'vectorized_elementwise_kernel' time of tools/record_train_configs.py is made of, with input shapes of the top ATen ops."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import record_train_configs as R  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "configs[3] BBBC-PCTrans.yaml"
model, step = R.build(R.CONFIGS[name])
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
top = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        t = e.device_time if hasattr(e, "device_time") else e.cuda_time
        k = e.name[:260]
        c = top.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += t
for k, (n, t) in sorted(top.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%8.2f ms %5d  %s" % (t / 1e3, n, k))
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=40,
                                                          max_shapes_column_width=70))
