#!/usr/bin/env python3
"""Time the device matcher's assignment kernel (csrc/lsap.hip) at the training configurations' sizes: 2 problems per launch
(the per-GPU batch), 100 x 20 (configs[2]) and 300 x 60 (configs[3]); ms per launch (10 launches per training step)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pctrans_amd import fused_ops

for Q, G, B in ((100, 20, 2), (300, 60, 2), (300, 60, 16), (1000, 400, 2)):
    rng = np.random.RandomState(Q)
    cost = torch.from_numpy((rng.standard_normal((B, Q, G)) * 3).astype(np.float32)).cuda()
    cnt = torch.full((B,), G, dtype=torch.int32)
    for _ in range(3):
        fused_ops.lsap(cost, cnt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fused_ops.lsap(cost, cnt)
    e1.record()
    torch.cuda.synchronize()
    print("Q=%d G=%d problems=%d: %.3f ms per launch" % (Q, G, B, e0.elapsed_time(e1) / 20))
