#!/usr/bin/env python3
"""Diagnostic for the one-off ~350-440 ms stall bench.py saw inside its timed region: does it follow the start of event
recording (fresh timing events, ~80 per step) rather than the time under load?  Phase 1: N1 steps without any event; phase 2:
N2 steps with fresh events around every hook-timed launch; phase 3: the same with events reserved in advance."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pctrans_amd import MultiScaleDeformableAttention as MSDA, _timing


class A:
    image, queries, levels, dtype, batch = 512, 100, 4, "bf16", 128


dev = torch.device("cuda", 0)
head, shapes = bench.build_head(A, dev)
feats = bench.synth_features(shapes, A.batch, A.image, dev, 1234)
bench.model_like_offsets(head, feats)


def step():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return head(feats)


def run(n, label):
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for m in marks:
        m.record()
    torch.cuda.synchronize()
    marks[0].record()
    for i in range(n):
        step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(n)]
    print("%-34s max %.1f  median %.1f  steps>1.5x median: %s" % (
        label, max(ms), sorted(ms)[n // 2], [(i, round(x)) for i, x in enumerate(ms) if x > 1.5 * sorted(ms)[n // 2]]), flush=True)


import gc
for _ in range(2):
    step()
gc.collect(); gc.disable()
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 20
run(n1, "no events (%d steps)" % n1)
MSDA.kernel_timing(True)
run(20, "fresh events")
rec = MSDA.kernel_timing(False)
print("   launches timed per step:", len(rec) // 20)
run(10, "no events again")
_timing.reserve(2 * (len(rec) // 20) * 21 + 16, dev)
MSDA.kernel_timing(True)
run(20, "reserved events")
MSDA.kernel_timing(False)
