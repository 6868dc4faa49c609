#!/usr/bin/env python3
"""Diagnostic: does the step time of bench.py's workload depend on how long the process has been running?  Runs the bench
step in back-to-back blocks (no synchronisation inside a block) and prints ms per step per block, plus the clocks rocm-smi
reports between blocks.  First GPU process on a fresh box vs. a second process."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


class A:
    image, queries, levels, dtype, batch = 512, 100, 4, "bf16", 128


dev = torch.device("cuda", 0)
head, shapes = bench.build_head(A, dev)
feats = bench.synth_features(shapes, A.batch, A.image, dev, 1234)
bench.model_like_offsets(head, feats)
amp = torch.autocast("cuda", dtype=torch.bfloat16)


def step():
    with torch.no_grad(), amp:
        return head(feats)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "mclk" in l or "fclk" in l or "Power" in l or "junction" in l.lower()) and "[0]" in l]
        return " | ".join(k.split(":", 1)[-1].strip() for k in keep)
    except Exception as e:  # noqa: BLE001
        return "rocm-smi: %r" % (e,)


blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t_start = time.perf_counter()
for b in range(blocks):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(per):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("block %2d  t=%6.1fs  %.2f ms/step (host enqueue %.2f ms/step)  %s" % (
        b, time.perf_counter() - t_start, 1e3 * dt / per, 1e3 * t_host / per, smi() if b % 3 == 0 else ""), flush=True)
