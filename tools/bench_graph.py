#!/usr/bin/env python3
"""Small-batch latency of the head forward, eager vs captured in a HIP graph (development tool)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class A:
    image, queries, levels, dtype = 512, 100, 4, "bf16"


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


args = A()
dev = torch.device("cuda", 0)
for batch in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("1", "2", "8"))]:
    args.batch = batch
    head, shapes = bench.build_head(args, dev)
    feats = bench.synth_features(shapes, batch, args.image, dev, 1)

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return head(feats)[0]["pred_masks"]

    t_eager = timed(fwd)
    ref = fwd().clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fwd()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fwd()
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    t_graph = timed(g.replay)
    print("batch %d: eager %.2f ms  graph %.2f ms  identical=%s" % (batch, t_eager, t_graph, same))
