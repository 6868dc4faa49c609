#!/usr/bin/env python3
"""The head forward over batch 128 as one call against two half-batches on two streams (development tool: is there anything
to gain from overlapping the step's small kernels and tails with its large ones?)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


class A:
    image, queries, levels, dtype, batch = 512, 100, 4, "bf16", 128


dev = torch.device("cuda", 0)
head, shapes = bench.build_head(A, dev)
feats = bench.synth_features(shapes, A.batch, A.image, dev, 1234)
bench.model_like_offsets(head, feats)
half = [{k: v[i * 64:(i + 1) * 64].contiguous() for k, v in feats.items()} for i in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def one():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return head(feats)[0]["pred_masks"]


def two():
    outs = []
    cur = torch.cuda.current_stream()
    for s, f in zip(streams, half):
        s.wait_stream(cur)
        with torch.cuda.stream(s), torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            outs.append(head(f)[0]["pred_masks"])
    for s in streams:
        cur.wait_stream(s)
    return outs


def timed(fn, n=10):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(2):
    print("one call %.2f ms   two half-batches on two streams %.2f ms" % (timed(one), timed(two)))
