#!/usr/bin/env python3
"""Probe: does running two half-batches of the head on two HIP streams (one half's MSDeformAttn / mask-head kernels beside
the other half's GEMMs) beat one full batch?  Development tool."""
import os
import sys
import time

os.environ.setdefault("MIOPEN_FIND_MODE", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
head, shapes = bench.build_head(args, dev)
feats = bench.synth_features(shapes, args.batch, args.image, dev, seed=1234)
bench.model_like_offsets(head, feats)
amp = torch.autocast("cuda", dtype=torch.bfloat16)
halves = [{k: v[:args.batch // 2].contiguous() for k, v in feats.items()}, {k: v[args.batch // 2:].contiguous() for k, v in feats.items()}]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def full():
    with torch.no_grad(), amp:
        head(feats)


def split_serial():
    with torch.no_grad(), amp:
        head(halves[0])
        head(halves[1])


def split_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.no_grad(), amp:
        with torch.cuda.stream(s1):
            head(halves[0])
        with torch.cuda.stream(s2):
            head(halves[1])
    cur.wait_stream(s1)
    cur.wait_stream(s2)


for name, fn in (("full batch, one stream", full), ("two halves, one stream", split_serial), ("two halves, two streams", split_streams),
                 ("full batch, one stream", full), ("two halves, two streams", split_streams)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    print("%-28s %.1f ms per %d images = %.0f samples/s" % (name, ms, args.batch, args.batch / ms * 1e3), flush=True)
