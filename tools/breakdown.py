#!/usr/bin/env python3
"""Stage-level timing of one head forward (HIP events): pixel decoder / transformer decoder, and per encoder layer."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class A:
    batch, image, queries, levels, dtype = 16, 512, 100, 4, "bf16"


args = A()
dev = torch.device("cuda", 0)
head, shapes = bench.build_head(args, dev)
feats = bench.synth_features(shapes, args.batch, args.image, dev, 1)
amp = torch.autocast("cuda", dtype=torch.bfloat16)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad(), amp:
    mf, enc, ms = head.pixel_decoder.forward_features(feats)
    t_pd = timed(lambda: head.pixel_decoder.forward_features(feats))
    t_dec = timed(lambda: head.predictor(ms, None, mf, None, 0.5, None))
    t_all = timed(lambda: head(feats))
    print("pixel decoder %.2f ms   transformer decoder %.2f ms   head %.2f ms" % (t_pd, t_dec, t_all))
    # one encoder layer
    pd = head.pixel_decoder
    srcs = [pd.input_proj[i](feats[f].float()) for i, f in enumerate(pd.transformer_in_features[::-1])]
    pos = [pd.pe_layer(s) for s in srcs]
    t_proj = timed(lambda: [pd.input_proj[i](feats[f].float()) for i, f in enumerate(pd.transformer_in_features[::-1])])
    tr = pd.transformer
    shapes_l = [(int(s.shape[2]), int(s.shape[3])) for s in srcs]
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    lvlpos = torch.cat([p.flatten(2).transpose(1, 2) + tr.level_embed[i].view(1, 1, -1) for i, p in enumerate(pos)], 1)
    ss, st, ref = tr._geometry(shapes_l, src.device)
    refb = ref.expand(src.shape[0], -1, -1, -1)
    layer = tr.encoder.layers[0]
    with torch.autocast("cuda", enabled=False):
        t_layer = timed(lambda: layer(src, lvlpos, refb, ss, st, None))
        t_attn = timed(lambda: layer.self_attn(src + lvlpos, refb, src, ss, st, None))
        t_ffn = timed(lambda: layer.forward_ffn(src))
    print("input_proj %.2f ms  encoder layer %.2f ms (self_attn module %.2f, ffn %.2f)" % (t_proj, t_layer, t_attn, t_ffn))
