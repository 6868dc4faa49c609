#!/usr/bin/env python3
"""Which memory format should the semantic head's first 3x3 convolution get?  mask_features is a transposed VIEW of the encoder's
token rows (channels fastest, batch stride = all levels' rows): times seg_head on (a) that view cast to bf16, (b) a dense NCHW copy,
(c) a dense channels-last copy."""
import os, sys, torch
os.environ.setdefault("MIOPEN_FIND_MODE", "1")
os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pctrans_amd.transformer_decoder.mask2former_transformer_decoder import conv_with_kaiming_uniform
N, C, H, W, S = 128, 128, 128, 128, 21760
torch.manual_seed(0)
blk = conv_with_kaiming_uniform("BN", activation=True)
seg = torch.nn.Sequential(blk(C, C, kernel_size=3, stride=1), blk(C, C, kernel_size=3, stride=1)).cuda().eval()
tok = torch.randn(N, S, C, device="cuda")
view = tok[:, S - H * W:, :].transpose(1, 2).reshape(N, C, H, W)          # what the pixel decoder hands over
print("view strides", view.stride(), "contiguous", view.is_contiguous(), "channels_last", view.is_contiguous(memory_format=torch.channels_last))


def timeit(fn, iters=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    a = view.to(torch.bfloat16)
    b = view.to(torch.bfloat16).contiguous()
    c = view.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    print("cast view -> strides", a.stride())
    for name, x in (("cast of the view", a), ("dense NCHW", b), ("dense channels-last", c)):
        print("%-22s seg_head %.3f ms   (cast+copy itself: %.3f ms)" % (name, timeit(lambda: seg(x)), timeit(lambda: view.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)) if "last" in name else 0.0))
    ra, rb, rc = seg(a), seg(b), seg(c)
    print("max diff NCHW vs view %.4g, channels-last vs view %.4g, out strides %s %s %s" % (
        float((ra.float() - rb.float()).abs().max()), float((ra.float() - rc.float()).abs().max()), ra.stride(), rb.stride(), rc.stride()))
