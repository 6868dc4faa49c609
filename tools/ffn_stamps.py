#!/usr/bin/env python3
"""Phase times of the fused FFN kernel from a -DPCT_FFN_STAMPS=1 build (tools/variant_one.sh ffn_st ffn_fused_split -DPCT_FFN_STAMPS=1;
run with the variant copied over the product library, e.g. through tools/ab_libs_cmd.sh).  s_memtime ticks at 100 MHz."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pctrans_amd import fused_ops, _lib

rows = 128 * 21760
with torch.no_grad():
    x = torch.randn(rows, 128, device="cuda")
    lin1 = torch.nn.Linear(128, 1024).cuda()
    lin2 = torch.nn.Linear(1024, 128).cuda()
    norm = torch.nn.LayerNorm(128).cuda()
    for _ in range(3):
        fused_ops.ffn_layer_norm(x, lin1, lin2, norm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fused_ops.ffn_layer_norm(x, lin1, lin2, norm)
    e1.record()
    torch.cuda.synchronize()
    print("kernel + weight split %.3f ms" % e0.elapsed_time(e1))
    buf = (ctypes.c_ulonglong * (256 * 8))()
    rc = _lib.lib().pct_ffn_stamps_read(buf)
    assert rc == 0, rc
    names = ["prologue", "x load+split", "GEMM1(0)+barrier", "phase A", "phase B 0-5", "barrier", "phase B 6-7", "epilogue"]
    tot = [0] * 8
    for b in range(256):
        for i in range(8):
            tot[i] += buf[b * 8 + i]
    s = sum(tot)
    for i in range(8):
        print("%-18s %10.1f ticks / workgroup  %5.1f %%" % (names[i], tot[i] / 256, 100.0 * tot[i] / s))
    print("sum %.1f ticks/workgroup = %.3f ms at 100 MHz" % (s / 256, s / 256 / 1e5))
