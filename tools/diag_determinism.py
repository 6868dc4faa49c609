#!/usr/bin/env python3
"""Diagnostic (GPU box): which sub-modules of the head differ between two EAGER forwards on the same input at the geometry
where a HIP-graph replay once differed from eager (batch 1, 256^2, 20 queries, bf16 autocast), and how a replay compares.
Prints one line per repetition.  The same comparison is a test (tests/test_head_gpu.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MIOPEN_FIND_MODE", "1")
from test_head_gpu import _feats, _head, _record_all_modules  # noqa: E402
from pctrans_amd.graph import GraphedForward  # noqa: E402

for batch in (1, 2):
    head, shapes = _head(4, Q=20)
    feats = _feats(shapes, batch, 256, 256, seed=5)
    for rep in range(3):
        a = _record_all_modules(head, feats, torch.bfloat16)
        b = _record_all_modules(head, feats, torch.bfloat16)
        diff = [n for n in a if any(not torch.equal(x, y) for x, y in zip(a[n], b[n]))]
        print("batch %d rep %d: %d modules, differing between two eager runs: %s" % (batch, rep, len(a), diff), flush=True)
    fwd = GraphedForward(head, feats, autocast_dtype=torch.bfloat16)
    for seed in (6, 7, 8):
        other = _feats(shapes, batch, 256, 256, seed=seed)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            want, want_mf = head(other)
        got, got_mf = fwd(other)
        print("batch %d seed %d: replay == eager: mask_features %s, pred_masks %s (max |d| %g)" % (
            batch, seed, torch.equal(got_mf, want_mf), torch.equal(got["pred_masks"], want["pred_masks"]),
            float((got["pred_masks"].float() - want["pred_masks"].float()).abs().max())), flush=True)
