"""Deterministic parameter values shared by tests/golden/make_golden_decoder.py (which fills the REFERENCE modules with
them) and the tests (which fill this package's modules): a function of the parameter's name, shape and a seed, so that
large state dicts need not be stored in the fixtures.  numpy's MT19937 stream is platform-independent."""
import zlib

import numpy as np
import torch


def deterministic_fill(module, seed):
    """Matrices ~ N(0, 2/(fan_in+fan_out)); norm weights ~ 1 + 0.2 N; other vectors ~ 0.05 N; embeddings ~ 0.5 N."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            rs = np.random.RandomState((zlib.crc32(name.encode()) + seed) % (2 ** 31))
            x = rs.standard_normal(tuple(p.shape))
            if p.dim() >= 2 and ("query_feat" in name or "query_embed" in name or "level_embed" in name):
                x = 0.5 * x
            elif p.dim() >= 2:
                fan_out, fan_in = p.shape[0], int(np.prod(p.shape[1:]))
                x = x * np.sqrt(2.0 / (fan_in + fan_out))
            elif "norm" in name and name.endswith("weight"):
                x = 1.0 + 0.2 * x
            else:
                x = 0.05 * x
            p.copy_(torch.from_numpy(x.astype(np.float32)))
    return module


def fill_pixel_decoder(module, seed):
    """deterministic_fill + GroupNorm scales of the input projections around 1 (their names carry no "norm"; left at
    0.05 N they would flatten the encoder's input)."""
    deterministic_fill(module, seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.startswith("input_proj.") and name.endswith(".1.weight"):
                rs = np.random.RandomState((zlib.crc32(name.encode()) + seed + 7) % (2 ** 31))
                p.copy_(torch.from_numpy((1.0 + 0.2 * rs.standard_normal(tuple(p.shape))).astype(np.float32)))
    return module
