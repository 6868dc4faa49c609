"""Seeded input builders shared by CPU and GPU tests (numpy only, no torch RNG so CPU/GPU boxes agree)."""
import numpy as np


def starts_of(shapes):
    shapes = np.asarray(shapes, dtype=np.int64)
    hw = shapes[:, 0] * shapes[:, 1]
    return np.concatenate([[0], np.cumsum(hw)[:-1]]).astype(np.int64)


def pixel_centres(shapes):
    """[S, 2] (x, y) normalised pixel centres of every level, flattened level after level (msdeformattn.py:141-153)."""
    ref = []
    for (h, wd) in np.asarray(shapes, dtype=np.int64):
        ys, xs = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(wd) + 0.5) / wd, indexing="ij")
        ref.append(np.stack([xs.ravel(), ys.ravel()], -1))
    return np.concatenate(ref, 0)


def init_like_offsets(M, P):
    """[M, P, 2] the module's initial sampling_offsets bias (ops/modules/ms_deform_attn.py:66-75): head m looks along
    direction m * 2pi / M, point p sits (p + 1) pixels out (max-norm) -- what a random-init network produces."""
    th = np.arange(M, dtype=np.float32) * (2.0 * np.pi / M)
    g = np.stack([np.cos(th), np.sin(th)], -1)
    g = g / np.abs(g).max(-1, keepdims=True)
    return g[:, None, :] * np.arange(1, P + 1, dtype=np.float32)[None, :, None]


def make_case(seed, N, M, D, Lq, P, shapes, dtype=np.float32, lo=0.0, hi=1.0, model_like=False, px_sigma=2.0,
              init_like=False):
    """value ~ N(0,1); weights normalised over L*P; locations uniform in [lo,hi)^2 (as OPS/test.py:37) or
    `model_like`: pixel-centre reference points of the query's own position (msdeformattn.py:141-153) plus
    N(0, px_sigma px) offsets (distribution M).  With model_like, Lq must equal S."""
    rng = np.random.RandomState(seed)
    shapes = np.asarray(shapes, dtype=np.int64)
    L = shapes.shape[0]
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = rng.standard_normal((N, S, M, D)).astype(dtype)
    w = rng.random_sample((N, Lq, M, L, P)) + 1e-5
    w = (w / w.sum((-1, -2), keepdims=True)).astype(dtype)
    if model_like or init_like:
        assert Lq == S
        ref = pixel_centres(shapes)                                    # [S, 2] (x, y)
        if init_like:   # distribution I: directional offsets of 1..P px per head (+ a little jitter when px_sigma > 0)
            off = np.broadcast_to(init_like_offsets(M, P)[None, None, :, None, :, :], (N, Lq, M, L, P, 2)).astype(np.float64)
            off = off + rng.standard_normal((N, Lq, M, L, P, 2)) * (px_sigma if model_like else 0.0)
        else:
            off = rng.standard_normal((N, Lq, M, L, P, 2)) * px_sigma
        norm = np.stack([shapes[:, 1], shapes[:, 0]], -1).astype(np.float64)   # (W, H)
        loc = ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
        loc = loc.astype(dtype)
    else:
        loc = (rng.random_sample((N, Lq, M, L, P, 2)) * (hi - lo) + lo).astype(dtype)
    return dict(value=value, shapes=shapes, starts=starts_of(shapes), loc=loc, attn=w)
