"""Sine positional encoding for images (2-D analogue of "Attention is all you need").

Mirrors transformer_decoder/position_encoding.py:12-52 of the reference (same constructor, same output layout:
channels = [y features | x features], each interleaving sin/cos over `num_pos_feats` frequencies).  The table
depends only on (H, W, mask) -- with no padding mask it is computed once per shape and cached per device instead of
being rebuilt with ~10 tiny kernels on each of the 6 (pixel decoder) + 3 (decoder) calls per forward.
"""
import math

import torch
from torch import nn


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def _table(self, not_mask):
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            y_embed = y_embed / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = x_embed / (x_embed[:, :, -1:] + eps) * self.scale
        k = torch.arange(self.num_pos_feats, dtype=torch.float32, device=not_mask.device)
        dim_t = self.temperature ** (2 * torch.div(k, 2, rounding_mode="floor") / self.num_pos_feats)

        def interleave(e):
            ang = e[:, :, :, None] / dim_t
            return torch.stack((ang[..., 0::2].sin(), ang[..., 1::2].cos()), dim=4).flatten(3)

        return torch.cat((interleave(y_embed), interleave(x_embed)), dim=3).permute(0, 3, 1, 2)

    def forward(self, x, mask=None):
        if mask is not None:
            return self._table(~mask)
        key = (x.size(2), x.size(3), x.device)
        pos = self._cache.get(key)
        if pos is None:
            ones = torch.ones((1, x.size(2), x.size(3)), device=x.device, dtype=torch.bool)
            pos = self._table(ones)
            if len(self._cache) > 16:
                self._cache.clear()
            self._cache[key] = pos
        return pos.expand(x.size(0), -1, -1, -1)

    def __repr__(self, _repr_indent=4):
        head = "Positional encoding " + self.__class__.__name__
        body = ["num_pos_feats: {}".format(self.num_pos_feats), "temperature: {}".format(self.temperature),
                "normalize: {}".format(self.normalize), "scale: {}".format(self.scale)]
        return "\n".join([head] + [" " * _repr_indent + line for line in body])
