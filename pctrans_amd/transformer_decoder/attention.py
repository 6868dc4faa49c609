"""Projection-free multi-head attention used by the PCTrans decoder.

Mirrors transformer_decoder/attention.py of the reference (a Conditional-DETR fork of nn.MultiheadAttention):
    class MultiheadAttention        :57-177   no in-projection; q/k width `embed_dim`, v width `vdim`; the only
                                              parameters are `out_proj.{weight,bias}` (Linear(vdim, vdim))
    multi_head_attention_forward    :180-387  q * head_dim^-0.5 -> [N*h, L, hd] -> q k^T -> bool mask = -inf /
                                              float mask added -> softmax -> (dropout) -> P v -> out_proj
The reference file does not import on torch >= 2 (attention.py:28 precedence bug), so this is a
restatement of the math above, not of its plumbing: the options the PCTrans decoder never exercises
(in_proj, bias_k/v, add_zero_attn, static k/v) are not carried over and raise if requested.

`need_weights` defaults to False here: the reference computes a head-averaged [N, L, S] weight tensor on every call
and every caller drops it (`[0]` at mask2former_transformer_decoder.py:92,177).  Ask for it explicitly to get it.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from .. import fused_ops
from ..layers import CachedLinear


def attention_core(q: Tensor, k: Tensor, v: Tensor, num_heads: int, attn_mask: Optional[Tensor] = None,
                   key_padding_mask: Optional[Tensor] = None, dropout_p: float = 0.0, training: bool = False,
                   need_weights: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
    """q [L, N, E], k [S, N, E], v [S, N, Ev] -> ([L, N, Ev], optional head-mean weights [N, L, S]).

    attn_mask: None | [L, S] | [N*h, L, S] | [N, 1|h, L, S]; bool (True = may not attend) or float (added).
    """
    L, N, E = q.shape
    S = k.shape[0]
    Ev = v.shape[2]
    hd, vhd = E // num_heads, Ev // num_heads
    assert hd * num_heads == E, "embed_dim must be divisible by num_heads"
    assert k.shape[1] == N and v.shape[0] == S and v.shape[1] == N

    if fused_ops.masked_attention_supported(q, k, v, num_heads, attn_mask, key_padding_mask, dropout_p, training,
                                            need_weights):
        # one MFMA kernel: scores, mask, online softmax and P.V never leave the registers
        return fused_ops.masked_attention(q, k, v, num_heads, attn_mask), None

    qh = (q * (float(hd) ** -0.5)).reshape(L, N, num_heads, hd).permute(1, 2, 0, 3)      # [N, h, L, hd]
    kh = k.reshape(S, N, num_heads, hd).permute(1, 2, 3, 0)                               # [N, h, hd, S]
    vh = v.reshape(S, N, num_heads, vhd).permute(1, 2, 0, 3)                              # [N, h, S, vhd]
    scores = torch.matmul(qh, kh)                                                         # [N, h, L, S]

    if attn_mask is not None:
        if attn_mask.dim() == 2:
            if tuple(attn_mask.shape) != (L, S):
                raise RuntimeError("The size of the 2D attn_mask is not correct.")
            m = attn_mask[None, None]
        elif attn_mask.dim() == 3:
            if tuple(attn_mask.shape) != (N * num_heads, L, S):
                raise RuntimeError("The size of the 3D attn_mask is not correct.")
            m = attn_mask.view(N, num_heads, L, S)
        elif attn_mask.dim() == 4:
            m = attn_mask
        else:
            raise RuntimeError("attn_mask's dimension {} is not supported".format(attn_mask.dim()))
        if m.dtype == torch.uint8:
            m = m.to(torch.bool)
        if m.dtype == torch.bool:
            scores = scores.masked_fill(m, float("-inf"))
        else:
            scores = scores + m
    if key_padding_mask is not None:
        assert tuple(key_padding_mask.shape) == (N, S)
        scores = scores.masked_fill(key_padding_mask.to(torch.bool)[:, None, None, :], float("-inf"))

    p = F.softmax(scores, dim=-1)
    if dropout_p > 0.0:
        p = F.dropout(p, p=dropout_p, training=training)
    out = torch.matmul(p, vh)                                                             # [N, h, L, vhd]
    out = out.permute(2, 0, 1, 3).reshape(L, N, Ev)
    return out, (p.sum(dim=1) / num_heads if need_weights else None)


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False,
                 kdim=None, vdim=None):
        super().__init__()
        if add_bias_kv or add_zero_attn or not bias:
            raise NotImplementedError("add_bias_kv / add_zero_attn / bias=False are never used by the PCTrans decoder")
        self.embed_dim = embed_dim
        self.kdim = kdim if kdim is not None else embed_dim
        self.vdim = vdim if vdim is not None else embed_dim
        self.num_heads = num_heads
        self.dropout = dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        self.out_proj = CachedLinear(self.vdim, self.vdim)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=False, attn_mask=None):
        if query.shape[2] != self.embed_dim:
            raise RuntimeError("query width %d != embed_dim %d" % (query.shape[2], self.embed_dim))
        out, w = attention_core(query, key, value, self.num_heads, attn_mask=attn_mask,
                                key_padding_mask=key_padding_mask, dropout_p=self.dropout, training=self.training,
                                need_weights=need_weights)
        return self.out_proj(out), w
