"""PCTrans transformer decoder: position-guided masked cross-attention, self-attention, FFN, iterative reference-point
refinement and a CondInst-style dynamic mask head with relative coordinates.

API / state-dict mirror of transformer_decoder/mask2former_transformer_decoder.py of the reference:
    gen_sineembed_for_position   :21-39     128 freqs per axis, temperature 20, output order [y | x]
    inverse_sigmoid              :41-45     eps = 1e-3
    SelfAttentionLayer           :47-103    5 Linears + projection-free MHA + LN
    CrossAttentionLayer          :105-193   content / position projections, per-head concat [content | position]
    FFNLayer, MLP                :195-261
    MultiScaleMaskedTransformerDecoder  :264-754   ctor kwargs, forward(x, targets, mask_features, mask,
                                            attn_mask_threshold, criterion) -> dict, parameter names
    compute_locations, parse_dynamic_params    :929-979
    select_pos_neg_query / select_pos_neg_mask / dice_for   :800-927  (training only; see query_contrast.py)

Same math, different execution:
  * dynamic mask head: the reference materialises, per call, a [1, N*Q*(mask_dim+2), H, W] input by gathering
    mask features once per query and concatenating per-query relative coordinates (:672-676), then runs three grouped
    convolutions with N*Q groups.  Here layer 0 is  W_feat @ F  (one batched GEMM over the shared feature map) plus a
    rank-2 relative-coordinate term, layers 1-2 are batched [8x8] / [1x8] products; nothing of size N*Q*18*H*W is built
    and no `.tolist()` host round trip happens (:664).
  * the boolean attention mask is kept as [N, 1, Q, HW] and broadcast over heads instead of being repeated 8x
    (:689-691), and the "fully masked row attends everywhere" rule (:561) is a mask AND instead of a
    boolean-index scatter (which forces a device sync).
  * K/V/positional projections, sine tables and level embeddings per level are computed once per forward.
"""
import logging
import math
from typing import Optional

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from .. import dynamic_mask_head as dmh
from .. import fused_ops
from ..layers import CachedLinear, Conv2d, c2_xavier_fill, get_norm
from .attention import MultiheadAttention
from .position_encoding import PositionEmbeddingSine


_SINE_TABLES = {}


_HEAD_TABLES = {}


def gen_sineembed_for_position(pos_tensor, temperature=20):
    """[Q, N, 2k] normalised (x, y) points -> [Q, N, 256k] sine embedding, (y, x) order per point.

    Element i of each 128-wide block is sin(c * 2pi / T^(2(i//2)/128)) for even i and cos(...) for odd i
    (mask2former_transformer_decoder.py:21-39); cos(a) is evaluated as sin(a + pi/2) so that the whole embedding is one
    fused multiply-add and one sin instead of ~14 small kernels (difference <= 1 ulp of the argument, ~5e-7)."""
    key = (pos_tensor.device, temperature)
    tab = _SINE_TABLES.get(key)
    if tab is None:
        i = torch.arange(128, dtype=torch.float32, device=pos_tensor.device)
        dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / 128)
        tab = _SINE_TABLES[key] = ((2 * math.pi) / dim_t, (i % 2) * (math.pi / 2))
    freq, phase = tab
    k = pos_tensor.shape[-1] // 2
    Q, N = pos_tensor.shape[:2]
    yx = pos_tensor.reshape(Q, N, k, 2).flip(-1)                        # (y, x) per point
    return torch.sin(torch.addcmul(phase, yx.unsqueeze(-1).float(), freq)).reshape(Q, N, k * 256)


def inverse_sigmoid(x, eps=1e-3):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def _get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError(F"activation should be relu/gelu, not {activation}.")


def _lp(x):
    """Under autocast every nn.Linear casts its fp32 input to the autocast dtype; when the same tensor feeds several
    Linears do that cast once (identical values, fewer launches).  No-op outside autocast."""
    if x is not None and x.is_cuda and x.dtype == torch.float32 and torch.is_autocast_enabled("cuda"):
        return x.to(torch.get_autocast_dtype("cuda"))
    return x


def _xavier_all(module):
    for p in module.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.sa_qcontent_proj = CachedLinear(d_model, d_model)
        self.sa_qpos_proj = CachedLinear(d_model, d_model)
        self.sa_kcontent_proj = CachedLinear(d_model, d_model)
        self.sa_kpos_proj = CachedLinear(d_model, d_model)
        self.sa_v_proj = CachedLinear(d_model, d_model)
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout, vdim=d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        _xavier_all(self)

    def forward_post(self, tgt, tgt_mask: Optional[Tensor] = None, tgt_key_padding_mask: Optional[Tensor] = None,
                     query_pos: Optional[Tensor] = None):
        t, qp = _lp(tgt), _lp(query_pos)
        q = self.sa_qcontent_proj(t) + self.sa_qpos_proj(qp)
        k = self.sa_kcontent_proj(t) + self.sa_kpos_proj(qp)
        v = self.sa_v_proj(t)
        tgt2 = self.self_attn(q, k, value=v, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)[0]
        return self.norm1(tgt + self.dropout1(tgt2))

    forward = forward_post


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False, points_num=1):
        super().__init__()
        self.ca_qcontent_proj = CachedLinear(d_model, d_model)
        self.ca_qpos_proj = CachedLinear(d_model, d_model)
        self.ca_kcontent_proj = CachedLinear(d_model, d_model)
        self.ca_kpos_proj = CachedLinear(d_model, d_model)
        self.ca_v_proj = CachedLinear(d_model, d_model)
        self.ca_qpos_sine_proj = CachedLinear(d_model * 2 * points_num, d_model)
        self.cross_attn = MultiheadAttention(d_model * 2, nhead, dropout=dropout, vdim=d_model)
        self.nhead = nhead
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout2 = nn.Dropout(dropout)
        _xavier_all(self)

    def forward_post(self, tgt, memory, memory_mask: Optional[Tensor] = None,
                     memory_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None,
                     query_pos: Optional[Tensor] = None, query_sine_embed=None, is_first=False,
                     memory_row_open: Optional[Tensor] = None):
        """`memory_row_open` (bool [N, 1, Q, 1] or None): queries whose row of `memory_mask` is to be ignored (the decoder's
        "a query whose mask rules out every pixel attends everywhere", :561).  The split-operand kernel takes it as it is;
        every other path folds it into the mask first."""
        Q, bs, C = tgt.shape
        hw = memory.shape[0]
        h, hd = self.nhead, C // self.nhead

        q = self.ca_qcontent_proj(tgt)
        k = self.ca_kcontent_proj(memory)
        k_pos = self.ca_kpos_proj(pos)
        fused = (memory_key_padding_mask is None and k.is_cuda and k.dtype == torch.bfloat16
                 and not (torch.is_grad_enabled() and (
                     tgt.requires_grad or memory.requires_grad or q.requires_grad or k.requires_grad
                     or k_pos.requires_grad or self.ca_v_proj.weight.requires_grad
                     or self.ca_qpos_sine_proj.weight.requires_grad or self.ca_qpos_proj.weight.requires_grad))
                 and C // h == 16 and self.cross_attn.dropout == 0.0
                 and (memory_mask is None or (memory_mask.dtype == torch.bool
                                              and tuple(memory_mask.shape) == (bs, 1, Q, hw))))
        # the split-operand kernel takes the projections' outputs as they are: no per-head concatenation, no V^T
        split = fused and fused_ops.cross_attention_supported(q, k, k_pos, h, memory_mask)
        if memory_row_open is not None and memory_mask is not None and not split:
            memory_mask = memory_mask & ~memory_row_open
            memory_row_open = None
        if fused and not split:
            # V^T [N, C, HW] straight out of the projection GEMM (W . memory^T + b): the generic MFMA attention kernel reads
            # value rows per channel, so no [HW, N, C] -> [N, C, HW] transpose pass is needed
            w = self.ca_v_proj.weight
            v_t = torch.baddbmm(self.ca_v_proj.bias.view(1, C, 1), w.unsqueeze(0).expand(bs, C, C),
                                memory.permute(1, 2, 0))
            v = None
        else:
            v = self.ca_v_proj(memory)
        if is_first:          # first layer: the learned query position also enters the content half (:150-156)
            q_pos = self.ca_qpos_proj(query_pos)
            q = q + q_pos
            k = k + k_pos
        if query_sine_embed is not None:
            q_side = self.ca_qpos_sine_proj(query_sine_embed)
        else:
            q_side = q_pos
        if split and q.dtype == torch.bfloat16 and q_side.dtype == torch.bfloat16 and v.dtype == torch.bfloat16:
            # per head [content (hd) | position (hd)] on both sides (:160-172), formed inside the kernel
            core = fused_ops.cross_attention(q, q_side, k, k_pos, v, h, memory_mask, row_open=memory_row_open)
            return self.norm2(tgt + self.dropout2(self.cross_attn.out_proj(core)))
        if memory_row_open is not None and memory_mask is not None:
            memory_mask = memory_mask & ~memory_row_open
        # per head: [content (hd) | position (hd)]
        q = torch.cat([q.view(Q, bs, h, hd), q_side.view(Q, bs, h, hd)], dim=3).view(Q, bs, 2 * C)
        k = torch.cat([k.view(hw, bs, h, hd), k_pos.view(hw, bs, h, hd)], dim=3).view(hw, bs, 2 * C)

        if fused and not split and q.dtype == torch.bfloat16 and v_t.dtype == torch.bfloat16:
            core = fused_ops.masked_attention(q, k, None, h, memory_mask, v_t=v_t)
            tgt2 = self.cross_attn.out_proj(core)
        else:
            if v is None:
                v = v_t.permute(2, 0, 1)
            tgt2 = self.cross_attn(query=q, key=k, value=v, attn_mask=memory_mask,
                                   key_padding_mask=memory_key_padding_mask)[0]
        return self.norm2(tgt + self.dropout2(tgt2))

    forward = forward_post


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.linear1 = CachedLinear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = CachedLinear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before
        _xavier_all(self)

    def with_pos_embed(self, tensor, pos: Optional[Tensor]):
        return tensor if pos is None else tensor + pos

    def _hidden(self, x):
        if self.activation is F.relu:
            return self.linear1(x, relu=True)                       # bias + ReLU in the GEMM epilogue under autocast
        return self.activation(self.linear1(x))

    def forward_post(self, tgt):
        tgt2 = self.linear2(self.dropout(self._hidden(tgt)))
        return self.norm(tgt + self.dropout(tgt2))

    def forward_pre(self, tgt):
        tgt2 = self.linear2(self.dropout(self._hidden(self.norm(tgt))))
        return tgt + self.dropout(tgt2)

    def forward(self, tgt):
        return self.forward_pre(tgt) if self.normalize_before else self.forward_post(tgt)


class MLP(nn.Module):
    """Linear -> ReLU -> ... -> Linear."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(CachedLinear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = layer(x, relu=True) if i < self.num_layers - 1 else layer(x)
        return x


def _init_mlp(mlp):
    for lin in mlp.layers:
        nn.init.xavier_uniform_(lin.weight)
        nn.init.zeros_(lin.bias)


def conv_with_kaiming_uniform(norm=None, activation=None):
    """conv (+ norm) (+ ReLU) factory as transformer_decoder/conv_with_kaiming_uniform.py:125-169 builds for
    `seg_head` (keys `seg_head.{i}.0.weight`, `seg_head.{i}.1.*`); deformable / separable variants are never
    instantiated by PCTrans and are not carried."""
    def make_conv(in_channels, out_channels, kernel_size, stride=1, dilation=1):
        conv = Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                      padding=dilation * (kernel_size - 1) // 2, dilation=dilation, bias=(norm is None))
        nn.init.kaiming_uniform_(conv.weight, a=1)
        if norm is None:
            nn.init.constant_(conv.bias, 0)
        module = [conv]
        if norm is not None and len(norm) > 0:
            module.append(nn.GroupNorm(32, out_channels) if norm == "GN" else get_norm(norm, out_channels))
        if activation is not None:
            module.append(nn.ReLU(inplace=True))
        return nn.Sequential(*module) if len(module) > 1 else conv
    return make_conv


def compute_locations(h, w, stride, device):
    """[h*w, 2] (x, y) pixel-centre coordinates of a stride-`stride` grid in input pixels (:929-943)."""
    shifts_x = torch.arange(0, w * stride, step=stride, dtype=torch.float32, device=device)
    shifts_y = torch.arange(0, h * stride, step=stride, dtype=torch.float32, device=device)
    shift_y, shift_x = torch.meshgrid(shifts_y, shifts_x, indexing="ij")
    return torch.stack((shift_x.reshape(-1), shift_y.reshape(-1)), dim=1) + stride // 2


def parse_dynamic_params(params, channels, weight_nums, bias_nums):
    """Split [num_insts, num_gen_params] into per-layer grouped-conv weights / biases (:945-979):
    weights[l] [num_insts*out_l, in_l, 1, 1], biases[l] [num_insts*out_l]."""
    assert params.dim() == 2
    num_insts = params.size(0)
    num_layers = len(weight_nums)
    splits = list(torch.split_with_sizes(params, list(weight_nums) + list(bias_nums), dim=1))
    weight_splits, bias_splits = splits[:num_layers], splits[num_layers:]
    for l in range(num_layers):
        out_l = channels if l < num_layers - 1 else 1
        weight_splits[l] = weight_splits[l].reshape(num_insts * out_l, -1, 1, 1)
        if bias_splits:
            bias_splits[l] = bias_splits[l].reshape(num_insts * out_l)
    return weight_splits, bias_splits


class MultiScaleMaskedTransformerDecoder(nn.Module):
    _version = 2

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            renamed = False
            for k in list(state_dict.keys()):
                if "static_query" in k:
                    state_dict[k.replace("static_query", "query_feat")] = state_dict.pop(k)
                    renamed = True
            if renamed:
                logging.getLogger(__name__).warning(
                    f"Weight format of {self.__class__.__name__} have changed! Applying automatic conversion now ...")
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def __init__(self, in_channels, mask_classification=True, *, hidden_dim: int, num_queries: int, nheads: int,
                 dim_feedforward: int, dec_layers: int, pre_norm: bool, mask_dim: int, enforce_input_project: bool,
                 points_num, sem_loss_on, norm, rel_coord):
        super().__init__()
        self.mask_classification = mask_classification
        self.pe_layer = PositionEmbeddingSine(hidden_dim // 2, normalize=True)

        self.num_heads = nheads
        self.num_layers = dec_layers
        self.transformer_self_attention_layers = nn.ModuleList()
        self.transformer_cross_attention_layers = nn.ModuleList()
        self.transformer_ffn_layers = nn.ModuleList()
        for _ in range(self.num_layers):
            self.transformer_self_attention_layers.append(
                SelfAttentionLayer(d_model=hidden_dim, nhead=nheads, dropout=0.0, normalize_before=pre_norm))
            self.transformer_cross_attention_layers.append(
                CrossAttentionLayer(d_model=hidden_dim, nhead=nheads, dropout=0.0, normalize_before=pre_norm,
                                    points_num=points_num))
            self.transformer_ffn_layers.append(
                FFNLayer(d_model=hidden_dim, dim_feedforward=dim_feedforward, dropout=0.0,
                         normalize_before=pre_norm))
        self.decoder_norm = nn.LayerNorm(hidden_dim)

        self.num_queries = num_queries
        self.query_feat = nn.Embedding(num_queries, hidden_dim)      # learnable query content
        self.query_embed = nn.Embedding(num_queries, hidden_dim)     # learnable query position
        self.hidden_dim = hidden_dim
        self.num_feature_levels = 3
        self.level_embed = nn.Embedding(self.num_feature_levels, hidden_dim)
        self.input_proj = nn.ModuleList()
        for _ in range(self.num_feature_levels):
            if in_channels != hidden_dim or enforce_input_project:
                self.input_proj.append(Conv2d(in_channels, hidden_dim, kernel_size=1))
                c2_xavier_fill(self.input_proj[-1])
            else:
                self.input_proj.append(nn.Sequential())

        # position query: reference points, their per-layer scale, and the iterative update
        self.ref_point_head = MLP(hidden_dim, hidden_dim, points_num * 2, 2)
        self.query_scale = MLP(hidden_dim, hidden_dim * 2, hidden_dim * 2 * points_num, 2)
        self.point_embed_diff_each_layer = False
        self.point_embed = MLP(hidden_dim, hidden_dim, 2 * points_num, 3)
        for m in (self.ref_point_head, self.query_scale, self.point_embed):
            _init_mlp(m)

        # dynamic mask head: (mask_dim [+2 rel coords]) -> 8 -> 8 -> 1, weights generated per query by `controller`
        self.in_channels = mask_dim
        self.dynamic_mask_channels = 8
        self.controller_layers = 3
        self.mask_out_stride = 4
        self.rel_coord = rel_coord
        ch = self.dynamic_mask_channels
        first_in = self.in_channels + 2 if self.rel_coord else self.in_channels
        self.weight_nums = [first_in * ch] + [ch * ch] * (self.controller_layers - 2) + [ch]
        self.bias_nums = [ch] * (self.controller_layers - 1) + [1]
        self.num_gen_params = sum(self.weight_nums) + sum(self.bias_nums)
        self.controller = MLP(hidden_dim, hidden_dim, self.num_gen_params, 3)
        _init_mlp(self.controller)

        self.mask_head = Conv2d(hidden_dim, mask_dim, 1, padding=0)
        nn.init.kaiming_uniform_(self.mask_head.weight, a=1)
        nn.init.constant_(self.mask_head.bias, 0)

        self.sem_loss_on = sem_loss_on
        if self.sem_loss_on:
            conv_block = conv_with_kaiming_uniform(norm, activation=True)
            self.seg_head = nn.Sequential(conv_block(hidden_dim, hidden_dim, kernel_size=3, stride=1),
                                          conv_block(hidden_dim, hidden_dim, kernel_size=3, stride=1))
            self.logits = Conv2d(hidden_dim, 1, kernel_size=1, stride=1)
            prior_prob = 0.01
            nn.init.constant_(self.logits.bias, -math.log((1 - prior_prob) / prior_prob))

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        """Same cfg keys as mask2former_transformer_decoder.py:471-500 (DEC_LAYERS counts the query-feature loss)."""
        mf = cfg.MODEL.MASK_FORMER
        assert mf.DEC_LAYERS >= 1
        return dict(in_channels=in_channels, mask_classification=mask_classification, hidden_dim=mf.HIDDEN_DIM,
                    num_queries=mf.NUM_OBJECT_QUERIES, nheads=mf.NHEADS, dim_feedforward=mf.DIM_FEEDFORWARD,
                    points_num=mf.POSITION_POINTS_NUM, dec_layers=mf.DEC_LAYERS - 1, pre_norm=mf.PRE_NORM,
                    enforce_input_project=mf.ENFORCE_INPUT_PROJ, mask_dim=cfg.MODEL.SEM_SEG_HEAD.MASK_DIM,
                    sem_loss_on=mf.SEMANTIC_LOSS_ON, norm=mf.SEMANTIC_NORM, rel_coord=mf.REL_COORD)

    # ------------------------------------------------------------------------------------------------------
    def forward(self, x, targets, mask_features, mask=None, attn_mask_threshold=0.5, criterion=None):
        """dec.py:502-645.  The tensor flow of the decoder (`_forward_core`: every shape follows (Q, N, feature sizes) only)
        is evaluated first, then the Hungarian matching of its ten mask predictions and the query-contrast items, whose
        shapes follow the targets.  The reference calls the matcher between the layers (:569, :627); its result is not read
        before the last layer's contrast items (:631-640) and the criterion, and the matcher is the only consumer of random
        numbers here (dropout 0), so the values and the random stream are those of the interleaved order.  The split is what
        lets graph.graph_training_decoder replay the core (forward and backward) from HIP graphs."""
        assert len(x) == self.num_feature_levels
        del mask          # padding masks are not applied on this path (:509-510)
        graphed = self.__dict__.get("_pct_graphed_core")
        mf_lp = _lp(mask_features)                 # one autocast cast shared by the semantic head and the mask head
        if graphed is not None and self.training and torch.is_grad_enabled():
            flat = graphed(mask_features, *x)
        else:
            flat = self._forward_core(mask_features, *x, _mf_lp=mf_lp)
        n_pred = self.num_layers + 1
        output, outputs_coords = flat[0], flat[1]
        predictions_mask = list(flat[2:2 + n_pred])
        # the semantic head (:533-534) stays outside the core: it holds the decoder's only BatchNorm (SyncBN in the shipped
        # yamls), whose cross-rank exchange cannot be part of a captured graph
        sem_logits_pred = self.logits(self.seg_head(mf_lp)) if self.sem_loss_on else None

        indices_list = []
        if targets is not None:
            many = getattr(criterion.matcher, "forward_many", None)
            if many is not None:                    # all heads' assignment problems in one device launch
                indices_list = many([{"pred_masks": pm} for pm in predictions_mask], targets)
            else:
                indices_list = [criterion.matcher({"pred_masks": pm}, targets) for pm in predictions_mask]
            from .query_contrast import query_contrast_items
            contrast_items_query, contrast_items_mask = query_contrast_items(output, predictions_mask[-1], indices_list[-1])

        out = {
            "pred_masks": predictions_mask[-1],
            "aux_outputs": self._set_aux_loss(predictions_mask),
            "reference_points": outputs_coords[-1],
            "aux_reference_points": self._set_refpoints_aux_loss(outputs_coords),
            "indices_list": indices_list,
        }
        if targets is not None:
            out["pred_qd_query"] = contrast_items_query
            out["pred_qd_mask"] = contrast_items_mask
        if self.sem_loss_on:
            out["sem_mask"] = sem_logits_pred
        return out

    def _forward_core(self, mask_features, *x, _mf_lp=None):
        """Everything of `forward` whose shapes do not depend on the targets.  Tensors in, a flat tuple of tensors out (what
        torch.cuda.make_graphed_callables captures): (query features [Q, N, C], stacked reference points [layers, N, Q, 2],
        the num_layers + 1 mask predictions).  No BatchNorm, no dropout, no random numbers inside."""
        src, pos, size_list = [], [], []
        for i in range(self.num_feature_levels):
            size_list.append(x[i].shape[-2:])
            p = self.pe_layer(x[i], None).flatten(2)
            s = self.input_proj[i](x[i]).flatten(2) + self.level_embed.weight[i][None, :, None]
            pos.append(_lp(p.permute(2, 0, 1)))       # NxCxHW -> HWxNxC (each level feeds 3 layers' projections)
            src.append(_lp(s.permute(2, 0, 1)))
        bs = src[0].shape[1]

        query_embed = _lp(self.query_embed.weight.unsqueeze(1).repeat(1, bs, 1))  # Q x N x C
        output = self.query_feat.weight.unsqueeze(1).repeat(1, bs, 1)

        predictions_mask, outputs_coords = [], []
        out_lp = None
        reference_points = self.ref_point_head(query_embed).sigmoid()
        ref_points = [reference_points]

        mf_lp = _mf_lp if _mf_lp is not None else _lp(mask_features)     # (the eager caller hands its cast over)
        feats_f32 = None
        if fused_ops.conv1x1_from_token_rows_supported(mask_features, self.mask_head):
            # small batches: the 16-channel projection on the deterministic K = 128 kernel, straight from the encoder's
            # token rows (a HIP-graph replay of the head then reproduces the eager forward bit for bit)
            mask_feat = feats_f32 = fused_ops.conv1x1_from_token_rows(mask_features, self.mask_head)
        else:
            mask_feat = self.mask_head(mf_lp)
        # the fused mask-head kernels read fp32 features: convert once, not once per prediction head
        if feats_f32 is None and mask_feat.is_cuda and mask_feat.dtype != torch.float32 and not (
                torch.is_grad_enabled() and mask_feat.requires_grad):
            feats_f32 = mask_feat.float().contiguous()

        outputs_mask, attn_mask = self.dynamic_mask_with_coords(
            mask_feat, reference_points, self.controller(output), mask_feat_stride=4, rel_coord=self.rel_coord,
            attn_mask_target_size=size_list[0], _feats_f32=feats_f32)
        predictions_mask.append(outputs_mask)

        for i in range(self.num_layers):
            query_sine_embed = gen_sineembed_for_position(reference_points)
            if i > 0:
                query_sine_embed = query_sine_embed * self.query_scale(out_lp)

            level_index = i % self.num_feature_levels
            # a query whose mask rules out every pixel attends everywhere instead (:561)
            output = self.transformer_cross_attention_layers[i](
                output, src[level_index], memory_mask=attn_mask, memory_key_padding_mask=None,
                pos=pos[level_index], query_pos=query_embed, query_sine_embed=query_sine_embed, is_first=(i == 0),
                memory_row_open=attn_mask.all(dim=-1, keepdim=True))
            output = self.transformer_self_attention_layers[i](
                output, tgt_mask=None, tgt_key_padding_mask=None, query_pos=query_embed)
            output = self.transformer_ffn_layers[i](output)

            out_lp = _lp(output)                       # shared by point_embed / controller (/ next query_scale)
            # iterative reference-point update (gradient flows through the new points only)
            # (.float(): a bf16 + fp32 add runs on a slow mixed-dtype kernel, 88 us for 12 800 elements)
            new_reference_points = (self.point_embed(out_lp).float() + inverse_sigmoid(reference_points)).sigmoid()
            if i != self.num_layers - 1:
                ref_points.append(new_reference_points)
            reference_points = new_reference_points.detach()

            outputs_mask, attn_mask = self.dynamic_mask_with_coords(
                mask_feat, new_reference_points, self.controller(out_lp), mask_feat_stride=4,
                rel_coord=self.rel_coord, attn_mask_target_size=size_list[(i + 1) % self.num_feature_levels],
                _feats_f32=feats_f32)

            decoder_output = self.decoder_norm(output).transpose(0, 1)
            outputs_coord = (self.point_embed(decoder_output).float()
                             + inverse_sigmoid(ref_points[i].transpose(0, 1))).sigmoid()
            predictions_mask.append(outputs_mask)
            outputs_coords.append(outputs_coord)

        return (output, torch.stack(outputs_coords)) + tuple(predictions_mask)

    # ------------------------------------------------------------------------------------------------------
    def dynamic_mask_with_coords(self, mask_feats, reference_points, mask_head_params, mask_feat_stride, rel_coord,
                                 attn_mask_target_size, _feats_f32=None):
        """mask_feats [N, C, H, W]; reference_points [Q, N, 2] in [0,1]; mask_head_params [Q, N, num_gen_params].
        -> (mask logits upsampled x2 [N, Q, 2H, 2W], bool attention mask [N, 1, Q, h*w] at `attn_mask_target_size`,
        True = may not attend; broadcast over heads -- the reference returns the same mask repeated per head as
        [N*heads, Q, h*w], mask2former_transformer_decoder.py:689-691)."""
        N, C, H, W = mask_feats.shape
        Q = reference_points.shape[0]
        params = mask_head_params.transpose(0, 1)                                    # [N, Q, G]
        needs_grad = torch.is_grad_enabled() and (mask_feats.requires_grad or mask_head_params.requires_grad
                                                  or reference_points.requires_grad)
        if mask_feats.is_cuda and not needs_grad and dmh.supported(mask_feats) and self.controller_layers == 3:
            # fused HIP kernel: MLP + x2 upsample + attention mask in one launch (forward only)
            out_dtype = torch.bfloat16 if mask_feats.dtype == torch.bfloat16 else torch.float32
            mask_logits, amask = dmh.dynamic_mask_head_forward(
                mask_feats, reference_points.transpose(0, 1), params, mask_feat_stride, rel_coord,
                attn_mask_target_size, out_dtype=out_dtype, feats_f32=_feats_f32)
            return mask_logits, amask.unsqueeze(1)
        mask_logits = self.mask_heads_forward_batched(
            mask_feats, reference_points.transpose(0, 1), params, mask_feat_stride, rel_coord)   # [N, Q, H, W]

        attn = F.interpolate(mask_logits, size=attn_mask_target_size, mode="bilinear", align_corners=False)
        attn_mask = (attn.sigmoid().flatten(2) < 0.5).unsqueeze(1).detach()          # [N, 1, Q, hw]
        mask_logits = F.interpolate(mask_logits, size=(H * 2, W * 2), mode="bilinear", align_corners=False)
        return mask_logits, attn_mask

    def mask_heads_forward_batched(self, mask_feats, ref_xy, params, mask_feat_stride, rel_coord):
        """The three dynamic 1x1-conv layers for all (image, query) pairs without building per-query inputs.
        ref_xy [N, Q, 2] normalised; params [N, Q, G] laid out as parse_dynamic_params splits them:
        [w0 (8 x (2+C), row-major, inputs ordered rel_x, rel_y, feat...) | w1 (8x8) | w2 (1x8) | b0 (8) | b1 (8) | b2]."""
        N, C, H, W = mask_feats.shape
        Q = ref_xy.shape[1]
        ch = self.dynamic_mask_channels
        cin = C + 2 if rel_coord else C
        wn, bn = self.weight_nums, self.bias_nums
        assert len(wn) == 3, "mask head is 3 layers"
        w0, w1, w2, b0, b1, b2 = torch.split_with_sizes(params, list(wn) + list(bn), dim=2)
        w0 = w0.reshape(N, Q, ch, cin)
        feats = mask_feats.reshape(N, C, H * W)

        if rel_coord:
            # x0[q,k,p] = sum_c w0[q,k,2+c] F[c,p] + w0[q,k,0] * (rx_q - lx_p) + w0[q,k,1] * (ry_q - ly_p) + b0[q,k]
            # (both depend on the map's shape only: built once per shape and device -- ten heads per forward would otherwise
            # rebuild them with a dozen small kernels and one pageable host->device copy each, which also cannot be
            # captured into a HIP graph)
            key = (H, W, mask_feat_stride, mask_feats.device, ref_xy.dtype)
            tab = _HEAD_TABLES.get(key)
            if tab is None:
                if len(_HEAD_TABLES) > 16:
                    _HEAD_TABLES.clear()
                tab = _HEAD_TABLES[key] = (ref_xy.new_tensor([W * mask_feat_stride, H * mask_feat_stride]),
                                           compute_locations(H, W, stride=mask_feat_stride, device=mask_feats.device))
            scale, loc = tab                                                         # [2], [HW, 2]
            inst = ref_xy * scale                                                    # [N, Q, 2]
            if (params.dtype == torch.float32 and feats.dtype == torch.float32 and ref_xy.dtype == torch.float32
                    and not torch.is_autocast_enabled()):
                # full-precision training path.  The two coordinate terms ride in the GEMM:
                #   x0 = [W_f | -w_x | -w_y] . [F; lx; ly]  +  (b0 + w_x rx + w_y ry)
                # -- one batched GEMM with the bias as its addend instead of a GEMM, the [N, Q, HW, 2] difference tensor, two
                # fused multiply-adds and a bias pass over [N, Q, 8, HW] (each of them a kernel forward and two backward on
                # 315 MB at 2 x 300 queries, 128^2 pixels).  Same sum in another order: the products w_x rx and w_x lx are
                # rounded separately (|error| <= 2^-24 * 2 * |w_x| * image size per term: 6e-5 |w_x| at 512 px).
                # (16-bit autocast keeps the form below: a pixel coordinate does not survive a bf16 GEMM operand; so does fp64,
                # where the reference's `.float()` of the relative coordinates is a rounding of its own)
                fe = torch.cat((feats, loc.to(feats.dtype).t().unsqueeze(0).expand(N, 2, H * W)), dim=1)     # [N, C + 2, HW]
                wx = torch.cat((w0[..., 2:], -w0[..., 0:2]), dim=-1).reshape(N, Q * ch, C + 2)
                bias0 = b0 + (w0[..., 0:2] * inst[:, :, None, :].to(w0.dtype)).sum(-1)                        # [N, Q, ch]
                x = torch.baddbmm(bias0.reshape(N, Q * ch, 1), wx, fe).relu_().view(N * Q, ch, H * W)
                x = torch.baddbmm(b1.reshape(N * Q, ch, 1), w1.reshape(N * Q, ch, ch), x).relu_()
                x = torch.baddbmm(b2.reshape(N * Q, 1, 1), w2.reshape(N * Q, 1, ch), x)
                return x.reshape(N, Q, H, W)
            rel = (inst[:, :, None, :] - loc[None, None, :, :]).float()              # [N, Q, HW, 2]
            x = torch.bmm(w0[..., 2:].reshape(N, Q * ch, C), feats).view(N, Q, ch, H * W)
            x = torch.addcmul(x, w0[..., 0:1], rel[:, :, None, :, 0])
            x = torch.addcmul(x, w0[..., 1:2], rel[:, :, None, :, 1])
        else:
            x = torch.bmm(w0.reshape(N, Q * ch, C), feats).view(N, Q, ch, H * W)
        x = F.relu(x + b0[..., None])
        x = F.relu(torch.matmul(w1.reshape(N, Q, ch, ch), x) + b1[..., None])
        x = torch.matmul(w2.reshape(N, Q, 1, ch), x) + b2[..., None]
        return x.reshape(N, Q, H, W)

    def mask_heads_forward(self, features, weights, biases, num_insts, FACTOR=1e4):
        """Grouped-conv formulation kept for API parity (:699-719): features [1, num_insts*cin, H, W]."""
        assert features.dim() == 4
        x = features
        for i, (w, b) in enumerate(zip(weights, biases)):
            x = F.conv2d(x, w, bias=b, stride=1, padding=0, groups=num_insts)
            if i < len(weights) - 1:
                x = F.relu(x)
        return x

    @torch.jit.unused
    def _set_aux_loss(self, outputs_seg_masks):
        return [{"pred_masks": b} for b in outputs_seg_masks[:-1]]

    @torch.jit.unused
    def _set_refpoints_aux_loss(self, outputs_coords):
        return [{"reference_points": b} for b in outputs_coords[:-1]]
