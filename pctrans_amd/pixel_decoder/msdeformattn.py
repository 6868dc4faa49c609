"""Multi-scale deformable-attention pixel decoder on the MI355X MSDeformAttn kernels.

API / state-dict mirror of pixel_decoder/msdeformattn.py of the reference:
    MSDeformAttnTransformerEncoderOnly    :23-89     flatten + concat levels, level_embed, run the encoder
    MSDeformAttnTransformerEncoderLayer   :92-131    MSDeformAttn(q = src + pos) -> +res -> LN -> FFN -> +res -> LN
    MSDeformAttnTransformerEncoder        :134-161   reference points = normalised pixel centres; N layers
    MSDeformAttnPixelDecoder              :164-360   1x1 conv + GN(32) input projections (res5 -> res3 order),
                                                     sine PE, encoder, split per level, FPN top-down to stride 4
Constructor kwargs, `forward_features` signature and return triple, and every parameter name
(`input_proj.{i}.{0,1}`, `transformer.level_embed`, `transformer.encoder.layers.{i}.*`, `adapter_{k}`, `layer_{k}`)
are kept so checkpoints interchange.  detectron2's `configurable` / registry are replaced by a plain
`from_config` classmethod that reads the same cfg keys.

Host-side differences (same math): the reference rebuilds spatial_shapes / level_start_index / reference points /
positional tables with dozens of tiny kernels and a host->device copy on every forward; they depend only on the
feature-map sizes, so they are cached per (shapes, device).  No padding masks exist on this path (the reference
builds all-False masks, msdeformattn.py:62), hence valid_ratios == 1 and the masks are never materialised.
"""
import copy
from typing import Callable, Dict, List, Optional, Union

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F
from torch.nn.init import normal_

from .. import fused_ops
from ..layers import Conv2d, ShapeSpec, c2_xavier_fill, get_norm
from ..transformer_decoder.position_encoding import PositionEmbeddingSine
from .ops.modules import MSDeformAttn


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def _get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError(f"activation should be relu/gelu, not {activation}.")


class MSDeformAttnTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def _eval_fused(self, src):
        # forward-only fused kernels: residual+LayerNorm in one pass, bias+ReLU in the GEMM epilogue
        return (not self.training or self.dropout1.p == 0.0) and src.is_cuda and src.dtype == torch.float32 \
            and not (torch.is_grad_enabled() and src.requires_grad) and self.activation is F.relu

    def forward_ffn(self, src):
        if self._eval_fused(src):
            if fused_ops.ffn_layer_norm_supported(src, self.linear1, self.linear2, self.norm2):
                # the whole block in one kernel: the [rows, d_ffn] hidden tensor stays in registers
                return fused_ops.ffn_layer_norm(src, self.linear1, self.linear2, self.norm2)
            return fused_ops.linear_layer_norm(fused_ops.linear(src, self.linear1, relu=True), self.linear2, src, self.norm2)
        src2 = self.linear2(self.dropout2(self.activation(self.linear1(src))))
        return self.norm2(src + self.dropout3(src2))

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        if self._eval_fused(src):
            # output_proj + residual + norm1 in one kernel
            src = self.self_attn.forward_add_norm(src, reference_points, src, spatial_shapes, level_start_index,
                                                  padding_mask, src, self.norm1, query_pos=pos)
        else:
            src2 = self.self_attn(self.with_pos_embed(src, pos), reference_points, src, spatial_shapes,
                                  level_start_index, padding_mask)
            src = self.norm1(src + self.dropout1(src2))
        return self.forward_ffn(src)


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """[N, S, L, 2]: normalised (x, y) centre of every pixel of every level, replicated per sampled level and
        scaled by that level's valid ratio (msdeformattn.py:141-153).  `spatial_shapes` may be a tensor or a list."""
        if torch.is_tensor(spatial_shapes):
            spatial_shapes = [(int(h), int(w)) for h, w in spatial_shapes.tolist()]
        pts = []
        for lvl, (H_, W_) in enumerate(spatial_shapes):
            ys = torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device)
            xs = torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device)
            ref_y, ref_x = torch.meshgrid(ys, xs, indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H_)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W_)
            pts.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(pts, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None,
                reference_points=None):
        if reference_points is None:
            reference_points = self.get_reference_points(spatial_shapes, valid_ratios, device=src.device)
        output = src
        for layer in self.layers:
            output = layer(output, pos, reference_points, spatial_shapes, level_start_index, padding_mask)
        return output


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024, dropout=0.1,
                 activation="relu", num_feature_levels=4, enc_n_points=4):
        super().__init__()
        self.d_model = d_model
        self.nhead = nhead
        encoder_layer = MSDeformAttnTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                            num_feature_levels, nhead, enc_n_points)
        self.encoder = MSDeformAttnTransformerEncoder(encoder_layer, num_encoder_layers)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self._geom_cache = {}
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        normal_(self.level_embed)

    def get_valid_ratio(self, mask):
        _, H, W = mask.shape
        valid_H = torch.sum(~mask[:, :, 0], 1)
        valid_W = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)

    def _geometry(self, shapes, device):
        """Device-resident spatial_shapes / level_start_index / unit reference points for a pyramid, built once."""
        key = (tuple(shapes), device)
        g = self._geom_cache.get(key)
        if g is None:
            ss = torch.as_tensor(shapes, dtype=torch.long, device=device)
            starts = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
            ones = torch.ones((1, len(shapes), 2), dtype=torch.float32, device=device)
            ref = MSDeformAttnTransformerEncoder.get_reference_points(list(shapes), ones, device)   # [1, S, L, 2]
            if len(self._geom_cache) > 16:
                self._geom_cache.clear()
            g = self._geom_cache[key] = (ss, starts, ref)
        return g

    def forward(self, srcs, pos_embeds, src_flatten=None):
        """srcs: per-level NCHW maps (or, with `src_flatten` [N, S, C] already filled, anything with their shapes)."""
        shapes = [(int(s.shape[2]), int(s.shape[3])) for s in srcs]
        bs = srcs[0].shape[0]
        if src_flatten is None:
            src_flatten = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        # the sine table is the same for every image (PositionEmbeddingSine returns an expanded view when there is no
        # padding mask): keep one [1, S, C] copy; every consumer broadcasts it over the batch
        pos_embeds = [p[:1] if (p.shape[0] > 1 and p.stride(0) == 0) else p for p in pos_embeds]
        lvl_pos_embed_flatten = torch.cat(
            [p.flatten(2).transpose(1, 2) + self.level_embed[lvl].view(1, 1, -1)
             for lvl, p in enumerate(pos_embeds)], 1)
        spatial_shapes, level_start_index, ref = self._geometry(shapes, src_flatten.device)
        valid_ratios = torch.ones((bs, len(shapes), 2), dtype=torch.float32, device=src_flatten.device)
        memory = self.encoder(src_flatten, spatial_shapes, level_start_index, valid_ratios, lvl_pos_embed_flatten,
                              None, reference_points=ref.expand(bs, -1, -1, -1))
        return memory, spatial_shapes, level_start_index


class MSDeformAttnPixelDecoder(nn.Module):
    def __init__(
        self,
        input_shape: Dict[str, ShapeSpec],
        *,
        transformer_dropout: float,
        transformer_nheads: int,
        transformer_dim_feedforward: int,
        transformer_enc_layers: int,
        conv_dim: int,
        mask_dim: int,
        norm: Optional[Union[str, Callable]] = None,
        transformer_in_features: List[str],
        common_stride: int,
    ):
        super().__init__()
        by_stride = sorted(input_shape.items(), key=lambda kv: kv[1].stride)
        self.in_features = [k for k, _ in by_stride]                       # "res2" .. "res5"
        self.feature_strides = [v.stride for _, v in by_stride]
        self.feature_channels = [v.channels for _, v in by_stride]

        tr = [(k, v) for k, v in by_stride if k in transformer_in_features]
        self.transformer_in_features = [k for k, _ in tr]
        transformer_in_channels = [v.channels for _, v in tr]
        self.transformer_feature_strides = [v.stride for _, v in tr]
        self.transformer_num_feature_levels = len(tr)

        # one 1x1 projection per encoder level, coarse -> fine (res5 first)
        chans = transformer_in_channels[::-1] if self.transformer_num_feature_levels > 1 \
            else [transformer_in_channels[-1]]
        self.input_proj = nn.ModuleList(
            nn.Sequential(Conv2d(c, conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim)) for c in chans)
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)

        self.transformer = MSDeformAttnTransformerEncoderOnly(
            d_model=conv_dim, dropout=transformer_dropout, nhead=transformer_nheads,
            dim_feedforward=transformer_dim_feedforward, num_encoder_layers=transformer_enc_layers,
            num_feature_levels=self.transformer_num_feature_levels)
        self.pe_layer = PositionEmbeddingSine(conv_dim // 2, normalize=True)

        self.mask_dim = mask_dim
        self.maskformer_num_feature_levels = 3       # the decoder always consumes 3 scales
        self.common_stride = common_stride

        # extra FPN levels between the finest encoder level and `common_stride`
        stride = min(self.transformer_feature_strides)
        self.num_fpn_levels = int(np.log2(stride) - np.log2(self.common_stride))
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channels in enumerate(self.feature_channels[:self.num_fpn_levels]):
            lateral_conv = Conv2d(in_channels, conv_dim, kernel_size=1, bias=use_bias, norm=get_norm(norm, conv_dim))
            output_conv = Conv2d(conv_dim, conv_dim, kernel_size=3, stride=1, padding=1, bias=use_bias,
                                 norm=get_norm(norm, conv_dim), activation=F.relu)
            c2_xavier_fill(lateral_conv)
            c2_xavier_fill(output_conv)
            self.add_module("adapter_{}".format(idx + 1), lateral_conv)
            self.add_module("layer_{}".format(idx + 1), output_conv)
            lateral_convs.append(lateral_conv)
            output_convs.append(output_conv)
        self.lateral_convs = lateral_convs[::-1]     # top-down order
        self.output_convs = output_convs[::-1]

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        """Same cfg keys as msdeformattn.py:294-312 (dim_feedforward is fixed at 1024 there, too)."""
        head, mf = cfg.MODEL.SEM_SEG_HEAD, cfg.MODEL.MASK_FORMER
        return dict(
            input_shape={k: v for k, v in input_shape.items() if k in head.IN_FEATURES},
            conv_dim=head.CONVS_DIM, mask_dim=head.MASK_DIM, norm=head.NORM,
            transformer_dropout=mf.DROPOUT, transformer_nheads=mf.NHEADS, transformer_dim_feedforward=1024,
            transformer_enc_layers=head.TRANSFORMER_ENC_LAYERS,
            transformer_in_features=head.DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES,
            common_stride=head.COMMON_STRIDE)

    def forward_features(self, features):
        """features: {"res2".."res5": NCHW} -> (mask_features [N,C,H/4,W/4], coarsest encoder map, 3 coarse->fine maps).
        Runs in fp32 with autocast off like the reference (msdeformattn.py:314-320)."""
        dev_type = next(iter(features.values())).device.type
        with torch.autocast(device_type=dev_type, enabled=False):
            srcs, pos = [], []
            xs = [features[f].float() for f in self.transformer_in_features[::-1]]
            src_flatten = None
            sizes = [x.shape[2] * x.shape[3] for x in xs]
            if all(fused_ops.conv1x1_groupnorm_tokens_supported(x, self.input_proj[idx][0], self.input_proj[idx][1])
                   for idx, x in enumerate(xs)):
                # 1x1 projection + GroupNorm + flatten(2).transpose(1, 2) + the concat over the levels: one entry per level, the
                # [N, 128, H, W] intermediate never exists (csrc/conv1x1_split.hip)
                import types
                src_flatten = torch.empty((xs[0].shape[0], sum(sizes), 128), dtype=torch.float32, device=xs[0].device)
                start = 0
                for idx, x in enumerate(xs):
                    fused_ops.conv1x1_groupnorm_tokens_into(x, self.input_proj[idx][0], self.input_proj[idx][1], src_flatten, start)
                    start += sizes[idx]
                srcs = [types.SimpleNamespace(shape=(x.shape[0], 128, x.shape[2], x.shape[3])) for x in xs]
            else:
                convs = [fused_ops.conv1x1_nchw(x, self.input_proj[idx][0]) for idx, x in enumerate(xs)]
                if all(fused_ops.groupnorm_flatten_supported(c, self.input_proj[idx][1]) for idx, c in enumerate(convs)):
                    # GroupNorm + flatten(2).transpose(1, 2) + the concat over the levels in one pass per level
                    src_flatten = torch.empty((convs[0].shape[0], sum(sizes), convs[0].shape[1]), dtype=torch.float32,
                                              device=convs[0].device)
                    start = 0
                    for idx, c in enumerate(convs):
                        fused_ops.groupnorm_flatten_into(c, self.input_proj[idx][1], src_flatten, start)
                        start += sizes[idx]
                    srcs = convs                                       # only their shapes are used from here on
                else:
                    srcs = [self.input_proj[idx][1](c) for idx, c in enumerate(convs)]
            pos = [self.pe_layer(x) for x in xs]

            y, spatial_shapes, level_start_index = self.transformer(srcs, pos, src_flatten=src_flatten)
            bs = y.shape[0]
            sizes = [int(s.shape[2]) * int(s.shape[3]) for s in srcs]
            out = [z.transpose(1, 2).reshape(bs, -1, s.shape[2], s.shape[3])
                   for z, s in zip(torch.split(y, sizes, dim=1), srcs)]

            for idx, f in enumerate(self.in_features[:self.num_fpn_levels][::-1]):
                cur_fpn = self.lateral_convs[idx](features[f].float())
                y = out[-1] + F.interpolate(cur_fpn, size=out[-1].shape[-2:], mode="bilinear", align_corners=False)
                out.append(self.output_convs[idx](y))

            multi_scale_features = out[:self.maskformer_num_feature_levels]
            return out[-1], out[0], multi_scale_features
